// builder.hpp -- C++ mirror of the reference's gsdf.Builder (same method names, argument meaning,
// validation and error behaviour), producing the flattened tree blob of include/gsdf_program.h.
//
// Reference: /root/reference/gsdf.go:42-139 (Builder, flags, error accumulation),
//   primitives.go, primitives2d.go (constructors + Bounds), operations.go, operations2d.go.
// Every node is appended to one array; a Shader3D/Shader2D is an index into it, so shared subtrees
// (the reference shares pointers, e.g. examples/knurled-cylinder/knurled-cyl.go:83-104) stay shared.
#pragma once
#include <cstdarg>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "../include/gsdf_program.h"
#include "ms.hpp"

namespace gsdf {

struct Shader3D { int id = -1; bool valid() const { return id >= 0; } };
struct Shader2D { int id = -1; bool valid() const { return id >= 0; } };

enum Flags : uint64_t {
  FlagNoDimensionPanic = 1 << 0,  // gsdf.go:33
  FlagUseShaderBuffers = 1 << 1,
  FlagNoShaderBuffers = 1 << 2,
};

constexpr float tribisect = 0.8660254037844386467637231707529361834714026269051903140279034897f;
constexpr float sqrt3 = 1.7320508075688772935274463415058723669428052538103806280558069794f;
constexpr float largenum = 1e20f;
constexpr float epstol = 6e-7f;

class Builder {
 public:
  explicit Builder(uint64_t flags = 0) : flags_(flags) {}
  void SetFlags(uint64_t f) { flags_ = f; }
  uint64_t GetFlags() const { return flags_; }
  // Err returns accumulated errors (gsdf.go Builder.Err) and keeps them.
  const std::vector<std::string>& Errs() const { return errs_; }
  bool HasErr() const { return !errs_.empty(); }
  void ClearErrs() { errs_.clear(); }

  // ---------------- 3D primitives (primitives.go) ----------------
  Shader3D NewSphere(float r);
  Shader3D NewBox(float x, float y, float z, float round);
  Shader3D NewBoxFrame(float dimX, float dimY, float dimZ, float e);
  Shader3D NewTorus(float greaterRadius, float lesserRadius);
  Shader3D NewCylinder(float r, float h, float rounding);
  Shader3D NewHexagonalPrism(float face2Face, float h);
  Shader3D NewTriangularPrism(float triHeight, float extrudeLength);
  Shader3D NewBoundsBoxFrame(Box3 bb);
  // ---------------- 3D operations (operations.go) ----------------
  Shader3D Union(const std::vector<Shader3D>& shaders);
  Shader3D Union(Shader3D a, Shader3D b) { return Union(std::vector<Shader3D>{a, b}); }
  Shader3D Difference(Shader3D a, Shader3D b);
  Shader3D Intersection(Shader3D a, Shader3D b);
  Shader3D Xor(Shader3D a, Shader3D b);
  Shader3D SmoothUnion(float k, Shader3D a, Shader3D b);
  Shader3D SmoothDifference(float k, Shader3D a, Shader3D b);
  Shader3D SmoothIntersect(float k, Shader3D a, Shader3D b);
  Shader3D Scale(Shader3D s, float scaleFactor);
  Shader3D Symmetry(Shader3D s, bool mirrorX, bool mirrorY, bool mirrorZ);
  Shader3D Transform(Shader3D s, const Mat4& m);
  Shader3D Rotate(Shader3D s, float radians, Vec3 axis);
  Shader3D Translate(Shader3D s, float dirX, float dirY, float dirZ);
  Shader3D Offset(Shader3D s, float sdfAdd);
  Shader3D Array(Shader3D s, float spacingX, float spacingY, float spacingZ, int nx, int ny, int nz);
  Shader3D Elongate(Shader3D s, float dirX, float dirY, float dirZ);
  Shader3D Shell(Shader3D s, float thickness);
  Shader3D CircularArray(Shader3D s, int numInstances, int circleDiv);
  Shader3D Twist(Shader3D s, float k);
  // ---------------- 2D -> 3D (operations2d.go:114-208) ----------------
  Shader3D Extrude(Shader2D s, float h);
  Shader3D Revolve(Shader2D s, float axisOffset);
  // threads.Screw node (forge/threads/threads.go:62-94): raw node; parameter derivation is in threads.hpp.
  Shader3D NewScrewNode(Shader2D thread, float pitch, float lead, float lengthDiv2, float taper);
  // ---------------- 2D primitives (primitives2d.go) ----------------
  Shader2D NewLine2D(float x0, float y0, float x1, float y1, float width);
  Shader2D NewLines2D(const std::vector<std::array<Vec2, 2>>& segments, float width);
  Shader2D NewArc(float radius, float arcAngle, float thick);
  Shader2D NewCircle(float radius);
  Shader2D NewEquilateralTriangle(float triangleHeight);
  Shader2D NewRectangle(float x, float y);
  Shader2D NewHexagon(float side);
  Shader2D NewOctagon(float constrain);
  Shader2D NewEllipse(float a, float b);
  Shader2D NewPolygon(std::vector<Vec2> vertices);
  Shader2D NewDiamond2D(float x_width, float y_height);
  Shader2D NewRoundedX(float width, float thick);
  Shader2D NewQuadraticBezier2D(Vec2 a, Vec2 b, Vec2 c, float thick);
  // ---------------- 2D operations (operations2d.go) ----------------
  Shader2D Union2D(const std::vector<Shader2D>& shaders);
  Shader2D Difference2D(Shader2D a, Shader2D b);
  Shader2D Intersection2D(Shader2D a, Shader2D b);
  Shader2D Xor2D(Shader2D a, Shader2D b);
  Shader2D Array2D(Shader2D s, float spacingX, float spacingY, int nx, int ny);
  Shader2D Offset2D(Shader2D s, float sdfAdd);
  Shader2D Translate2D(Shader2D s, float dirX, float dirY);
  Shader2D Rotate2D(Shader2D s, float theta);
  Shader2D Symmetry2D(Shader2D s, bool mirrorX, bool mirrorY);
  Shader2D Annulus(Shader2D s, float sub);
  Shader2D CircularArray2D(Shader2D s, int numInstances, int circleDiv);
  Shader2D Scale2D(Shader2D s, float scale);
  Shader2D TranslateMulti2D(Shader2D s, const std::vector<Vec2>& displacements);
  Shader2D Elongate2D(Shader2D s, float dirX, float dirY);

  // ---------------- introspection / flattening ----------------
  Box3 Bounds(Shader3D s) const { return bb3_.at((size_t)s.id); }
  Box2 Bounds(Shader2D s) const { return bb2_.at((size_t)s.id); }
  int Op(int id) const { return nodes_.at((size_t)id).op; }
  size_t NumNodes() const { return nodes_.size(); }
  // Tree returns a borrowed view rooted at `root` (valid until the next builder mutation).
  gsdf_tree Tree(Shader3D root) const;
  gsdf_tree Tree2D(Shader2D root) const;
  const std::vector<gsdf_node>& Nodes() const { return nodes_; }
  const std::vector<uint32_t>& Links() const { return links_; }
  const std::vector<float>& Aux() const { return aux_; }

  void shapeErrorf(const char* fmt, ...);  // gsdf.go: panics unless FlagNoDimensionPanic
  void nilsdf(const char* ctx);

 private:
  int push(int op, std::initializer_list<float> params, std::initializer_list<int> children);
  int push3(int op, std::initializer_list<float> params, std::initializer_list<int> children, Box3 bb);
  int push2(int op, std::initializer_list<float> params, std::initializer_list<int> children, Box2 bb);
  bool is3(int id) const { return id >= 0 && (size_t)id < nodes_.size() && !gsdf_op_is2d(nodes_[(size_t)id].op); }
  bool is2(int id) const { return id >= 0 && (size_t)id < nodes_.size() && gsdf_op_is2d(nodes_[(size_t)id].op); }
  bool useShaderBuffer(int) const { return false; }  // SSBO/non-SSBO nodes evaluate identically on the CPU

  uint64_t flags_;
  std::vector<std::string> errs_;
  std::vector<gsdf_node> nodes_;
  std::vector<uint32_t> links_;
  std::vector<float> aux_;
  std::vector<Box3> bb3_;
  std::vector<Box2> bb2_;
};

}  // namespace gsdf
