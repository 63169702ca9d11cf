// polygon_builder.hpp -- ms2.PolygonBuilder mirror [external: soypat/geometry ms2, sdfx lineage].
// Call sites in the reference: forge/threads/iso.go:46-72 (AddXY, Smooth), hexhead.go:18-20 (Nagon),
// knurl.go:29-36, bolt.go:83-88 (Chamfer). Restated from the published sdfx algorithm
// (deadsy/sdfx sdf/polygon.go: smoothVertex, Nagon) in float32; PARITY UNPINNED (module not vendored).
#pragma once
#include <stdexcept>
#include <vector>

#include "ms.hpp"

namespace gsdf {

class PolygonBuilder {
 public:
  struct Vertex {
    Vec2 v;
    float radius = 0;
    int facets = 0;
    bool smooth = false;
    // Smooth marks the vertex for rounding with the given radius and facet count.
    Vertex& Smooth(float r, int f) {
      if (r > 0 && f > 0) { radius = r; facets = f; smooth = true; }
      return *this;
    }
    // Chamfer: 1-facet smoothing (size exact only for 90 degree corners).
    Vertex& Chamfer(float size) {
      const float sqrtHalf = 0.7071067811865476f;
      if (size > 0) { radius = size * sqrtHalf; facets = 1; smooth = true; }
      return *this;
    }
  };

  Vertex& AddXY(float x, float y) {
    verts_.push_back(Vertex{{x, y}});
    return verts_.back();
  }
  Vertex& AddVec(Vec2 v) { return AddXY(v.X, v.Y); }

  // Nagon appends the vertices of an n sided regular polygon of given center-to-vertex radius.
  void Nagon(int n, float radius) {
    if (n < 3) return;
    Mat2 m = RotationMat2(2 * kPiF / (float)n);
    Vec2 p{radius, 0};
    for (int i = 0; i < n; i++) {
      AddXY(p.X, p.Y);
      p = MulMatVec(m, p);
    }
  }

  // AppendVecs resolves smoothing/chamfers and returns the final closed polygon vertices.
  std::vector<Vec2> AppendVecs() const {
    std::vector<Vertex> vl = verts_;
    if (vl.size() < 3) throw std::invalid_argument("polygon needs at least 3 vertices");
    // smoothVertices: repeat until no vertex is smoothed.
    bool done = false;
    while (!done) {
      done = true;
      for (size_t i = 0; i < vl.size(); i++) {
        if (smoothVertex(vl, i)) { done = false; break; }
      }
    }
    std::vector<Vec2> out;
    out.reserve(vl.size());
    for (auto& v : vl) out.push_back(v.v);
    return out;
  }

 private:
  static float sign(float x) { return x == 0 ? 0.0f : (x < 0 ? -1.0f : 1.0f); }

  static bool smoothVertex(std::vector<Vertex>& vl, size_t i) {
    Vertex v = vl[i];
    if (!v.smooth) return false;
    size_t n = vl.size();
    const Vertex& vn = vl[(i + 1) % n];
    const Vertex& vp = vl[(i + n - 1) % n];
    Vec2 v0 = Unit(Sub(vp.v, v.v));
    Vec2 v1 = Unit(Sub(vn.v, v.v));
    float theta = acosf32(Dot(v0, v1));
    float d1 = v.radius / tanf32(theta / 2);
    if (d1 > Norm(Sub(vp.v, v.v)) || d1 > Norm(Sub(vn.v, v.v))) {
      vl[i].smooth = false;  // unable to smooth - radius too large
      return false;
    }
    Vec2 p0 = Add(v.v, Scale(d1, v0));                       // tangent point
    float d2 = v.radius / (float)std::sin((double)(theta / 2));  // vertex -> circle centre
    Vec2 vc = Unit(Add(v0, v1));
    Vec2 c = Add(v.v, Scale(d2, vc));
    float dtheta = sign(Cross(v1, v0)) * (kPiF - theta) / (float)v.facets;
    Mat2 rm = RotationMat2(dtheta);
    Vec2 rv = Sub(p0, c);
    std::vector<Vertex> pts;
    for (int j = 0; j <= v.facets; j++) {
      pts.push_back(Vertex{Add(c, rv)});
      rv = MulMatVec(rm, rv);
    }
    vl.erase(vl.begin() + (long)i);
    vl.insert(vl.begin() + (long)i, pts.begin(), pts.end());
    return true;
  }

  std::vector<Vertex> verts_;
};

}  // namespace gsdf
