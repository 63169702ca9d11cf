// builder.cpp -- see builder.hpp. Constructors mirror the reference's validation and Bounds().
#include <array>

#include "builder.hpp"

namespace gsdf {

void Builder::shapeErrorf(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (flags_ & FlagNoDimensionPanic) errs_.push_back(buf);
  else throw std::invalid_argument(buf);
}
void Builder::nilsdf(const char* ctx) { throw std::invalid_argument(std::string("nil shader argument: ") + ctx); }

int Builder::push(int op, std::initializer_list<float> params, std::initializer_list<int> children) {
  gsdf_node n{};
  n.op = (uint16_t)op;
  n.nchild = (uint16_t)children.size();
  n.link_off = (uint32_t)links_.size();
  n.aux_off = (uint32_t)aux_.size();
  n.aux_len = 0;
  int i = 0;
  for (float p : params) n.p[i++] = p;
  for (int c : children) links_.push_back((uint32_t)c);
  nodes_.push_back(n);
  bb3_.push_back(Box3{});
  bb2_.push_back(Box2{});
  return (int)nodes_.size() - 1;
}
int Builder::push3(int op, std::initializer_list<float> params, std::initializer_list<int> children, Box3 bb) {
  int id = push(op, params, children);
  bb3_[(size_t)id] = bb;
  return id;
}
int Builder::push2(int op, std::initializer_list<float> params, std::initializer_list<int> children, Box2 bb) {
  int id = push(op, params, children);
  bb2_[(size_t)id] = bb;
  return id;
}

gsdf_tree Builder::Tree(Shader3D root) const {
  if (!is3(root.id)) throw std::invalid_argument("Tree: root is not a 3D shader");
  gsdf_tree t{};
  t.nodes = nodes_.data(); t.n_nodes = (uint32_t)nodes_.size();
  t.links = links_.data(); t.n_links = (uint32_t)links_.size();
  t.aux = aux_.data(); t.n_aux = (uint32_t)aux_.size();
  t.root = (uint32_t)root.id;
  Box3 b = Bounds(root);
  t.bb[0] = b.Min.X; t.bb[1] = b.Min.Y; t.bb[2] = b.Min.Z; t.bb[3] = b.Max.X; t.bb[4] = b.Max.Y; t.bb[5] = b.Max.Z;
  return t;
}
gsdf_tree Builder::Tree2D(Shader2D root) const {
  if (!is2(root.id)) throw std::invalid_argument("Tree2D: root is not a 2D shader");
  gsdf_tree t{};
  t.nodes = nodes_.data(); t.n_nodes = (uint32_t)nodes_.size();
  t.links = links_.data(); t.n_links = (uint32_t)links_.size();
  t.aux = aux_.data(); t.n_aux = (uint32_t)aux_.size();
  t.root = (uint32_t)root.id;
  Box2 b = Bounds(root);
  t.bb[0] = b.Min.X; t.bb[1] = b.Min.Y; t.bb[2] = 0; t.bb[3] = b.Max.X; t.bb[4] = b.Max.Y; t.bb[5] = 0;
  return t;
}

// =============================== 3D primitives (primitives.go) ===============================
Shader3D Builder::NewSphere(float r) {  // :27-33, bounds :50-55
  if (!(r > 0)) shapeErrorf("zero or negative sphere radius");
  return {push3(GSDF_SPHERE, {r}, {}, Box3{{-r, -r, -r}, {r, r, r}})};
}
Shader3D Builder::NewBox(float x, float y, float z, float round) {  // :58-66, bounds :90-92
  if (round < 0 || round > x / 2 || round > y / 2 || round > z / 2) shapeErrorf("invalid box rounding value");
  if (x <= 0 || y <= 0 || z <= 0) shapeErrorf("zero or negative box dimension");
  return {push3(GSDF_BOX, {x, y, z, round}, {}, NewCenteredBox(Vec3{}, Vec3{x, y, z}))};
}
Shader3D Builder::NewBoxFrame(float dimX, float dimY, float dimZ, float e) {  // :254-264
  e /= 2;
  if (dimX <= 0 || dimY <= 0 || dimZ <= 0 || e <= 0) shapeErrorf("negative or zero BoxFrame dimension");
  Vec3 d{dimX, dimY, dimZ};
  if (2 * e > MinOf(d)) shapeErrorf("BoxFrame edge thickness too large");
  return {push3(GSDF_BOXFRAME, {dimX, dimY, dimZ, e}, {}, NewCenteredBox(Vec3{}, d))};
}
Shader3D Builder::NewTorus(float R, float r) {  // :211-219, bounds :236-242
  if (R < 2 * r) shapeErrorf("too large torus lesser radius");
  if (R <= 0 || r <= 0) shapeErrorf("invalid torus parameter");
  float RR = r + R;
  return {push3(GSDF_TORUS, {R, r}, {}, Box3{{-RR, -RR, -r}, {RR, RR, r}})};
}
Shader3D Builder::NewCylinder(float r, float h, float rounding) {  // :99-109, bounds :117-122
  bool okRounding = rounding >= 0 && rounding < r && rounding < h / 2;
  if (!okRounding) shapeErrorf("invalid cylinder rounding");
  if (!(r > 0 && h > 0)) shapeErrorf("bad cylinder dimension");
  return {push3(GSDF_CYLINDER, {r, h, rounding}, {}, Box3{{-r, -r, -h / 2}, {r, r, h / 2}})};
}
Shader3D Builder::NewHexagonalPrism(float face2Face, float h) {  // :157-162, bounds :169-176
  if (face2Face <= 0 || h <= 0) shapeErrorf("invalid hexagonal prism parameter");
  float l = face2Face, lx = l / tribisect;
  return {push3(GSDF_HEX, {face2Face, h}, {}, Box3{{-lx, -l, -h}, {lx, l, h}})};
}
Shader3D Builder::NewTriangularPrism(float triHeight, float extrudeLength) {  // :198-205
  bool ok = extrudeLength > 0 && !(std::isinf(extrudeLength) && extrudeLength > 0);
  if (!ok) shapeErrorf("bad triangular prism extrude length");
  return Extrude(NewEquilateralTriangle(triHeight), extrudeLength);
}
Shader3D Builder::NewBoundsBoxFrame(Box3 bb) {  // :11-19
  Vec3 size = bb.Size();
  float ft = MaxOf(size) / 256;
  size = AddScalar(2 * ft, size);
  Shader3D b = NewBoxFrame(size.X, size.Y, size.Z, ft);
  Vec3 c = bb.Center();
  return Translate(b, c.X, c.Y, c.Z);
}

// =============================== 3D operations (operations.go) ===============================
Shader3D Builder::Union(const std::vector<Shader3D>& shaders) {  // :31-53
  if (shaders.size() < 2) throw std::invalid_argument("need at least 2 arguments to Union");
  std::vector<uint32_t> joined;
  for (size_t i = 0; i < shaders.size(); i++) {
    Shader3D s = shaders[i];
    if (!is3(s.id)) nilsdf("Union");
    const gsdf_node& n = nodes_[(size_t)s.id];
    if (n.op == GSDF_UNION) {  // nested unions are spliced
      for (uint32_t c = 0; c < n.nchild; c++) joined.push_back(links_[n.link_off + c]);
    } else {
      joined.push_back((uint32_t)s.id);
    }
  }
  Box3 bb = bb3_[joined[0]];
  for (size_t i = 1; i < joined.size(); i++) bb = bb.Union(bb3_[joined[i]]);
  int id = push3(GSDF_UNION, {}, {}, bb);
  nodes_[(size_t)id].nchild = (uint16_t)joined.size();
  nodes_[(size_t)id].link_off = (uint32_t)links_.size();
  for (uint32_t c : joined) links_.push_back(c);
  return {id};
}
Shader3D Builder::Difference(Shader3D a, Shader3D b) {  // :113-118 bounds = s1
  if (!is3(a.id) || !is3(b.id)) nilsdf("Difference");
  return {push3(GSDF_DIFF, {}, {a.id, b.id}, Bounds(a))};
}
Shader3D Builder::Intersection(Shader3D a, Shader3D b) {  // :156-161
  if (!is3(a.id) || !is3(b.id)) nilsdf("Intersection");
  return {push3(GSDF_INTERSECT, {}, {a.id, b.id}, Bounds(a).Intersect(Bounds(b)))};
}
Shader3D Builder::Xor(Shader3D a, Shader3D b) {  // :201-206
  if (!is3(a.id) || !is3(b.id)) nilsdf("Xor");
  return {push3(GSDF_XOR, {}, {a.id, b.id}, Bounds(a).Union(Bounds(b)))};
}
Shader3D Builder::SmoothUnion(float k, Shader3D a, Shader3D b) {  // :562-567
  if (!is3(a.id) || !is3(b.id)) nilsdf("SmoothUnion");
  return {push3(GSDF_SMOOTH_UNION, {k}, {a.id, b.id}, Bounds(a).Union(Bounds(b)))};
}
Shader3D Builder::SmoothDifference(float k, Shader3D a, Shader3D b) {  // :610-615 (embeds diff: bounds s1)
  if (!is3(a.id) || !is3(b.id)) nilsdf("SmoothDifference");
  return {push3(GSDF_SMOOTH_DIFF, {k}, {a.id, b.id}, Bounds(a))};
}
Shader3D Builder::SmoothIntersect(float k, Shader3D a, Shader3D b) {  // :642-647 (embeds intersect)
  if (!is3(a.id) || !is3(b.id)) nilsdf("SmoothIntersect");
  return {push3(GSDF_SMOOTH_INTERSECT, {k}, {a.id, b.id}, Bounds(a).Intersect(Bounds(b)))};
}
Shader3D Builder::Scale(Shader3D s, float f) {  // :246-258
  if (!is3(s.id)) nilsdf("Scale");
  return {push3(GSDF_SCALE, {f}, {s.id}, Bounds(s).Scale(Vec3{f, f, f}))};
}
Shader3D Builder::Symmetry(Shader3D s, bool mx, bool my, bool mz) {  // :285-310
  if (!is3(s.id)) nilsdf("Symmetry");
  if (!mx && !my && !mz) shapeErrorf("ineffective symmetry");
  Box3 box = Bounds(s);
  if (mx) box.Min.X = minf(box.Min.X, -box.Max.X);
  if (my) box.Min.Y = minf(box.Min.Y, -box.Max.Y);
  if (mz) box.Min.Z = minf(box.Min.Z, -box.Max.Z);
  float bits = (float)((mx ? 1 : 0) | (my ? 2 : 0) | (mz ? 4 : 0));
  return {push3(GSDF_SYMMETRY, {bits}, {s.id}, box)};
}
Shader3D Builder::Transform(Shader3D s, const Mat4& m) {  // :339-364
  if (!is3(s.id)) nilsdf("Transform");
  float det = m.Determinant();
  if (absf(det) < epstol) shapeErrorf("singular Mat4");
  Mat4 inv = m.Inverse();
  int id = push3(GSDF_TRANSFORM, {}, {s.id}, m.MulBox(Bounds(s)));
  nodes_[(size_t)id].aux_off = (uint32_t)aux_.size();
  nodes_[(size_t)id].aux_len = 16;
  for (int i = 0; i < 16; i++) aux_.push_back(inv.m[i]);
  return {id};
}
Shader3D Builder::Rotate(Shader3D s, float radians, Vec3 axis) {  // :394-400
  if (axis == Vec3{}) shapeErrorf("null vector");
  return Transform(s, RotationMat4(radians, axis));
}
Shader3D Builder::Translate(Shader3D s, float x, float y, float z) {  // :403-414
  if (!is3(s.id)) nilsdf("Translate");
  return {push3(GSDF_TRANSLATE, {x, y, z}, {s.id}, Bounds(s).Add(Vec3{x, y, z}))};
}
Shader3D Builder::Offset(Shader3D s, float off) {  // :445-460
  if (!is3(s.id)) nilsdf("Offset");
  Box3 bb = Bounds(s);
  bb.Max = AddScalar(-off, bb.Max);
  bb.Min = AddScalar(off, bb.Min);
  return {push3(GSDF_OFFSET, {off}, {s.id}, bb.Canon())};
}
Shader3D Builder::Array(Shader3D s, float sx, float sy, float sz, int nx, int ny, int nz) {  // :484-508
  if (!is3(s.id)) nilsdf("Array");
  if (nx <= 0 || ny <= 0 || nz <= 0) shapeErrorf("invalid array repeat param");
  if (sx <= 0 || sy <= 0 || sz <= 0) shapeErrorf("invalid array spacing");
  Box3 sbb = Bounds(s);
  Vec3 size = MulElem(Vec3{(float)nx, (float)ny, (float)nz}, Vec3{sx, sy, sz});
  sbb.Max = Add(sbb.Max, size);
  return {push3(GSDF_ARRAY, {sx, sy, sz, (float)nx, (float)ny, (float)nz}, {s.id}, sbb)};
}
Shader3D Builder::Elongate(Shader3D s, float x, float y, float z) {  // :676-692
  if (!is3(s.id)) nilsdf("Elongate");
  Box3 box = Bounds(s);
  box.Max = MaxElem(box.Max, Vec3{});
  box.Max = Add(box.Max, gsdf::Scale(0.5f, Vec3{x, y, z}));
  box.Min = gsdf::Scale(-1, box.Max);
  return {push3(GSDF_ELONGATE, {x, y, z}, {s.id}, box)};
}
Shader3D Builder::Shell(Shader3D s, float thickness) {  // :722-734
  if (!is3(s.id)) nilsdf("Shell");
  return {push3(GSDF_SHELL, {thickness}, {s.id}, Bounds(s))};
}
Shader3D Builder::CircularArray(Shader3D s, int numInstances, int circleDiv) {  // :760-801
  if (!is3(s.id)) nilsdf("nil argument to circarray");
  if (circleDiv <= 1 || numInstances <= 0) shapeErrorf("invalid circarray repeat param");
  if (numInstances > circleDiv) shapeErrorf("bad circular array instances, must be less than or equal to circleDiv");
  Box3 bb = Bounds(s);
  Box2 bb2{{bb.Min.X, bb.Min.Y}, {bb.Max.X, bb.Max.Y}};
  Vec2 verts[4];
  bb2.Vertices(verts);
  float angle = 2 * kPiF / (float)circleDiv;
  Mat2 m = RotationMat2(angle);
  for (int i = 0; i < numInstances - 1; i++)
    for (int j = 0; j < 4; j++) {
      verts[j] = MulMatVec(m, verts[j]);
      bb2 = bb2.IncludePoint(verts[j]);
    }
  bb.Max.X = bb2.Max.X; bb.Max.Y = bb2.Max.Y;
  bb.Min.X = bb2.Min.X; bb.Min.Y = bb2.Min.Y;
  return {push3(GSDF_CIRCARRAY, {(float)numInstances, (float)circleDiv}, {s.id}, bb)};
}
Shader3D Builder::Twist(Shader3D s, float k) {  // :833-865
  if (!is3(s.id)) nilsdf("Twist");
  if (k == 0) shapeErrorf("zero twist parameter");
  Box3 bb = Bounds(s);
  Vec3 v[8];
  bb.Vertices(v);
  float maxR = 0;
  for (auto& p : v) {
    float r = hypotf32(p.X, p.Y);
    if (r > maxR) maxR = r;
  }
  return {push3(GSDF_TWIST, {k}, {s.id}, Box3{{-maxR, -maxR, bb.Min.Z}, {maxR, maxR, bb.Max.Z}})};
}

// =============================== 2D -> 3D ===============================
Shader3D Builder::Extrude(Shader2D s, float h) {  // operations2d.go:103-128
  if (!is2(s.id)) nilsdf("Extrude");
  if (h < 0) shapeErrorf("bad extrusion length");
  Box2 b2 = Bounds(s);
  float hd2 = h / 2;
  return {push3(GSDF_EXTRUSION, {h}, {s.id}, Box3{{b2.Min.X, b2.Min.Y, -hd2}, {b2.Max.X, b2.Max.Y, hd2}})};
}
Shader3D Builder::Revolve(Shader2D s, float axisOffset) {  // operations2d.go:152-178
  if (!is2(s.id)) shapeErrorf("nil argument to Revolve");
  if (axisOffset < 0) shapeErrorf("negative axis offset");
  Box2 b2 = Bounds(s);
  float radius = maxf(0, b2.Max.X - axisOffset);
  return {push3(GSDF_REVOLUTION, {axisOffset}, {s.id}, Box3{{-radius, b2.Min.Y, -radius}, {radius, b2.Max.Y, radius}})};
}
Shader3D Builder::NewScrewNode(Shader2D thread, float pitch, float lead, float lengthDiv2, float taper) {
  // bounds: forge/threads/threads.go:184-196
  if (!is2(thread.id)) nilsdf("Screw");
  float r = Bounds(thread).Max.Y;
  r += lengthDiv2 * tanf32(taper);
  return {push3(GSDF_SCREW, {pitch, lead, lengthDiv2, taper}, {thread.id}, Box3{{-r, -r, -lengthDiv2}, {r, r, lengthDiv2}})};
}

// =============================== 2D primitives (primitives2d.go) ===============================
Shader2D Builder::NewLine2D(float x0, float y0, float x1, float y1, float width) {  // :14-31, bounds :38-44
  bool hasNaN = x0 != x0 || y0 != y0 || x1 != x1 || y1 != y1 || width != width;
  if (hasNaN) shapeErrorf("NaN argument to NewLine2D");
  else if (width < 0) shapeErrorf("negative thickness to NewLine2D");
  Vec2 a{x0, y0}, b{x1, y1};
  float lineLen = Norm(Sub(a, b));
  if (lineLen < width * 1e-6f || lineLen < epstol) {
    if (width == 0) shapeErrorf("infimal line");
    return NewCircle(width / 2);
  }
  float w = width / 2;
  Box2 bb = Box2{a, b}.Canon();
  bb.Max = AddScalar(w, bb.Max);
  bb.Min = AddScalar(-w, bb.Min);
  return {push2(GSDF_LINE2D, {x0, y0, x1, y1, width}, {}, bb)};
}
Shader2D Builder::NewLines2D(const std::vector<std::array<Vec2, 2>>& segs, float width) {  // :68-88
  if (width < 0) shapeErrorf("negative thickness to NewLines2D");
  if (segs.size() < 2) { shapeErrorf("empty or single points"); if (segs.empty()) return {-1}; }
  for (size_t i = 0; i + 1 < segs.size(); i++)
    if (segs[i][0] == segs[i][1]) shapeErrorf("superimposed points in NewLines2D");
  float w = width / 2;
  Box2 bb = NewBox2(segs[0][0].X, segs[0][0].Y, segs[0][1].X, segs[0][1].Y);
  for (size_t i = 1; i < segs.size(); i++) { bb = bb.IncludePoint(segs[i][0]); bb = bb.IncludePoint(segs[i][1]); }
  bb.Max = AddScalar(w, bb.Max);
  bb.Min = AddScalar(-w, bb.Min);
  int id = push2(GSDF_LINES2D, {width}, {}, bb);
  nodes_[(size_t)id].aux_off = (uint32_t)aux_.size();
  nodes_[(size_t)id].aux_len = (uint32_t)(4 * segs.size());
  for (auto& s : segs) { aux_.push_back(s[0].X); aux_.push_back(s[0].Y); aux_.push_back(s[1].X); aux_.push_back(s[1].Y); }
  return {id};
}
Shader2D Builder::NewArc(float radius, float arcAngle, float thick) {  // :173-187, bounds :198-205
  bool ok = radius > 0 && arcAngle > 0 && thick >= 0;
  if (!ok) shapeErrorf("invalid argument to NewArc2D");
  const float twopi = (float)(2 * kPi);
  if ((double)arcAngle > 2 * kPi) shapeErrorf("arc angle exceeds full circle");
  else if (twopi - arcAngle < epstol) arcAngle = twopi - 1e-7f;
  float r = radius + thick;
  float rcos = radius * cosf32(arcAngle / 2) - thick;
  return {push2(GSDF_ARC2D, {radius, arcAngle, thick}, {}, Box2{{-r, rcos}, {r, r}})};
}
Shader2D Builder::NewCircle(float radius) {  // :226-232
  bool ok = radius > 0 && !(std::isinf(radius) && radius > 0);
  if (!ok) shapeErrorf("bad circle radius: %g", (double)radius);
  return {push2(GSDF_CIRCLE2D, {radius}, {}, NewBox2(-radius, -radius, radius, radius))};
}
Shader2D Builder::NewEquilateralTriangle(float h) {  // :265-271, bounds :273-282
  bool ok = h > 0 && !(std::isinf(h) && h > 0);
  if (!ok) shapeErrorf("bad equilateral triangle height");
  float side = h / tribisect;
  float longBisect = side / sqrt3;
  float shortBisect = longBisect / 2;
  return {push2(GSDF_EQTRI2D, {h}, {}, Box2{{-side / 2, -shortBisect}, {side / 2, longBisect}})};
}
Shader2D Builder::NewRectangle(float x, float y) {  // :307-313
  bool ok = x > 0 && y > 0 && !std::isinf(x) && !std::isinf(y);
  if (!ok) shapeErrorf("bad rectangle dimension");
  return {push2(GSDF_RECT2D, {x, y}, {}, Box2{{-(x / 2), -(y / 2)}, {x / 2, y / 2}})};
}
Shader2D Builder::NewHexagon(float side) {  // :348-354
  bool ok = side > 0 && !std::isinf(side);
  if (!ok) shapeErrorf("bad hexagon dimension");
  float w = side / tribisect;
  return {push2(GSDF_HEX2D, {side}, {}, NewBox2(-w, -side, w, side))};
}
Shader2D Builder::NewOctagon(float c) {  // :385-391
  if (!(c > 0)) shapeErrorf("bad octagon dimension %f", (double)c);
  return {push2(GSDF_OCT2D, {c}, {}, NewBox2(-c, -c, c, c))};
}
Shader2D Builder::NewEllipse(float a, float b) {  // :421-427
  bool ok = a > 0 && b > 0 && !std::isinf(a) && !std::isinf(b);
  if (!ok) shapeErrorf("bad ellipse dimension (a=%f, b=%f)", (double)a, (double)b);
  return {push2(GSDF_ELLIPSE2D, {a, b}, {}, NewBox2(-a, -b, a, b))};
}
Shader2D Builder::NewPolygon(std::vector<Vec2> v) {  // :458-495, bounds :497-505
  // validatePolygon
  const char* err = nullptr;
  if (v.empty()) { shapeErrorf("polygon needs at least 3 distinct vertices"); return {-1}; }
  size_t prevIdx = v.size() - 1;
  if (v[0] == v[prevIdx]) { v.pop_back(); prevIdx = v.size() ? v.size() - 1 : 0; }
  if (v.size() < 3) err = "polygon needs at least 3 distinct vertices";
  if (!err)
    for (size_t i = 0; i < v.size(); i++) {
      if (v[i].X != v[i].X || v[i].Y != v[i].Y) { err = "NaN value in vertices"; break; }
      if (v[i] == v[prevIdx]) { err = "found two consecutive equal vertices in polygon"; break; }
      prevIdx = i;
    }
  if (err) { shapeErrorf("%s", err); if (v.size() < 3) return {-1}; }
  Vec2 mn{largenum, largenum}, mx{-largenum, -largenum};
  for (auto& p : v) { mn = MinElem(mn, p); mx = MaxElem(mx, p); }
  int id = push2(GSDF_POLY2D, {}, {}, Box2{mn, mx});
  nodes_[(size_t)id].aux_off = (uint32_t)aux_.size();
  nodes_[(size_t)id].aux_len = (uint32_t)(2 * v.size());
  for (auto& p : v) { aux_.push_back(p.X); aux_.push_back(p.Y); }
  return {id};
}
Shader2D Builder::NewDiamond2D(float x, float y) {  // :560-566
  bool ok = x > 0 && y > 0 && !std::isinf(x) && !std::isinf(y);
  if (!ok) shapeErrorf("bad diamond dimension");
  return {push2(GSDF_DIAMOND2D, {x, y}, {}, Box2{{-(x / 2), -(y / 2)}, {x / 2, y / 2}})};
}
Shader2D Builder::NewRoundedX(float width, float thick) {  // :602-608
  bool ok = width > 0 && thick > 0 && !std::isinf(width) && !std::isinf(thick);
  if (!ok) shapeErrorf("bad x dimension");
  float xd2 = width / 2 + thick;
  return {push2(GSDF_X2D, {width, thick}, {}, Box2{{-xd2, -xd2}, {xd2, xd2}})};
}
Shader2D Builder::NewQuadraticBezier2D(Vec2 p0, Vec2 p1, Vec2 p2, float thick) {  // :642-672
  Vec2 mn = MinElem(p0, p2), mx = MaxElem(p0, p2);
  Vec2 one{1, 1};
  if (p1.X < mn.X || p1.X > mx.X || p1.Y < mn.Y || p1.Y > mx.Y) {
    Vec2 denom = Add(p0, Sub(p2, gsdf::Scale(2, p1)));
    Vec2 t = ClampElem(DivElem(Sub(p0, p1), denom), Vec2{}, one);
    Vec2 s = Sub(one, t);
    Vec2 q1 = MulElem(MulElem(s, s), p0);
    Vec2 q2 = gsdf::Scale(2, MulElem(MulElem(s, t), p1));
    Vec2 q3 = MulElem(p2, MulElem(t, t));
    Vec2 q = Add(q1, Add(q2, q3));
    mn = MinElem(mn, q);
    mx = MaxElem(mx, q);
  }
  mn = AddScalar(-thick / 2, mn);
  mx = AddScalar(thick / 2, mx);
  return {push2(GSDF_QUADBEZIER2D, {p0.X, p0.Y, p1.X, p1.Y, p2.X, p2.Y, thick}, {}, Box2{mn, mx})};
}

// =============================== 2D operations (operations2d.go) ===============================
Shader2D Builder::Union2D(const std::vector<Shader2D>& shaders) {  // :19-45
  if (shaders.size() < 2) throw std::invalid_argument("need at least 2 arguments to Union2D");
  std::vector<uint32_t> joined;
  for (Shader2D s : shaders) {
    if (!is2(s.id)) nilsdf("Union2D");
    const gsdf_node& n = nodes_[(size_t)s.id];
    if (n.op == GSDF_UNION2D) for (uint32_t c = 0; c < n.nchild; c++) joined.push_back(links_[n.link_off + c]);
    else joined.push_back((uint32_t)s.id);
  }
  Box2 bb = bb2_[joined[0]];
  for (size_t i = 1; i < joined.size(); i++) bb = bb.Union(bb2_[joined[i]]);
  int id = push2(GSDF_UNION2D, {}, {}, bb);
  nodes_[(size_t)id].nchild = (uint16_t)joined.size();
  nodes_[(size_t)id].link_off = (uint32_t)links_.size();
  for (uint32_t c : joined) links_.push_back(c);
  return {id};
}
Shader2D Builder::Difference2D(Shader2D a, Shader2D b) {  // :198-203
  if (!is2(a.id) || !is2(b.id)) nilsdf("Difference2D");
  return {push2(GSDF_DIFF2D, {}, {a.id, b.id}, Bounds(a))};
}
Shader2D Builder::Intersection2D(Shader2D a, Shader2D b) {  // :242-247
  if (!is2(a.id) || !is2(b.id)) nilsdf("Intersection2D");
  return {push2(GSDF_INTERSECT2D, {}, {a.id, b.id}, Bounds(a).Intersect(Bounds(b)))};
}
Shader2D Builder::Xor2D(Shader2D a, Shader2D b) {  // :286-291
  if (!is2(a.id) || !is2(b.id)) nilsdf("Xor2D");
  return {push2(GSDF_XOR2D, {}, {a.id, b.id}, Bounds(a).Union(Bounds(b)))};
}
Shader2D Builder::Array2D(Shader2D s, float sx, float sy, int nx, int ny) {  // :331-355
  if (!is2(s.id)) nilsdf("Array2D");
  if (nx <= 0 || ny <= 0) shapeErrorf("invalid array repeat param");
  bool ok = sx > 0 && sy > 0 && !std::isinf(sx) && !std::isinf(sy);
  if (!ok) shapeErrorf("bad array spacing");
  Box2 sbb = Bounds(s);
  sbb.Max = Add(sbb.Max, MulElem(Vec2{(float)nx, (float)ny}, Vec2{sx, sy}));
  return {push2(GSDF_ARRAY2D, {sx, sy, (float)nx, (float)ny}, {s.id}, sbb)};
}
Shader2D Builder::Offset2D(Shader2D s, float f) {  // :410-431
  if (!is2(s.id)) nilsdf("Offset2D");
  Box2 bb = Bounds(s);
  if (!(f > 0)) {
    bb.Max = AddScalar(-f, bb.Max);
    bb.Min = AddScalar(f, bb.Min);
  }
  return {push2(GSDF_OFFSET2D, {f}, {s.id}, bb)};
}
Shader2D Builder::Translate2D(Shader2D s, float x, float y) {  // :455-468
  if (!is2(s.id)) nilsdf("Translate2D");
  return {push2(GSDF_TRANSLATE2D, {x, y}, {s.id}, Bounds(s).Add(Vec2{x, y}))};
}
Shader2D Builder::Rotate2D(Shader2D s, float theta) {  // :495-530
  if (!is2(s.id)) nilsdf("Rotate2D");
  Mat2 m = RotationMat2(theta);
  if (absf(m.Determinant()) < epstol) shapeErrorf("badly conditioned rotation");
  Mat2 inv = m.Inverse();
  Box2 bb = Bounds(s);
  Vec2 v[4];
  bb.Vertices(v);
  Vec2 v1 = MulMatVec(m, v[0]);
  bb.Max = v1; bb.Min = v1;
  for (int i = 1; i < 4; i++) {
    Vec2 t = MulMatVec(m, v[i]);
    bb.Max = MaxElem(bb.Max, t);
    bb.Min = MinElem(bb.Min, t);
  }
  return {push2(GSDF_ROTATION2D, {inv.x00, inv.x01, inv.x10, inv.x11}, {s.id}, bb)};
}
Shader2D Builder::Symmetry2D(Shader2D s, bool mx, bool my) {  // :555-578
  if (!is2(s.id)) nilsdf("Symmetry2D");
  if (!mx && !my) shapeErrorf("ineffective symmetry");
  Box2 box = Bounds(s);
  if (mx) box.Min.X = minf(box.Min.X, -box.Max.X);
  if (my) box.Min.Y = minf(box.Min.Y, -box.Max.Y);
  return {push2(GSDF_SYMMETRY2D, {(float)((mx ? 1 : 0) | (my ? 2 : 0))}, {s.id}, box)};
}
Shader2D Builder::Annulus(Shader2D s, float sub) {  // :603-627
  if (!is2(s.id)) nilsdf("Annulus");
  if (sub <= 0) shapeErrorf("invalid annular parameter");
  Box2 bb = Bounds(s);
  Vec2 e{sub, sub};
  return {push2(GSDF_ANNULUS2D, {sub}, {s.id}, Box2{Sub(bb.Min, e), Add(bb.Max, e)})};
}
Shader2D Builder::CircularArray2D(Shader2D s, int numInstances, int circleDiv) {  // :650-688
  if (!is2(s.id)) nilsdf("circarray2D");
  if (circleDiv <= 1 || numInstances <= 0) shapeErrorf("invalid circarray repeat param");
  if (numInstances > circleDiv) shapeErrorf("bad circular array instances, must be less than or equal to circleDiv");
  Box2 bb = Bounds(s);
  Vec2 verts[4];
  bb.Vertices(verts);
  float angle = (float)(2 * kPi) / (float)circleDiv;
  Mat2 m = RotationMat2(angle);
  for (int i = 0; i < numInstances - 1; i++)
    for (int j = 0; j < 4; j++) { verts[j] = MulMatVec(m, verts[j]); bb = bb.IncludePoint(verts[j]); }
  return {push2(GSDF_CIRCARRAY2D, {(float)numInstances, (float)circleDiv}, {s.id}, bb)};
}
Shader2D Builder::Scale2D(Shader2D s, float scale) {  // :717-732
  if (!is2(s.id)) nilsdf("Scale2D");
  return {push2(GSDF_SCALE2D, {scale}, {s.id}, Bounds(s).Scale(Vec2{scale, scale}))};
}
Shader2D Builder::TranslateMulti2D(Shader2D s, const std::vector<Vec2>& disp) {  // :755-785
  if (!is2(s.id)) nilsdf("TranslateMulti2D");
  Box2 bb{};
  Box2 elem = Bounds(s);
  for (auto& d : disp) bb = bb.Union(elem.Add(d));
  int id = push2(GSDF_TRANSLATEMULTI2D, {}, {s.id}, bb);
  nodes_[(size_t)id].aux_off = (uint32_t)aux_.size();
  nodes_[(size_t)id].aux_len = (uint32_t)(2 * disp.size());
  for (auto& d : disp) { aux_.push_back(d.X); aux_.push_back(d.Y); }
  return {id};
}
Shader2D Builder::Elongate2D(Shader2D s, float x, float y) {  // :812-826
  if (!is2(s.id)) nilsdf("Elongate2D");
  Box2 box = Bounds(s);
  box.Max = MaxElem(box.Max, Vec2{});
  box.Max = Add(box.Max, gsdf::Scale(0.5f, Vec2{x, y}));
  box.Min = gsdf::Scale(-1, box.Max);
  return {push2(GSDF_ELONGATE2D, {x, y}, {s.id}, box)};
}

}  // namespace gsdf
