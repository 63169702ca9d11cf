// ms.hpp -- host-side float32 geometry helpers used while BUILDING trees (bounds, matrices,
// polygon vertices). Mirrors the subset of github.com/soypat/geometry (ms1/ms2/ms3, un-vendored,
// go.mod:11) and github.com/chewxy/math32 (go.mod:8) that the reference's constructors and Bounds()
// methods call (operations.go, operations2d.go, primitives*.go, forge/threads/*.go). Those modules
// are not under /root/reference, so semantics marked [external] are restated from their published
// API (sdfx / gonum r3 lineage); see DESIGN.md "externals".
//
// Compile with -ffp-contract=off: Go on amd64 never fuses a*b+c.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../gsdf_amd/csrc/host_math.h"

namespace gsdf {

// ---- math32 subset (host only; builder-time scalars) ----
inline float absf(float x) { return std::fabs(x); }
inline float minf(float a, float b) {  // math.Min semantics
  if (a != a || b != b) return NAN;
  if (a == 0 && a == b) return std::signbit(a) ? a : b;
  return a < b ? a : b;
}
inline float maxf(float a, float b) {
  if (a != a || b != b) return NAN;
  if (a == 0 && a == b) return std::signbit(a) ? b : a;
  return a > b ? a : b;
}
inline float hypotf32(float p, float q) {  // math32.Hypot (float32 port of go/src/math/hypot.go)
  if (std::isinf(p) || std::isinf(q)) return INFINITY;
  if (p != p || q != q) return NAN;
  p = absf(p); q = absf(q);
  if (p < q) std::swap(p, q);
  if (p == 0) return 0;
  q = q / p;
  return p * std::sqrt(1 + q * q);
}
// math32.Sincos, math32.Tan, math32.Log2, kPi: ../gsdf_amd/csrc/host_math.h (the lowering and the meshers need them too)
// float32(math.X(float64)) wrappers (math32 doc.go); libm double is within 1 ulp(double) of Go's.
inline float atanf32(float x) { return (float)std::atan((double)x); }
inline float cosf32(float x) { return (float)std::cos((double)x); }
inline float acosf32(float x) { return (float)std::acos((double)x); }

// ---- ms2 / ms3 ----
struct Vec2 { float X = 0, Y = 0; };
struct Vec3 { float X = 0, Y = 0, Z = 0; };
inline bool operator==(Vec2 a, Vec2 b) { return a.X == b.X && a.Y == b.Y; }
inline bool operator==(Vec3 a, Vec3 b) { return a.X == b.X && a.Y == b.Y && a.Z == b.Z; }
inline Vec2 Add(Vec2 a, Vec2 b) { return {a.X + b.X, a.Y + b.Y}; }
inline Vec2 Sub(Vec2 a, Vec2 b) { return {a.X - b.X, a.Y - b.Y}; }
inline Vec2 Scale(float f, Vec2 a) { return {f * a.X, f * a.Y}; }
inline Vec2 AddScalar(float f, Vec2 a) { return {a.X + f, a.Y + f}; }
inline Vec2 MulElem(Vec2 a, Vec2 b) { return {a.X * b.X, a.Y * b.Y}; }
inline Vec2 DivElem(Vec2 a, Vec2 b) { return {a.X / b.X, a.Y / b.Y}; }
inline Vec2 MinElem(Vec2 a, Vec2 b) { return {minf(a.X, b.X), minf(a.Y, b.Y)}; }
inline Vec2 MaxElem(Vec2 a, Vec2 b) { return {maxf(a.X, b.X), maxf(a.Y, b.Y)}; }
inline float Dot(Vec2 a, Vec2 b) { return a.X * b.X + a.Y * b.Y; }
inline float Cross(Vec2 a, Vec2 b) { return a.X * b.Y - a.Y * b.X; }
inline float Norm(Vec2 a) { return hypotf32(a.X, a.Y); }
inline Vec2 Unit(Vec2 a) { return Scale(1 / Norm(a), a); }
inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
inline Vec2 ClampElem(Vec2 v, Vec2 lo, Vec2 hi) { return {clampf(v.X, lo.X, hi.X), clampf(v.Y, lo.Y, hi.Y)}; }

inline Vec3 Add(Vec3 a, Vec3 b) { return {a.X + b.X, a.Y + b.Y, a.Z + b.Z}; }
inline Vec3 Sub(Vec3 a, Vec3 b) { return {a.X - b.X, a.Y - b.Y, a.Z - b.Z}; }
inline Vec3 Scale(float f, Vec3 a) { return {f * a.X, f * a.Y, f * a.Z}; }
inline Vec3 AddScalar(float f, Vec3 a) { return {a.X + f, a.Y + f, a.Z + f}; }
inline Vec3 MulElem(Vec3 a, Vec3 b) { return {a.X * b.X, a.Y * b.Y, a.Z * b.Z}; }
inline Vec3 MinElem(Vec3 a, Vec3 b) { return {minf(a.X, b.X), minf(a.Y, b.Y), minf(a.Z, b.Z)}; }
inline Vec3 MaxElem(Vec3 a, Vec3 b) { return {maxf(a.X, b.X), maxf(a.Y, b.Y), maxf(a.Z, b.Z)}; }
inline float Norm(Vec3 a) { return hypotf32(a.X, hypotf32(a.Y, a.Z)); }
inline Vec3 Unit(Vec3 a) { return Scale(1 / Norm(a), a); }
inline float MaxOf(Vec3 a) { return maxf(a.X, maxf(a.Y, a.Z)); }
inline float MinOf(Vec3 a) { return minf(a.X, minf(a.Y, a.Z)); }

struct Box2 {
  Vec2 Min, Max;
  Vec2 Size() const { return Sub(Max, Min); }
  Vec2 Center() const { return gsdf::Scale(0.5f, gsdf::Add(Min, Max)); }
  bool Empty() const { return Min.X >= Max.X || Min.Y >= Max.Y; }
  Box2 Union(Box2 b) const {
    if (Empty()) return b;
    if (b.Empty()) return *this;
    return {MinElem(Min, b.Min), MaxElem(Max, b.Max)};
  }
  Box2 Intersect(Box2 b) const {
    Box2 r{MaxElem(Min, b.Min), MinElem(Max, b.Max)};
    if (r.Empty()) return Box2{};
    return r;
  }
  Box2 Add(Vec2 v) const { return {gsdf::Add(Min, v), gsdf::Add(Max, v)}; }
  Box2 Canon() const { return {MinElem(Min, Max), MaxElem(Min, Max)}; }
  Box2 Scale(Vec2 s) const { return Box2{MulElem(Min, s), MulElem(Max, s)}.Canon(); }
  Box2 IncludePoint(Vec2 p) const { return {MinElem(Min, p), MaxElem(Max, p)}; }
  void Vertices(Vec2 out[4]) const {
    out[0] = Min; out[1] = {Max.X, Min.Y}; out[2] = Max; out[3] = {Min.X, Max.Y};
  }
};
inline Box2 NewBox2(float x0, float y0, float x1, float y1) {
  return Box2{{minf(x0, x1), minf(y0, y1)}, {maxf(x0, x1), maxf(y0, y1)}};
}

struct Box3 {
  Vec3 Min, Max;
  Vec3 Size() const { return Sub(Max, Min); }
  Vec3 Center() const { return gsdf::Scale(0.5f, gsdf::Add(Min, Max)); }
  float Diagonal() const { return Norm(Size()); }
  bool Empty() const { return Min.X >= Max.X || Min.Y >= Max.Y || Min.Z >= Max.Z; }
  Box3 Union(Box3 b) const {
    if (Empty()) return b;
    if (b.Empty()) return *this;
    return {MinElem(Min, b.Min), MaxElem(Max, b.Max)};
  }
  Box3 Intersect(Box3 b) const {
    Box3 r{MaxElem(Min, b.Min), MinElem(Max, b.Max)};
    if (r.Empty()) return Box3{};
    return r;
  }
  Box3 Add(Vec3 v) const { return {gsdf::Add(Min, v), gsdf::Add(Max, v)}; }
  Box3 Canon() const { return {MinElem(Min, Max), MaxElem(Min, Max)}; }
  Box3 Scale(Vec3 s) const { return Box3{MulElem(Min, s), MulElem(Max, s)}.Canon(); }
  void Vertices(Vec3 o[8]) const {
    o[0] = Min; o[1] = {Max.X, Min.Y, Min.Z}; o[2] = {Max.X, Max.Y, Min.Z}; o[3] = {Min.X, Max.Y, Min.Z};
    o[4] = {Min.X, Min.Y, Max.Z}; o[5] = {Max.X, Min.Y, Max.Z}; o[6] = Max; o[7] = {Min.X, Max.Y, Max.Z};
  }
};
inline Box3 NewCenteredBox(Vec3 center, Vec3 size) {
  size = MaxElem(size, Vec3{});
  Vec3 half = Scale(0.5f, size);
  return {Sub(center, half), Add(center, half)};
}
inline Box3 ScaleCentered(Box3 a, Vec3 scale) {
  scale = MaxElem(scale, Vec3{});
  return NewCenteredBox(a.Center(), MulElem(scale, a.Size()));
}

// ---- Mat2 ----
struct Mat2 {
  float x00 = 0, x01 = 0, x10 = 0, x11 = 0;
  float Determinant() const { return x00 * x11 - x01 * x10; }
  Mat2 Inverse() const {
    float d = Determinant();
    float id = 1 / d;
    return {x11 * id, -x01 * id, -x10 * id, x00 * id};
  }
};
inline Mat2 RotationMat2(float a) {
  float s, c;
  sincosf32(a, s, c);
  return {c, -s, s, c};
}
inline Vec2 MulMatVec(Mat2 m, Vec2 v) { return {m.x00 * v.X + m.x01 * v.Y, m.x10 * v.X + m.x11 * v.Y}; }
inline Vec2 MulMatVecTrans(Mat2 m, Vec2 v) { return {m.x00 * v.X + m.x10 * v.Y, m.x01 * v.X + m.x11 * v.Y}; }

// ---- Mat4 (row-major x00..x33) ----
struct Mat4 {
  float m[16] = {0};
  float at(int r, int c) const { return m[4 * r + c]; }
  float Determinant() const;
  Mat4 Inverse() const;
  Vec3 MulPosition(Vec3 v) const {
    return {m[0] * v.X + m[1] * v.Y + m[2] * v.Z + m[3], m[4] * v.X + m[5] * v.Y + m[6] * v.Z + m[7],
            m[8] * v.X + m[9] * v.Y + m[10] * v.Z + m[11]};
  }
  Box3 MulBox(Box3 box) const;
};
inline Mat4 RotationMat4(float angle, Vec3 axis) {
  Vec3 v = Unit(axis);
  float s, c;
  sincosf32(angle, s, c);
  float k = 1 - c;
  Mat4 r;
  float a[16] = {k * v.X * v.X + c,       k * v.X * v.Y - v.Z * s, k * v.Z * v.X + v.Y * s, 0,
                 k * v.X * v.Y + v.Z * s, k * v.Y * v.Y + c,       k * v.Y * v.Z - v.X * s, 0,
                 k * v.Z * v.X - v.Y * s, k * v.Y * v.Z + v.X * s, k * v.Z * v.Z + c,       0,
                 0, 0, 0, 1};
  std::memcpy(r.m, a, sizeof(a));
  return r;
}
inline float Mat4::Determinant() const {
  const float x00 = m[0], x01 = m[1], x02 = m[2], x03 = m[3], x10 = m[4], x11 = m[5], x12 = m[6], x13 = m[7];
  const float x20 = m[8], x21 = m[9], x22 = m[10], x23 = m[11], x30 = m[12], x31 = m[13], x32 = m[14], x33 = m[15];
  return x03 * x12 * x21 * x30 - x02 * x13 * x21 * x30 - x03 * x11 * x22 * x30 + x01 * x13 * x22 * x30 +
         x02 * x11 * x23 * x30 - x01 * x12 * x23 * x30 - x03 * x12 * x20 * x31 + x02 * x13 * x20 * x31 +
         x03 * x10 * x22 * x31 - x00 * x13 * x22 * x31 - x02 * x10 * x23 * x31 + x00 * x12 * x23 * x31 +
         x03 * x11 * x20 * x32 - x01 * x13 * x20 * x32 - x03 * x10 * x21 * x32 + x00 * x13 * x21 * x32 +
         x01 * x10 * x23 * x32 - x00 * x11 * x23 * x32 - x02 * x11 * x20 * x33 + x01 * x12 * x20 * x33 +
         x02 * x10 * x21 * x33 - x00 * x12 * x21 * x33 - x01 * x10 * x22 * x33 + x00 * x11 * x22 * x33;
}
inline Mat4 Mat4::Inverse() const {
  const float x00 = m[0], x01 = m[1], x02 = m[2], x03 = m[3], x10 = m[4], x11 = m[5], x12 = m[6], x13 = m[7];
  const float x20 = m[8], x21 = m[9], x22 = m[10], x23 = m[11], x30 = m[12], x31 = m[13], x32 = m[14], x33 = m[15];
  Mat4 r;
  float* o = r.m;
  o[0] = x12 * x23 * x31 - x13 * x22 * x31 + x13 * x21 * x32 - x11 * x23 * x32 - x12 * x21 * x33 + x11 * x22 * x33;
  o[1] = x03 * x22 * x31 - x02 * x23 * x31 - x03 * x21 * x32 + x01 * x23 * x32 + x02 * x21 * x33 - x01 * x22 * x33;
  o[2] = x02 * x13 * x31 - x03 * x12 * x31 + x03 * x11 * x32 - x01 * x13 * x32 - x02 * x11 * x33 + x01 * x12 * x33;
  o[3] = x03 * x12 * x21 - x02 * x13 * x21 - x03 * x11 * x22 + x01 * x13 * x22 + x02 * x11 * x23 - x01 * x12 * x23;
  o[4] = x13 * x22 * x30 - x12 * x23 * x30 - x13 * x20 * x32 + x10 * x23 * x32 + x12 * x20 * x33 - x10 * x22 * x33;
  o[5] = x02 * x23 * x30 - x03 * x22 * x30 + x03 * x20 * x32 - x00 * x23 * x32 - x02 * x20 * x33 + x00 * x22 * x33;
  o[6] = x03 * x12 * x30 - x02 * x13 * x30 - x03 * x10 * x32 + x00 * x13 * x32 + x02 * x10 * x33 - x00 * x12 * x33;
  o[7] = x02 * x13 * x20 - x03 * x12 * x20 + x03 * x10 * x22 - x00 * x13 * x22 - x02 * x10 * x23 + x00 * x12 * x23;
  o[8] = x11 * x23 * x30 - x13 * x21 * x30 + x13 * x20 * x31 - x10 * x23 * x31 - x11 * x20 * x33 + x10 * x21 * x33;
  o[9] = x03 * x21 * x30 - x01 * x23 * x30 - x03 * x20 * x31 + x00 * x23 * x31 + x01 * x20 * x33 - x00 * x21 * x33;
  o[10] = x01 * x13 * x30 - x03 * x11 * x30 + x03 * x10 * x31 - x00 * x13 * x31 - x01 * x10 * x33 + x00 * x11 * x33;
  o[11] = x03 * x11 * x20 - x01 * x13 * x20 - x03 * x10 * x21 + x00 * x13 * x21 + x01 * x10 * x23 - x00 * x11 * x23;
  o[12] = x12 * x21 * x30 - x11 * x22 * x30 - x12 * x20 * x31 + x10 * x22 * x31 + x11 * x20 * x32 - x10 * x21 * x32;
  o[13] = x01 * x22 * x30 - x02 * x21 * x30 + x02 * x20 * x31 - x00 * x22 * x31 - x01 * x20 * x32 + x00 * x21 * x32;
  o[14] = x02 * x11 * x30 - x01 * x12 * x30 - x02 * x10 * x31 + x00 * x12 * x31 + x01 * x10 * x32 - x00 * x11 * x32;
  o[15] = x01 * x12 * x20 - x02 * x11 * x20 + x02 * x10 * x21 - x00 * x12 * x21 - x01 * x10 * x22 + x00 * x11 * x22;
  float id = 1 / Determinant();
  for (int i = 0; i < 16; i++) o[i] *= id;
  return r;
}
inline Box3 Mat4::MulBox(Box3 box) const {
  Vec3 r{m[0], m[4], m[8]}, u{m[1], m[5], m[9]}, b{m[2], m[6], m[10]}, t{m[3], m[7], m[11]};
  Vec3 xa = Scale(box.Min.X, r), xb = Scale(box.Max.X, r);
  Vec3 ya = Scale(box.Min.Y, u), yb = Scale(box.Max.Y, u);
  Vec3 za = Scale(box.Min.Z, b), zb = Scale(box.Max.Z, b);
  Vec3 xmn = MinElem(xa, xb), xmx = MaxElem(xa, xb);
  Vec3 ymn = MinElem(ya, yb), ymx = MaxElem(ya, yb);
  Vec3 zmn = MinElem(za, zb), zmx = MaxElem(za, zb);
  return {Add(Add(Add(xmn, ymn), zmn), t), Add(Add(Add(xmx, ymx), zmx), t)};
}

}  // namespace gsdf
