// threads.hpp -- C++ mirror of the reference's forge/threads package (tree builders for the
// benchmark configs). Reference: /root/reference/forge/threads/threads.go:28-251, iso.go:19-77,
// npt.go:11-74, nut.go:34-80, bolt.go:12-76, hexhead.go:15-48, knurl.go:18-101.
#pragma once
#include <stdexcept>
#include <string>

#include "builder.hpp"
#include "polygon_builder.hpp"

namespace gsdf {
namespace threads {

constexpr float cosd30 = (float)(1.7320508075688772935274463415058723669428052538103806280558069794L / 2);
constexpr float sind30 = 0.5f;

struct Parameters {  // threads.go:33-40
  std::string Name;
  float Radius = 0, Pitch = 0;
  int Starts = 0;
  float Taper = 0, HexF2F = 0;
  float HexRadius() const { return HexF2F / (float)(2.0L * 1.7320508075688772935274463415058723669428052538103806280558069794L / 2); }
  float HexHeight() const { return 2.0f * HexRadius() * (float)(5.0L / 12.0L); }
};

class Threader {
 public:
  virtual ~Threader() = default;
  virtual Shader2D Thread(Builder& bld) const = 0;
  virtual Parameters ThreadParams() const = 0;
};

// metricf2f (threads.go:226-251)
inline float metricf2f(float radius) {
  static const float table[] = {1.75f, 2, 3.2f, 4, 5, 6, 7, 8, 10, 13, 17, 19, 24, 30, 36, 46, 55, 65, 75, 85, 95};
  float est;
  if (radius < (float)(1.2L / 2)) est = 3.2f * radius;
  else if (radius < (float)(3.8L / 2)) est = 4.5f * radius;
  else if (radius < (float)(4.2L / 2)) est = 4.f * radius;
  else est = 3.5f * radius;
  if (absf(radius - (float)(56.L / 2)) < 1) est = 86;
  for (int i = (int)(sizeof(table) / sizeof(table[0])) - 1; i >= 0; i--)
    if (est - 1e-2f > table[i]) return table[i];
  return table[0];
}

struct ISO : Threader {  // iso.go:19-77
  float D = 0, P = 0;
  bool Ext = false;
  ISO() = default;
  ISO(float d, float p, bool ext) : D(d), P(p), Ext(ext) {}
  Parameters ThreadParams() const override {  // basic.ThreadParams threads.go:212-222
    float radius = D / 2;
    return Parameters{"basic", radius, P, 1, 0, metricf2f(radius)};
  }
  Shader2D Thread(Builder& bld) const override {
    const float twoTanTheta = (float)(2.0L * (0.5L / (1.7320508075688772935274463415058723669428052538103806280558069794L / 2)));
    float radius = D / 2;
    float h = P / twoTanTheta;
    float rMajor = radius;
    float r0 = rMajor - (float)(7.0L / 8.0L) * h;
    PolygonBuilder poly;
    if (Ext) {
      float rRoot = (P / 8.0f) / cosd30;
      float xOfs = (float)(1.0L / 16.0L) * P;
      poly.AddXY(P, 0);
      poly.AddXY(P, r0 + h);
      poly.AddXY(P / 2.0f, r0).Smooth(rRoot, 5);
      poly.AddXY(xOfs, rMajor);
      poly.AddXY(-xOfs, rMajor);
      poly.AddXY(-P / 2.0f, r0).Smooth(rRoot, 5);
      poly.AddXY(-P, r0 + h);
      poly.AddXY(-P, 0);
    } else {
      float rMinor = r0 + (float)(1.0L / 4.0L) * h;
      float rCrest = (P / 16.0f) / cosd30;
      float xOfs = (float)(1.0L / 8.0L) * P;
      poly.AddXY(P, 0);
      poly.AddXY(P, rMinor);
      poly.AddXY(P / 2 - xOfs, rMinor);
      poly.AddXY(0, r0 + h).Smooth(rCrest, 5);
      poly.AddXY(-P / 2 + xOfs, rMinor);
      poly.AddXY(-P, rMinor);
      poly.AddXY(-P, 0);
    }
    return bld.NewPolygon(poly.AppendVecs());
  }
};

struct NPT : Threader {  // npt.go:11-74
  float D = 0, TPI = 0, F2F = 0;
  Parameters ThreadParams() const override {
    Parameters p = ISO(D, 1.0f / TPI, false).ThreadParams();
    p.Name = "NPT";
    p.Taper = atanf32((float)(1.0L / 32.0L));
    if (F2F > 0) p.HexF2F = F2F;
    return p;
  }
  Shader2D Thread(Builder& bld) const override { return ISO(D, 1.0f / TPI, false).Thread(bld); }
  void SetFromNominal(float nominal) {
    struct Spec { float N, D, tpi, ftof; };
    static const Spec tbl[] = {
        {(float)(1.0L / 8), 0.405f, 27, (float)(11.2L / 25.4L)},   {(float)(1.0L / 4), 0.540f, 18, (float)(15.7L / 25.4L)},
        {(float)(3.0L / 8), 0.675f, 18, (float)(17.5L / 25.4L)},   {(float)(1.0L / 2), 0.840f, 14, (float)(22.4L / 25.4L)},
        {(float)(3.0L / 4), 1.050f, 14, (float)(26.9L / 25.4L)},   {1.0f, 1.315f, 11.5f, (float)(35.1L / 25.4L)},
        {(float)(1 + 1.0L / 4), 1.660f, 11.5f, (float)(44.5L / 25.4L)}, {(float)(1 + 1.0L / 2), 1.900f, 11.5f, (float)(50.8L / 25.4L)},
        {2, 2.375f, 11.5f, (float)(63.5L / 25.4L)},                {(float)(2 + 1.0L / 2), 2.875f, 8, (float)(76.2L / 25.4L)},
        {3, 3.500f, 8, (float)(88.9L / 25.4L)},                    {4, 4.500f, 8, (float)(117.3L / 25.4L)}};
    const float lookupTol = (float)(1.L / 32.L);
    for (auto& a : tbl)
      if (absf(a.N - nominal) < lookupTol) { D = a.D; F2F = a.ftof; TPI = a.tpi; return; }
    throw std::invalid_argument("nominal measurement not found");
  }
};

// Screw (threads.go:71-94)
inline Shader3D Screw(Builder& bld, float length, const Threader& thread) {
  if (length <= 0) throw std::invalid_argument("need greater than zero length");
  Shader2D tsdf = thread.Thread(bld);
  Parameters p = thread.ThreadParams();
  return bld.NewScrewNode(tsdf, p.Pitch, -p.Pitch * (float)p.Starts, length / 2, p.Taper);
}

// HexHead (hexhead.go:15-48)
inline Shader3D HexHead(Builder& bld, float radius, float height, bool roundNeg, bool roundPos) {
  float cornerRound = radius * 0.08f;
  PolygonBuilder poly;
  poly.Nagon(6, radius - cornerRound);
  Shader2D hex2d = bld.NewPolygon(poly.AppendVecs());
  hex2d = bld.Offset2D(hex2d, -cornerRound);
  Shader3D hex3d = bld.Extrude(hex2d, height);
  if (roundPos || roundNeg) {
    float topRound = radius * 1.6f;
    float d = radius * cosd30;
    Shader3D sphere = bld.NewSphere(topRound);
    float zOfs = std::sqrt(topRound * topRound - d * d) - height / 2;
    if (roundNeg) hex3d = bld.Intersection(hex3d, bld.Translate(sphere, 0, 0, -zOfs));
    if (roundPos) hex3d = bld.Intersection(hex3d, bld.Translate(sphere, 0, 0, zOfs));
  }
  return hex3d;
}

// PlasticButtress (plasticbuttress.go:9-53): screw-top style buttress thread, similar to ANSI 45/7 with more rounding.
// Go evaluates the untyped constant expressions exactly and rounds them once to float32 where they meet a float32
// operand: t0 + t1, threadEngage / 2.0, 0.05, 0.15 are single literals here for that reason.
struct PlasticButtress : Threader {
  float D = 0, P = 0;
  PlasticButtress() = default;
  PlasticButtress(float d, float p) : D(d), P(p) {}
  Parameters ThreadParams() const override {  // basic(butt).ThreadParams() threads.go:212-222
    float radius = D / 2;
    return Parameters{"basic", radius, P, 1, 0, metricf2f(radius)};
  }
  Shader2D Thread(Builder& bld) const override {
    const float radius = D / 2;
    const float t0 = 1.0f;
    const float t1 = (float)0.1227845609029046L;          // math.Tan(7 deg)
    const float t0t1 = (float)(1.0L + 0.1227845609029046L);
    const float p = P;
    const float h0 = p / t0t1;
    const float h1 = ((float)(0.6L / 2.0L) * p) + (0.5f * h0);
    const float hp = p / 2.0f;
    PolygonBuilder tp;
    tp.AddXY(p, 0);
    tp.AddXY(p, radius);
    const float p2 = hp - ((h0 - h1) * t1);
    tp.AddXY(p2, radius).Smooth((float)0.05L * p, 5);
    const float p3 = t0 * h0 - hp;
    tp.AddXY(p3, radius - h1).Smooth((float)0.15L * p, 5);
    const float p4 = (h0 - h1) * t0 - hp;
    tp.AddXY(p4, radius).Smooth((float)0.15L * p, 5);
    tp.AddXY(-p, radius);
    tp.AddXY(-p, 0);
    return bld.NewPolygon(tp.AppendVecs());
  }
};

// Knurl (knurl.go:18-101)
struct KnurlParams : Threader {
  float Length = 0, Radius = 0, Pitch = 0, Height = 0, Theta = 0;
  int starts = 0;
  Shader2D Thread(Builder& bld) const override {
    PolygonBuilder k;
    k.AddXY(Pitch / 2, 0);
    k.AddXY(Pitch / 2, Radius);
    k.AddXY(0, Radius + Height);
    k.AddXY(-Pitch / 2, Radius);
    k.AddXY(-Pitch / 2, 0);
    return bld.NewPolygon(k.AppendVecs());
  }
  Parameters ThreadParams() const override {
    Parameters p = ISO(Radius * 2, Pitch, true).ThreadParams();
    p.Starts = starts;
    return p;
  }
};
inline Shader3D Knurl(Builder& bld, KnurlParams k) {
  if (k.Length <= 0) throw std::invalid_argument("zero or negative Knurl length");
  if (k.Radius <= 0) throw std::invalid_argument("zero or negative Knurl radius");
  if (k.Pitch <= 0) throw std::invalid_argument("zero or negative Knurl pitch");
  if (k.Height <= 0) throw std::invalid_argument("zero or negative Knurl height");
  if (k.Theta < 0) throw std::invalid_argument("zero Knurl helix angle");
  if (k.Theta >= kPiF / 2) throw std::invalid_argument("too large Knurl helix angle");
  k.starts = (int)(2 * kPiF * k.Radius * tanf32(k.Theta) / k.Pitch);
  Shader3D k0 = Screw(bld, k.Length, k);
  k.starts *= -1;
  Shader3D k1 = Screw(bld, k.Length, k);
  return bld.Intersection(k0, k1);
}
inline Shader3D KnurledHead(Builder& bld, float radius, float height, float pitch) {
  float cylinderRound = radius * 0.05f;
  float knurlLength = pitch * std::floor((height - cylinderRound) / pitch);
  KnurlParams k;
  k.Length = knurlLength; k.Radius = radius; k.Pitch = pitch; k.Height = pitch * 0.3f;
  k.Theta = (float)(45.0L * kPi / 180);
  Shader3D knurl = Knurl(bld, k);
  Shader3D cyl = bld.NewCylinder(radius, height, cylinderRound);
  return bld.Union(cyl, knurl);
}

enum NutStyle { NutCircular = 1, NutHex, NutKnurl };  // nut.go:10-17

// Nut (nut.go:41-80)
inline Shader3D Nut(Builder& bld, const Threader& thread, NutStyle style, float tolerance = 0) {
  if (tolerance < 0) throw std::invalid_argument("tolerance < 0");
  Parameters params = thread.ThreadParams();
  float nr = params.HexRadius(), nh = params.HexHeight();
  if (nr <= 0 || nh <= 0) throw std::invalid_argument("bad hex nut dimensions");
  Shader3D nut;
  switch (style) {
    case NutHex: nut = HexHead(bld, nr, nh, true, true); break;
    case NutKnurl: nut = KnurledHead(bld, nr, nh, nr * 0.25f); break;
    case NutCircular: nut = bld.NewCylinder(nr * 1.1f, nh, 0); break;
    default: throw std::invalid_argument("passed argument NutStyle not defined for Nut");
  }
  Shader3D thr = Screw(bld, nh * (float)(1 + 1e-2L), thread);
  return bld.Difference(nut, thr);
}

// Bolt (bolt.go:21-76)
struct BoltParams {
  const Threader* Thread = nullptr;
  NutStyle Style = NutHex;
  float Tolerance = 0, TotalLength = 0, ShankLength = 0;
};
inline Shader3D Bolt(Builder& bld, const BoltParams& k) {
  if (!k.Thread) throw std::invalid_argument("nil Threader");
  if (k.TotalLength < 0) throw std::invalid_argument("total length < 0");
  if (k.ShankLength >= k.TotalLength) throw std::invalid_argument("shank length must be less than total length");
  if (k.ShankLength <= 0) throw std::invalid_argument("shank length <= 0");
  if (k.Tolerance < 0) throw std::invalid_argument("tolerance < 0");
  Parameters param = k.Thread->ThreadParams();
  float hr = param.HexRadius(), hh = param.HexHeight();
  if (hr <= 0 || hh <= 0) throw std::invalid_argument("bad hex head dimension");
  Shader3D head;
  switch (k.Style) {
    case NutHex: head = HexHead(bld, hr, hh, false, true); break;
    case NutKnurl: head = KnurledHead(bld, hr, hh, hr * 0.25f); break;
    default: throw std::invalid_argument("unknown style for bolt");
  }
  float screwLen = k.TotalLength - k.ShankLength;
  Shader3D screw = Screw(bld, screwLen, *k.Thread);
  Shader3D shank = bld.NewCylinder(param.Radius, k.ShankLength, hh * 0.08f);
  float shankOff = k.ShankLength / 2 + hh / 2;
  shank = bld.Translate(shank, 0, 0, shankOff);
  screw = bld.Translate(screw, 0, 0, shankOff + screwLen / 2);
  return bld.Union(screw, bld.SmoothUnion(hh * 0.12f, shank, head));
}

}  // namespace threads

// ---------------------------------------------------------------------------------------------
// Scenes of the benchmark configs (BASELINE.json configs[0..3]).
// ---------------------------------------------------------------------------------------------
namespace scenes {

// examples/npt-flange/flange.go:23-59
inline Shader3D NptFlange(Builder& bld) {
  const float tlen = (float)(18.L / 25.4L);
  const float internalDiameter = (float)(1.5L / 2.L);
  const float flangeH = (float)(7.L / 25.4L);
  (void)tlen; (void)internalDiameter;
  threads::NPT npt;
  npt.SetFromNominal((float)(1.0L / 2.0L));
  Shader3D pipe = threads::Nut(bld, npt, threads::NutCircular);
  Shader3D flange = bld.NewCylinder((float)(60.L / 25.4L / 2), flangeH, (float)(7.L / 25.4L / 8));
  flange = bld.Translate(flange, 0, 0, (float)(-(18.L / 25.4L) / 2));
  Shader3D u = bld.SmoothUnion(0.2f, pipe, flange);
  Shader3D hole = bld.NewCylinder((float)(1.5L / 2.L / 2), (float)(4 * (7.L / 25.4L)), 0);
  u = bld.Difference(u, hole);
  u = bld.Scale(u, 25.4f);
  return u;
}

// examples/bolt/main.go:26-40
inline Shader3D Bolt(Builder& bld) {
  const float L = 8, shank = 3;
  threads::ISO threader(3, 0.5f, true);
  threads::BoltParams bp;
  bp.Thread = &threader; bp.Style = threads::NutHex; bp.TotalLength = L + shank; bp.ShankLength = shank;
  Shader3D M3 = threads::Bolt(bld, bp);
  return bld.Rotate(M3, (float)(2.5L * kPi / 2), Vec3{1, 0, 0.1f});
}

// examples/knurled-cylinder/knurled-cyl.go:57-107 (defaults: d=20, length 0 -> 5r, hole 0 -> r, knurl 0 -> r)
inline Shader3D KnurledCylinder(Builder& bld, float diameter = 20) {
  float r = (float)((double)diameter / 2);
  float length = 5 * r;
  float holeDiam = r;
  float knurlSide = r;
  const float smoothRatio = 0.1f, twistK = 0.75f, knurlOffsetR = 1.6f;
  const int knurlN = 24;
  float sk = smoothRatio * r;
  Shader3D obj = bld.NewCylinder(r, length, smoothRatio * r);
  Shader3D knurlBox = bld.NewBox(knurlSide, knurlSide, length * 0.8f, 0);
  knurlBox = bld.Rotate(knurlBox, kPiF / 4, Vec3{0, 0, 1});
  knurlBox = bld.Translate(knurlBox, knurlOffsetR * r, 0, 0);
  knurlBox = bld.CircularArray(knurlBox, knurlN, knurlN);
  Shader3D knurl = bld.Union(bld.Twist(knurlBox, twistK / r), bld.Twist(knurlBox, -twistK / r));
  obj = bld.SmoothDifference(sk, obj, knurl);
  obj = bld.SmoothDifference(sk, obj, bld.NewCylinder(holeDiam / 2, length + 2 * r, 0));
  Shader3D ventCyl = bld.NewCylinder(0.25f * r, 3 * r, 0);
  ventCyl = bld.Rotate(ventCyl, kPiF / 2, Vec3{0, 1, 0});
  obj = bld.SmoothDifference(sk, obj, bld.Translate(ventCyl, 0, 0, -length / 2));
  obj = bld.SmoothDifference(sk, obj, bld.Translate(ventCyl, 0, 0, length / 2));
  return obj;
}

// examples/fibonacci-showerhead/main.go:30-88,138-148: knurled cap with an internal plastic-buttress thread on a base plate
// drilled with 131 holes on a Fibonacci spiral. The reference's README holds its triangle count at resdiv 350 (309,872).
inline Vec2 fibonacci(int n) {
  const float angleOfDivergence = 137.3f, spacing = 2.6f;
  const float nf = (float)n;
  const float a = nf * angleOfDivergence / 360 * kPiF;
  const float r = spacing * std::sqrt(nf);
  float sa, ca;
  sincosf32(a, sa, ca);
  return Vec2{r * ca, r * sa};
}
inline Shader3D Showerhead(Builder& bld) {
  const float threadExtDiameter = 65.f, threadedLength = 5.f;
  const float threadPitch = (float)(5.0L / 3.0L);  // threadedLength / threadTurns, an exact constant expression in Go
  const float showerheadBaseThick = 2.5f, showerheadWall = 4.f, threadheight = 5.f;
  threads::PlasticButtress showerThread(threadExtDiameter, threadPitch);
  Shader3D knurled = threads::KnurledHead(bld, threadExtDiameter / 2 + showerheadWall, threadheight, 1);
  Shader3D thr = threads::Screw(bld, threadheight + .5f, showerThread);
  Shader3D object = bld.Difference(knurled, thr);
  Shader3D base = bld.NewCylinder(threadExtDiameter / 2 + showerheadWall, showerheadBaseThick, 0);
  base = bld.Translate(base, 0, 0, -(threadedLength / 2 + showerheadBaseThick / 2 - 1));
  Shader3D hole = bld.NewCylinder(0.8f, showerheadBaseThick * 10, 0);
  Shader3D holes = hole;
  for (int i = 0; i < 130; i++) {
    const Vec2 v = fibonacci(i);
    holes = bld.Union(holes, bld.Translate(hole, v.X, v.Y, 0));
  }
  base = bld.Difference(base, holes);
  object = bld.Union(object, base);
  return object;
}

// Synthetic stand-in for BASELINE.json configs[4] (forge/textsdf multi-glyph plate): forge/textsdf needs
// golang.org/x/image/font/sfnt (TTF parsing, 26.6 fixed-point scaling, kerning), which is external to the reference
// and is not restated. The computational shape is reproduced instead: a wide Union2D of translated glyph polygons
// (outlines with holes via Difference2D, like font.go:214-337), extruded and united with a base plate
// (examples/ui-text/uitext.go:30-42 pattern).
inline Shader3D GlyphPlate(Builder& bld, int nglyphs = 24) {
  auto poly = [&](std::initializer_list<Vec2> v) { return bld.NewPolygon(std::vector<Vec2>(v)); };
  // block letters on a 6 x 10 cell
  Shader2D G = poly({{0, 0}, {6, 0}, {6, 5}, {3, 5}, {3, 3.5f}, {4.5f, 3.5f}, {4.5f, 1.5f}, {1.5f, 1.5f}, {1.5f, 8.5f}, {6, 8.5f}, {6, 10}, {0, 10}});
  Shader2D S = poly({{0, 0}, {6, 0}, {6, 5.75f}, {1.5f, 5.75f}, {1.5f, 8.5f}, {6, 8.5f}, {6, 10}, {0, 10}, {0, 4.25f}, {4.5f, 4.25f}, {4.5f, 1.5f}, {0, 1.5f}});
  Shader2D Dout = poly({{0, 0}, {4, 0}, {6, 2}, {6, 8}, {4, 10}, {0, 10}});
  Shader2D Din = poly({{1.5f, 1.5f}, {3.4f, 1.5f}, {4.5f, 2.6f}, {4.5f, 7.4f}, {3.4f, 8.5f}, {1.5f, 8.5f}});
  Shader2D D = bld.Difference2D(Dout, Din);
  Shader2D F = poly({{0, 0}, {1.5f, 0}, {1.5f, 4.25f}, {4.5f, 4.25f}, {4.5f, 5.75f}, {1.5f, 5.75f}, {1.5f, 8.5f}, {6, 8.5f}, {6, 10}, {0, 10}});
  Shader2D glyphs[4] = {G, S, D, F};
  std::vector<Shader2D> line;
  const float advance = 7.5f;
  const int per_row = 12;
  for (int i = 0; i < nglyphs; i++) {
    const float x = advance * (float)(i % per_row), y = -13.0f * (float)(i / per_row);
    line.push_back(bld.Translate2D(glyphs[i % 4], x, y));
  }
  Shader2D text = line.size() == 1 ? line[0] : bld.Union2D(line);
  Shader3D text3 = bld.Extrude(text, 2.0f);
  Box3 tb = bld.Bounds(text3);
  Vec3 sz = tb.Size();
  Shader3D plate = bld.NewBox(sz.X + 4, sz.Y + 4, 1.0f, 0.25f);
  Vec3 c = tb.Center();
  plate = bld.Translate(plate, c.X, c.Y, -1.25f);
  return bld.Union(text3, plate);
}

}  // namespace scenes
}  // namespace gsdf
