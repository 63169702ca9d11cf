// textsdf.hpp -- C++ mirror of the reference's forge/textsdf (font.go:20-337): Font{LoadTTFBytes, Configure, TextLine,
// Glyph, AdvanceWidth, Kern}. Same control flow, error texts and float32 arithmetic as font.go; two dependencies of the
// reference are external to /root/reference and restated (parity unpinned at the value level):
//   * glyph outlines: golang.org/x/image/font/sfnt -> ttf.hpp (raw font units at ppem == UnitsPerEm);
//   * curve flattening: soypat/geometry ms2.Spline3Sampler.SampleBisect(dst, 4) with Tolerance = reltol. Restated here
//     as recursive bisection to depth <= 4 that stops where the curve's midpoint deviates from the chord's midpoint
//     by at most reltol * chord length; it appends interior points only, like font.go:300-316 expects (the segment's
//     start is appended by the caller, its end arrives as the next segment's start).
#pragma once
#include <cmath>
#include <map>
#include <string>
#include <vector>

#include "builder.hpp"
#include "ttf.hpp"

namespace gsdf {
namespace textsdf {

struct FontConfig {
  float RelativeGlyphTolerance = 0;  // 0: a reasonable value (0.15) is chosen (font.go:22,72-74)
};

class Font {
 public:
  explicit Font(Builder& b) : bld_(b) {}
  void Configure(const FontConfig& cfg) {  // font.go:39-49
    if (cfg.RelativeGlyphTolerance < 0 || cfg.RelativeGlyphTolerance >= 1) throw std::invalid_argument("invalid RelativeGlyphTolerance");
    reset();
    reltol_ = cfg.RelativeGlyphTolerance;
    if (reltol_ == 0) reltol_ = 0.15f;
  }
  void LoadTTFBytes(const uint8_t* p, size_t n) {  // font.go:52-60
    ttf::Font f;
    f.Parse(p, n);
    reset();
    sfn_ = f;
    loaded_ = true;
  }
  // TextLine (font.go:86-137): glyphs placed from x = 0 in +x by advance width and kerning.
  Shader2D TextLine(const std::string& utf8) {
    require_font();
    std::vector<Shader2D> shapes;
    unsigned idxPrev = 0;
    long long xOfs = 0;  // fixed.Int26_6 raw value == font units at ppem == UnitsPerEm
    const float scaleout = this->scaleout();
    size_t ic = 0;
    for (size_t pos = 0; pos < utf8.size(); ic++) {
      const size_t at = pos;
      const uint32_t c = decode(utf8, pos);
      if (!is_graphic(c)) throw std::invalid_argument("char " + quote(c) + " not graphic");
      const unsigned idx = sfn_.GlyphIndex(c);
      long long advance = sfn_.Advance(idx);
      if (is_space(c)) {
        if (c == '\t') advance *= 4;
        xOfs += advance;
        continue;
      }
      Shader2D charshape = Glyph(c);
      if (at > 0) xOfs += sfn_.Kern(idxPrev, idx);  // `ic > 0` in the reference: ic is the byte index of the rune
      idxPrev = idx;
      charshape = bld_.Translate2D(charshape, (float)xOfs * scaleout, 0);
      shapes.push_back(charshape);
      xOfs += advance;
    }
    if (shapes.size() == 1) return shapes[0];
    if (shapes.empty()) throw std::invalid_argument("no text provided");
    return bld_.Union2D(shapes);
  }
  float Kern(uint32_t c0, uint32_t c1) { require_font(); return (float)sfn_.Kern(sfn_.GlyphIndex(c0), sfn_.GlyphIndex(c1)) * scaleout(); }
  float AdvanceWidth(uint32_t c) { require_font(); return (float)sfn_.Advance(sfn_.GlyphIndex(c)) * scaleout(); }
  Shader2D Glyph(uint32_t c) {  // font.go:155-190: cached per rune
    require_font();
    auto it = glyphs_.find(c);
    if (it != glyphs_.end()) return it->second;
    Shader2D g = makeGlyph(c);
    glyphs_[c] = g;
    return g;
  }
  float scaleout() const {  // font.go:207-212: 1 / min(size of the font bounding box)
    const int* bb = sfn_.BBox();
    const float w = (float)bb[2] - (float)bb[0], h = (float)bb[3] - (float)bb[1];
    const float sz = w < h ? w : h;
    return 1.f / sz;
  }

 private:
  Builder& bld_;
  ttf::Font sfn_;
  bool loaded_ = false;
  float reltol_ = 0.15f;
  std::map<uint32_t, Shader2D> glyphs_;

  void reset() { glyphs_.clear(); if (reltol_ == 0) reltol_ = 0.15f; }
  void require_font() const { if (!loaded_) throw std::invalid_argument("textsdf: no font loaded"); }
  static bool is_space(uint32_t c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f' || c == 0x85 || c == 0xA0; }
  static bool is_graphic(uint32_t c) {  // unicode.IsGraphic restricted to what matters here: no controls
    if (c == ' ') return true;
    return !(c < 0x20 || (c >= 0x7f && c < 0xa0));
  }
  static std::string quote(uint32_t c) { char b[16]; snprintf(b, sizeof b, c < 0x80 && c >= 0x20 ? "'%c'" : "'\\\\u%04x'", (unsigned)c); return b; }
  static uint32_t decode(const std::string& s, size_t& i) {
    const unsigned char b0 = (unsigned char)s[i++];
    if (b0 < 0x80) return b0;
    int n = b0 >= 0xf0 ? 3 : (b0 >= 0xe0 ? 2 : 1);
    uint32_t c = b0 & (0x3f >> n);
    while (n-- && i < s.size()) c = (c << 6) | ((unsigned char)s[i++] & 0x3f);
    return c;
  }
  static Vec2 fixedToVec(int x, int y, float scale) { return Vec2{(float)x * scale, (float)y * scale}; }  // font.go:331-336 (y up again)

  Shader2D makeGlyph(uint32_t ch) {  // font.go:214-257
    const unsigned idx = sfn_.GlyphIndex(ch);
    const std::vector<ttf::Segment> segments = sfn_.LoadGlyph(idx);
    const float scaleout = this->scaleout(), tol = reltol_;
    std::vector<std::vector<ttf::Segment>> contours;  // splitContours: each MoveTo starts a contour
    for (const ttf::Segment& s : segments) {
      if (s.op == ttf::Segment::MoveTo || contours.empty()) contours.emplace_back();
      contours.back().push_back(s);
    }
    if (contours.empty()) throw std::invalid_argument("glyph has no contours");
    bool fill = false;
    Shader2D shape = segmentsToPolygon(contours[0], tol, scaleout, &fill);
    for (size_t k = 1; k < contours.size(); k++) {
      Shader2D sdf = segmentsToPolygon(contours[k], tol, scaleout, &fill);
      shape = fill ? bld_.Union2D(std::vector<Shader2D>{shape, sdf}) : bld_.Difference2D(shape, sdf);
    }
    return shape;
  }

  // quadratic Bezier through (a, ctrl, b) at t
  static Vec2 quadAt(Vec2 a, Vec2 c, Vec2 b, float t) {
    const float u = 1.f - t;
    return Vec2{u * u * a.X + 2.f * u * t * c.X + t * t * b.X, u * u * a.Y + 2.f * u * t * c.Y + t * t * b.Y};
  }
  static void bisect(std::vector<Vec2>& dst, Vec2 a, Vec2 c, Vec2 b, float t0, float t1, Vec2 p0, Vec2 p1, float tol, int depth) {
    if (depth <= 0) return;
    const float tm = 0.5f * (t0 + t1);
    const Vec2 pm = quadAt(a, c, b, tm);
    const float mx = 0.5f * (p0.X + p1.X), my = 0.5f * (p0.Y + p1.Y);
    const float dev = std::hypot(pm.X - mx, pm.Y - my), chord = std::hypot(p1.X - p0.X, p1.Y - p0.Y);
    if (dev <= tol * chord) return;
    bisect(dst, a, c, b, t0, tm, p0, pm, tol, depth - 1);
    dst.push_back(pm);
    bisect(dst, a, c, b, tm, t1, pm, p1, tol, depth - 1);
  }

  Shader2D segmentsToPolygon(const std::vector<ttf::Segment>& segs, float tol, float scale, bool* fill) {  // font.go:276-329
    std::vector<Vec2> poly;
    float windingSum = 0;
    Vec2 prev{0, 0};
    for (const ttf::Segment& s : segs) {
      switch (s.op) {
        case ttf::Segment::MoveTo: prev = fixedToVec(s.x[0], s.y[0], scale); break;
        case ttf::Segment::LineTo: {
          const Vec2 p = fixedToVec(s.x[0], s.y[0], scale);
          poly.push_back(prev);
          windingSum += (prev.X - p.X) * (prev.Y + p.Y);
          prev = p;
          break;
        }
        case ttf::Segment::QuadTo: {
          const Vec2 ctrl = fixedToVec(s.x[0], s.y[0], scale), end = fixedToVec(s.x[1], s.y[1], scale);
          poly.push_back(prev);
          bisect(poly, prev, ctrl, end, 0.f, 1.f, prev, end, tol, 4);
          windingSum += (prev.X - end.X) * (prev.Y + end.Y);
          prev = end;
          break;
        }
      }
    }
    *fill = windingSum < 0;
    const size_t nerr = bld_.Errs().size();
    Shader2D sdf = bld_.NewPolygon(poly);
    if (bld_.Errs().size() > nerr) throw std::invalid_argument(bld_.Errs().back());  // `bld.Err()` at font.go:328
    return sdf;
  }
};

}  // namespace textsdf
}  // namespace gsdf
