// ttf.hpp -- minimal TrueType reader for forge/textsdf: cmap (format 4 / 12), head, maxp, hhea/hmtx, loca, glyf
// (simple glyphs and offset-only composites), kern format 0.
//
// The reference gets glyph outlines from golang.org/x/image/font/sfnt v0.22.0 (go.mod), which is NOT in
// /root/reference: this is a restatement of what textsdf asks of it (font.go:86-255) for the one call pattern it uses --
// LoadGlyph / GlyphAdvance / Kern / Bounds at ppem == UnitsPerEm with HintingNone, where sfnt's fixed-point scaling
// `(ppem*x +- upem/2) / upem` is the identity, so every coordinate is the raw font unit. Contours become MoveTo /
// LineTo / QuadTo segments the TrueType way (implied on-curve points at the integer midpoint, truncating toward zero
// like Go's `/`, of two consecutive off-curve points; a contour that starts off-curve starts at its last point if that
// is on-curve, else at the midpoint of first and last), closed explicitly. Parity unpinned at the value level.
#pragma once
#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace gsdf {
namespace ttf {

struct Segment {
  enum Op { MoveTo, LineTo, QuadTo } op;
  int x[2], y[2];  // LineTo/MoveTo: [0]; QuadTo: [0] control, [1] end. Font units, y up.
};

class Font {
 public:
  void Parse(const uint8_t* p, size_t n) {
    d_.assign(p, p + n);
    if (n < 12) throw std::runtime_error("sfnt: invalid font");
    const uint32_t ver = u32(0);
    if (ver != 0x00010000u && ver != 0x74727565u /* 'true' */) throw std::runtime_error("sfnt: unsupported font format (TrueType outlines only)");
    const unsigned nt = u16(4);
    for (unsigned i = 0; i < nt; i++) {
      const size_t r = 12 + 16 * (size_t)i;
      need(r, 16);
      std::string tag((const char*)&d_[r], 4);
      tab_[tag] = {u32(r + 8), u32(r + 12)};
    }
    const auto head = table("head", 54);
    upem_ = u16(head.off + 18);
    if (upem_ == 0) throw std::runtime_error("sfnt: invalid head table");
    for (int k = 0; k < 4; k++) bbox_[k] = i16(head.off + 36 + 2 * k);
    loca_long_ = i16(head.off + 50) != 0;
    nglyph_ = u16(table("maxp", 6).off + 4);
    nhm_ = u16(table("hhea", 36).off + 34);
    hmtx_ = table("hmtx", 4 * (size_t)(nhm_ ? nhm_ : 1));
    loca_ = table("loca", (size_t)(nglyph_ + 1) * (loca_long_ ? 4 : 2));
    glyf_ = table("glyf", 0);
    parse_cmap();
    if (tab_.count("kern")) parse_kern();
  }
  unsigned UnitsPerEm() const { return upem_; }
  unsigned NumGlyphs() const { return nglyph_; }
  // font bounding box in font units, y up: xMin yMin xMax yMax
  const int* BBox() const { return bbox_; }
  // 0 = .notdef (sfnt.GlyphIndex returns 0, nil for unmapped runes)
  unsigned GlyphIndex(uint32_t rune) const {
    for (const Seg& s : cmap4_) {
      if (rune > s.end) continue;
      if (rune < s.start) return 0;
      if (s.range_off == 0) return (rune + s.delta) & 0xffffu;
      const size_t a = s.range_addr + s.range_off + 2 * (size_t)(rune - s.start);
      if (a + 2 > d_.size()) return 0;
      unsigned g = u16(a);
      return g ? ((g + s.delta) & 0xffffu) : 0;
    }
    for (const Grp& g : cmap12_) if (rune >= g.start && rune <= g.end) return g.gid + (rune - g.start);
    return 0;
  }
  int Advance(unsigned gid) const {
    if (gid >= nglyph_) throw std::runtime_error("sfnt: glyph index out of range");
    const unsigned k = gid < nhm_ ? gid : (nhm_ ? nhm_ - 1 : 0);
    return (int)u16(hmtx_.off + 4 * (size_t)k);
  }
  int Kern(unsigned a, unsigned b) const {
    auto it = kern_.find(((uint32_t)a << 16) | b);
    return it == kern_.end() ? 0 : it->second;
  }
  std::vector<Segment> LoadGlyph(unsigned gid) const {
    std::vector<Segment> out;
    load(gid, 0, 0, out, 0);
    return out;
  }

 private:
  struct Tab { size_t off = 0, len = 0; };
  struct Seg { uint32_t end, start, delta, range_off; size_t range_addr; };
  struct Grp { uint32_t start, end, gid; };
  std::vector<uint8_t> d_;
  std::map<std::string, Tab> tab_;
  unsigned upem_ = 0, nglyph_ = 0, nhm_ = 0;
  int bbox_[4] = {0, 0, 0, 0};
  bool loca_long_ = false;
  Tab hmtx_, loca_, glyf_;
  std::vector<Seg> cmap4_;
  std::vector<Grp> cmap12_;
  std::map<uint32_t, int> kern_;

  void need(size_t off, size_t n) const { if (off + n > d_.size() || off + n < off) throw std::runtime_error("sfnt: invalid font (truncated)"); }
  uint32_t u16(size_t o) const { need(o, 2); return ((uint32_t)d_[o] << 8) | d_[o + 1]; }
  int i16(size_t o) const { return (int)(int16_t)u16(o); }
  uint32_t u32(size_t o) const { need(o, 4); return ((uint32_t)d_[o] << 24) | ((uint32_t)d_[o + 1] << 16) | ((uint32_t)d_[o + 2] << 8) | d_[o + 3]; }
  Tab table(const std::string& tag, size_t minlen) const {
    auto it = tab_.find(tag);
    if (it == tab_.end()) throw std::runtime_error("sfnt: missing table " + tag);
    need(it->second.off, it->second.len);
    if (it->second.len < minlen) throw std::runtime_error("sfnt: short table " + tag);
    return it->second;
  }
  void parse_cmap() {
    const Tab c = table("cmap", 4);
    const unsigned n = u16(c.off + 2);
    size_t best4 = 0, best12 = 0;
    for (unsigned i = 0; i < n; i++) {
      const size_t r = c.off + 4 + 8 * (size_t)i;
      const unsigned pid = u16(r), eid = u16(r + 2);
      const size_t sub = c.off + u32(r + 4);
      const unsigned fmt = u16(sub);
      const bool unicode = pid == 0 || (pid == 3 && (eid == 1 || eid == 10));
      if (!unicode) continue;
      if (fmt == 4 && !best4) best4 = sub;
      if (fmt == 12 && !best12) best12 = sub;
    }
    if (best12) {
      const uint32_t ng = u32(best12 + 12);
      for (uint32_t g = 0; g < ng; g++) cmap12_.push_back({u32(best12 + 16 + 12 * (size_t)g), u32(best12 + 20 + 12 * (size_t)g), u32(best12 + 24 + 12 * (size_t)g)});
    } else if (best4) {
      const unsigned sc = u16(best4 + 6) / 2;
      const size_t endc = best4 + 14, startc = endc + 2 * (size_t)sc + 2, delta = startc + 2 * (size_t)sc, roff = delta + 2 * (size_t)sc;
      for (unsigned s = 0; s < sc; s++)
        cmap4_.push_back({u16(endc + 2 * (size_t)s), u16(startc + 2 * (size_t)s), u16(delta + 2 * (size_t)s), u16(roff + 2 * (size_t)s), roff + 2 * (size_t)s});
    } else {
      throw std::runtime_error("sfnt: unsupported cmap encoding");
    }
  }
  void parse_kern() {
    const Tab k = tab_.at("kern");
    if (k.len < 4 || u16(k.off) != 0) return;
    const unsigned nt = u16(k.off + 2);
    size_t o = k.off + 4;
    for (unsigned t = 0; t < nt; t++) {
      const unsigned len = u16(o + 2), cov = u16(o + 4);
      if ((cov >> 8) == 0 && (cov & 1)) {  // format 0, horizontal
        const unsigned np = u16(o + 6);
        for (unsigned p = 0; p < np; p++) {
          const size_t e = o + 14 + 6 * (size_t)p;
          kern_[((uint32_t)u16(e) << 16) | u16(e + 2)] = i16(e + 4);
        }
      }
      o += len;
    }
  }
  size_t glyph_off(unsigned gid, size_t* len) const {
    const size_t a = loca_long_ ? u32(loca_.off + 4 * (size_t)gid) : 2 * (size_t)u16(loca_.off + 2 * (size_t)gid);
    const size_t b = loca_long_ ? u32(loca_.off + 4 * (size_t)gid + 4) : 2 * (size_t)u16(loca_.off + 2 * (size_t)gid + 2);
    if (b < a || b > glyf_.len) throw std::runtime_error("sfnt: invalid loca table");
    *len = b - a;
    return glyf_.off + a;
  }
  static int mid(int a, int b) { return (a + b) / 2; }  // Go integer division: truncates toward zero
  void load(unsigned gid, int dx, int dy, std::vector<Segment>& out, int depth) const {
    if (gid >= nglyph_) throw std::runtime_error("sfnt: glyph index out of range");
    if (depth > 8) throw std::runtime_error("sfnt: compound glyph recursion too deep");
    size_t len = 0;
    const size_t g = glyph_off(gid, &len);
    if (len == 0) return;  // empty glyph (space)
    const int nc = i16(g);
    if (nc < 0) {  // compound glyph: offset-only components
      size_t o = g + 10;
      for (;;) {
        const unsigned fl = u16(o), comp = u16(o + 2);
        o += 4;
        int ax, ay;
        if (fl & 0x0001) { ax = i16(o); ay = i16(o + 2); o += 4; } else { ax = (int8_t)d_.at(o); ay = (int8_t)d_.at(o + 1); o += 2; }
        if (!(fl & 0x0002)) throw std::runtime_error("sfnt: compound glyph with point-matching arguments is not supported");
        if (fl & (0x0008 | 0x0040 | 0x0080)) throw std::runtime_error("sfnt: compound glyph with a scale/transform is not supported");
        load(comp, dx + ax, dy + ay, out, depth + 1);
        if (!(fl & 0x0020)) break;
      }
      return;
    }
    std::vector<unsigned> ends((size_t)nc);
    size_t o = g + 10;
    for (int c = 0; c < nc; c++, o += 2) ends[(size_t)c] = u16(o);
    const unsigned npts = nc ? ends.back() + 1 : 0;
    o += 2 + u16(o);  // instructions
    std::vector<uint8_t> fl(npts);
    for (unsigned i = 0; i < npts;) {
      const uint8_t f = d_.at(o++);
      fl[i++] = f;
      if (f & 8) { unsigned rep = d_.at(o++); while (rep-- && i < npts) fl[i++] = f; }
    }
    std::vector<int> xs(npts), ys(npts);
    int v = 0;
    for (unsigned i = 0; i < npts; i++) {
      if (fl[i] & 2) { const int b = d_.at(o++); v += (fl[i] & 16) ? b : -b; } else if (!(fl[i] & 16)) { v += i16(o); o += 2; }
      xs[i] = v + dx;
    }
    v = 0;
    for (unsigned i = 0; i < npts; i++) {
      if (fl[i] & 4) { const int b = d_.at(o++); v += (fl[i] & 32) ? b : -b; } else if (!(fl[i] & 32)) { v += i16(o); o += 2; }
      ys[i] = v + dy;
    }
    unsigned first = 0;
    for (int c = 0; c < nc; c++) {
      const unsigned last = ends[(size_t)c];
      if (last < first || last >= npts) throw std::runtime_error("sfnt: invalid glyf contour");
      contour(&xs[first], &ys[first], &fl[first], last - first + 1, out);
      first = last + 1;
    }
  }
  static void contour(const int* x, const int* y, const uint8_t* fl, unsigned n, std::vector<Segment>& out) {
    if (n == 0) return;
    auto on = [&](unsigned i) { return (fl[i] & 1) != 0; };
    int sx, sy;
    unsigned b = 0, e = n;  // iterate points [b, e)
    if (on(0)) { sx = x[0]; sy = y[0]; b = 1; }
    else if (on(n - 1)) { sx = x[n - 1]; sy = y[n - 1]; e = n - 1; }
    else { sx = mid(x[0], x[n - 1]); sy = mid(y[0], y[n - 1]); }
    out.push_back({Segment::MoveTo, {sx, 0}, {sy, 0}});
    bool pend = false;
    int px = 0, py = 0, cx = sx, cy = sy;  // pending off-curve point; current pen position
    auto line = [&](int ex, int ey) { if (ex != cx || ey != cy) out.push_back({Segment::LineTo, {ex, 0}, {ey, 0}}); cx = ex; cy = ey; };
    auto quad = [&](int qx, int qy, int ex, int ey) { out.push_back({Segment::QuadTo, {qx, ex}, {qy, ey}}); cx = ex; cy = ey; };
    for (unsigned i = b; i < e; i++) {
      if (on(i)) {
        if (pend) quad(px, py, x[i], y[i]); else line(x[i], y[i]);
        pend = false;
      } else {
        if (pend) quad(px, py, mid(px, x[i]), mid(py, y[i]));
        px = x[i]; py = y[i]; pend = true;
      }
    }
    if (pend) quad(px, py, sx, sy); else line(sx, sy);  // close (a zero-length closing line is dropped)
  }
};

}  // namespace ttf
}  // namespace gsdf
