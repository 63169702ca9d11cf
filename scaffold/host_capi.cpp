// host_capi.cpp -- plain-C access to the C++ host mirror (Builder / threads / scenes) so that the
// Python tests and bench can build the same trees a Go caller would (ctypes; no torch types).
// This is host tooling above the drop-in boundary, not part of the gsdf_hip C ABI (include/gsdf_hip.h).
#include <array>
#include <cstring>
#include <map>
#include <string>

#include "builder.hpp"
#include "textsdf.hpp"
#include "threads.hpp"

using namespace gsdf;

namespace {
thread_local std::string g_err;
struct Handle {
  Builder b;
  explicit Handle(uint64_t f) : b(f) {}
};
}  // namespace

extern "C" {

const char* gsdfb_last_error(void) { return g_err.c_str(); }
void* gsdfb_new(uint64_t flags) { return new Handle(flags); }
void gsdfb_free(void* h) { delete (Handle*)h; }
int gsdfb_num_errs(void* h) { return (int)((Handle*)h)->b.Errs().size(); }
const char* gsdfb_err(void* h, int i) { return ((Handle*)h)->b.Errs().at((size_t)i).c_str(); }
int gsdfb_num_nodes(void* h) { return (int)((Handle*)h)->b.NumNodes(); }
int gsdfb_node_op(void* h, int id) { return ((Handle*)h)->b.Op(id); }

// Generic constructor dispatch: name = reference Builder method name. f = float args, i = int args
// (shader ids first, then ints/bools). Returns node id, or -1 with gsdfb_last_error() set.
int gsdfb_op(void* hv, const char* name, const float* f, int nf, const int* i, int ni) {
  Builder& b = ((Handle*)hv)->b;
  std::string n(name);
  auto need = [&](int wf, int wi) {
    if (nf < wf || ni < wi) throw std::invalid_argument("bad argument count for " + n);
  };
  try {
    // ---- 3D primitives
    if (n == "NewSphere") { need(1, 0); return b.NewSphere(f[0]).id; }
    if (n == "NewBox") { need(4, 0); return b.NewBox(f[0], f[1], f[2], f[3]).id; }
    if (n == "NewBoxFrame") { need(4, 0); return b.NewBoxFrame(f[0], f[1], f[2], f[3]).id; }
    if (n == "NewTorus") { need(2, 0); return b.NewTorus(f[0], f[1]).id; }
    if (n == "NewCylinder") { need(3, 0); return b.NewCylinder(f[0], f[1], f[2]).id; }
    if (n == "NewHexagonalPrism") { need(2, 0); return b.NewHexagonalPrism(f[0], f[1]).id; }
    if (n == "NewTriangularPrism") { need(2, 0); return b.NewTriangularPrism(f[0], f[1]).id; }
    if (n == "NewBoundsBoxFrame") { need(6, 0); return b.NewBoundsBoxFrame(Box3{{f[0], f[1], f[2]}, {f[3], f[4], f[5]}}).id; }
    // ---- 3D ops
    if (n == "Union") { std::vector<Shader3D> v; for (int k = 0; k < ni; k++) v.push_back({i[k]}); return b.Union(v).id; }
    if (n == "Difference") { need(0, 2); return b.Difference({i[0]}, {i[1]}).id; }
    if (n == "Intersection") { need(0, 2); return b.Intersection({i[0]}, {i[1]}).id; }
    if (n == "Xor") { need(0, 2); return b.Xor({i[0]}, {i[1]}).id; }
    if (n == "SmoothUnion") { need(1, 2); return b.SmoothUnion(f[0], {i[0]}, {i[1]}).id; }
    if (n == "SmoothDifference") { need(1, 2); return b.SmoothDifference(f[0], {i[0]}, {i[1]}).id; }
    if (n == "SmoothIntersect") { need(1, 2); return b.SmoothIntersect(f[0], {i[0]}, {i[1]}).id; }
    if (n == "Scale") { need(1, 1); return b.Scale({i[0]}, f[0]).id; }
    if (n == "Symmetry") { need(0, 4); return b.Symmetry({i[0]}, i[1], i[2], i[3]).id; }
    if (n == "Transform") { need(16, 1); Mat4 m; std::memcpy(m.m, f, 64); return b.Transform({i[0]}, m).id; }
    if (n == "Rotate") { need(4, 1); return b.Rotate({i[0]}, f[0], Vec3{f[1], f[2], f[3]}).id; }
    if (n == "Translate") { need(3, 1); return b.Translate({i[0]}, f[0], f[1], f[2]).id; }
    if (n == "Offset") { need(1, 1); return b.Offset({i[0]}, f[0]).id; }
    if (n == "Array") { need(3, 4); return b.Array({i[0]}, f[0], f[1], f[2], i[1], i[2], i[3]).id; }
    if (n == "Elongate") { need(3, 1); return b.Elongate({i[0]}, f[0], f[1], f[2]).id; }
    if (n == "Shell") { need(1, 1); return b.Shell({i[0]}, f[0]).id; }
    if (n == "CircularArray") { need(0, 3); return b.CircularArray({i[0]}, i[1], i[2]).id; }
    if (n == "Twist") { need(1, 1); return b.Twist({i[0]}, f[0]).id; }
    if (n == "Extrude") { need(1, 1); return b.Extrude({i[0]}, f[0]).id; }
    if (n == "Revolve") { need(1, 1); return b.Revolve({i[0]}, f[0]).id; }
    // ---- 2D primitives
    if (n == "NewLine2D") { need(5, 0); return b.NewLine2D(f[0], f[1], f[2], f[3], f[4]).id; }
    if (n == "NewLines2D") {
      need(1, 0);
      std::vector<std::array<Vec2, 2>> segs;
      for (int k = 1; k + 3 < nf; k += 4) segs.push_back({Vec2{f[k], f[k + 1]}, Vec2{f[k + 2], f[k + 3]}});
      return b.NewLines2D(segs, f[0]).id;
    }
    if (n == "NewArc") { need(3, 0); return b.NewArc(f[0], f[1], f[2]).id; }
    if (n == "NewCircle") { need(1, 0); return b.NewCircle(f[0]).id; }
    if (n == "NewEquilateralTriangle") { need(1, 0); return b.NewEquilateralTriangle(f[0]).id; }
    if (n == "NewRectangle") { need(2, 0); return b.NewRectangle(f[0], f[1]).id; }
    if (n == "NewHexagon") { need(1, 0); return b.NewHexagon(f[0]).id; }
    if (n == "NewOctagon") { need(1, 0); return b.NewOctagon(f[0]).id; }
    if (n == "NewEllipse") { need(2, 0); return b.NewEllipse(f[0], f[1]).id; }
    if (n == "NewPolygon") {
      std::vector<Vec2> v;
      for (int k = 0; k + 1 < nf; k += 2) v.push_back({f[k], f[k + 1]});
      return b.NewPolygon(v).id;
    }
    if (n == "NewDiamond2D") { need(2, 0); return b.NewDiamond2D(f[0], f[1]).id; }
    if (n == "NewRoundedX") { need(2, 0); return b.NewRoundedX(f[0], f[1]).id; }
    if (n == "NewQuadraticBezier2D") { need(7, 0); return b.NewQuadraticBezier2D({f[0], f[1]}, {f[2], f[3]}, {f[4], f[5]}, f[6]).id; }
    // ---- 2D ops
    if (n == "Union2D") { std::vector<Shader2D> v; for (int k = 0; k < ni; k++) v.push_back({i[k]}); return b.Union2D(v).id; }
    if (n == "Difference2D") { need(0, 2); return b.Difference2D({i[0]}, {i[1]}).id; }
    if (n == "Intersection2D") { need(0, 2); return b.Intersection2D({i[0]}, {i[1]}).id; }
    if (n == "Xor2D") { need(0, 2); return b.Xor2D({i[0]}, {i[1]}).id; }
    if (n == "Array2D") { need(2, 3); return b.Array2D({i[0]}, f[0], f[1], i[1], i[2]).id; }
    if (n == "Offset2D") { need(1, 1); return b.Offset2D({i[0]}, f[0]).id; }
    if (n == "Translate2D") { need(2, 1); return b.Translate2D({i[0]}, f[0], f[1]).id; }
    if (n == "Rotate2D") { need(1, 1); return b.Rotate2D({i[0]}, f[0]).id; }
    if (n == "Symmetry2D") { need(0, 3); return b.Symmetry2D({i[0]}, i[1], i[2]).id; }
    if (n == "Annulus") { need(1, 1); return b.Annulus({i[0]}, f[0]).id; }
    if (n == "CircularArray2D") { need(0, 3); return b.CircularArray2D({i[0]}, i[1], i[2]).id; }
    if (n == "Scale2D") { need(1, 1); return b.Scale2D({i[0]}, f[0]).id; }
    if (n == "TranslateMulti2D") {
      need(0, 1);
      std::vector<Vec2> d;
      for (int k = 0; k + 1 < nf; k += 2) d.push_back({f[k], f[k + 1]});
      return b.TranslateMulti2D({i[0]}, d).id;
    }
    if (n == "Elongate2D") { need(2, 1); return b.Elongate2D({i[0]}, f[0], f[1]).id; }
    // ---- forge/threads
    if (n == "threads.ISO.Thread") { need(2, 1); return threads::ISO(f[0], f[1], i[0] != 0).Thread(b).id; }
    if (n == "threads.Screw.ISO") { need(3, 1); return threads::Screw(b, f[2], threads::ISO(f[0], f[1], i[0] != 0)).id; }
    if (n == "threads.Screw.NPT") { need(2, 0); threads::NPT t; t.SetFromNominal(f[0]); return threads::Screw(b, f[1], t).id; }
    if (n == "threads.Nut.NPT") { need(1, 1); threads::NPT t; t.SetFromNominal(f[0]); return threads::Nut(b, t, (threads::NutStyle)i[0]).id; }
    if (n == "threads.Nut.ISO") { need(2, 2); return threads::Nut(b, threads::ISO(f[0], f[1], i[0] != 0), (threads::NutStyle)i[1]).id; }
    if (n == "threads.Bolt.ISO") {
      need(4, 2);
      threads::ISO t(f[0], f[1], i[0] != 0);
      threads::BoltParams bp;
      bp.Thread = &t; bp.Style = (threads::NutStyle)i[1]; bp.TotalLength = f[2]; bp.ShankLength = f[3];
      return threads::Bolt(b, bp).id;
    }
    if (n == "threads.HexHead") { need(2, 2); return threads::HexHead(b, f[0], f[1], i[0] != 0, i[1] != 0).id; }
    if (n == "threads.KnurledHead") { need(3, 0); return threads::KnurledHead(b, f[0], f[1], f[2]).id; }
    if (n == "threads.Screw.PlasticButtress") { need(3, 0); return threads::Screw(b, f[2], threads::PlasticButtress(f[0], f[1])).id; }
    if (n == "threads.Knurl") {
      need(5, 0);
      threads::KnurlParams k;
      k.Length = f[0]; k.Radius = f[1]; k.Pitch = f[2]; k.Height = f[3]; k.Theta = f[4];
      return threads::Knurl(b, k).id;
    }
    // ---- scenes (benchmark configs)
    if (n == "scene.npt-flange") return scenes::NptFlange(b).id;
    if (n == "scene.bolt") return scenes::Bolt(b).id;
    if (n == "scene.fibonacci-showerhead") return scenes::Showerhead(b).id;
    if (n == "threads.PlasticButtress.Thread") { need(2, 0); return threads::PlasticButtress(f[0], f[1]).Thread(b).id; }
    if (n == "scene.glyph-plate") return scenes::GlyphPlate(b, ni > 0 ? i[0] : 24).id;
    if (n == "scene.knurled-cylinder") return scenes::KnurledCylinder(b, nf > 0 ? f[0] : 20.f).id;
    g_err = "unknown builder method: " + n;
    return -1;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

// forge/textsdf: Font.LoadTTFBytes + Configure(RelativeGlyphTolerance) + TextLine(text) on this builder
// (font.go:39-137). Returns the 2-D node id of the line, or -1 with gsdfb_last_error() set (the reference's error
// texts: "no text provided", "char ... not graphic", "invalid RelativeGlyphTolerance", sfnt parse errors).
// advance_out / kern_out (optional): Font.AdvanceWidth / Font.Kern of the first (two) runes, for tests.
int gsdfb_textsdf_line(void* hv, const uint8_t* ttf, size_t ttf_len, const char* utf8, float reltol, float* advance_out, float* kern_out) {
  Builder& b = ((Handle*)hv)->b;
  try {
    if (!ttf || !utf8) throw std::invalid_argument("null argument");
    textsdf::Font f(b);
    textsdf::FontConfig cfg;
    cfg.RelativeGlyphTolerance = reltol;
    f.Configure(cfg);
    f.LoadTTFBytes(ttf, ttf_len);
    const std::string s(utf8);
    if (advance_out && !s.empty()) *advance_out = f.AdvanceWidth((unsigned char)s[0]);
    if (kern_out && s.size() > 1) *kern_out = f.Kern((unsigned char)s[0], (unsigned char)s[1]);
    return f.TextLine(s).id;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

// Bounds of node id (3D: min xyz max xyz ; 2D: min xy 0 max xy 0).
int gsdfb_bounds(void* hv, int id, float bb[6]) {
  Builder& b = ((Handle*)hv)->b;
  if (id < 0 || (size_t)id >= b.NumNodes()) { g_err = "bad node id"; return -1; }
  if (gsdf_op_is2d(b.Op(id))) {
    Box2 x = b.Bounds(Shader2D{id});
    bb[0] = x.Min.X; bb[1] = x.Min.Y; bb[2] = 0; bb[3] = x.Max.X; bb[4] = x.Max.Y; bb[5] = 0;
  } else {
    Box3 x = b.Bounds(Shader3D{id});
    bb[0] = x.Min.X; bb[1] = x.Min.Y; bb[2] = x.Min.Z; bb[3] = x.Max.X; bb[4] = x.Max.Y; bb[5] = x.Max.Z;
  }
  return 0;
}

// Borrowed tree view rooted at id (valid until the builder is mutated or freed).
int gsdfb_tree(void* hv, int id, gsdf_tree* out) {
  Builder& b = ((Handle*)hv)->b;
  try {
    if (id < 0 || (size_t)id >= b.NumNodes()) throw std::invalid_argument("bad node id");
    *out = gsdf_op_is2d(b.Op(id)) ? b.Tree2D(Shader2D{id}) : b.Tree(Shader3D{id});
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}

}  // extern "C"
