// compile.cpp -- lowers a gsdf tree blob to the device instruction stream (dev_ops.h).
//
// Every constant emitted here is computed with the same float32 operation sequence the reference's
// Evaluate methods use for their loop-invariant values (cpu_evaluators.go, cited per case), so the
// device sees bit-identical parameters. MUST be compiled with -ffp-contract=off.
#include "compile.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>

#include "host_math.h"
#include "dev_ops.h"

namespace gsdf_dev {

// r = RN(1/d) for the device's div_by_uniform, or 0 when d is unsuitable (the device then uses the IEEE
// expansion). The candidate from double arithmetic is checked against its float neighbours with the exact
// residual 1 - d*r (d*r has <= 48 significant bits: exact in long double), so r is the correctly rounded
// reciprocal, not a double-rounded one.
float recip_for(float d) {
  const float a = std::fabs(d);
  if (!(a >= 9.313225746154785e-10f && a <= 1073741824.0f)) return 0.f;  // outside [2^-30, 2^30], zero, NaN
  float best = (float)(1.0 / (double)d);
  long double berr = fabsl(1.0L - (long double)d * (long double)best);
  const float cand[2] = {std::nextafterf(best, INFINITY), std::nextafterf(best, -INFINITY)};
  for (float c : cand) {
    long double err = fabsl(1.0L - (long double)d * (long double)c);
    if (err < berr) { berr = err; best = c; }
  }
  return best;
}

namespace {

constexpr float SQRT3 = 1.7320508075688772935274463415058723669428052538103806280558069794f;

struct Ctx {
  const gsdf_tree* t;
  std::vector<uint32_t> code;
  int slots = 0;      // currently allocated
  int max_slots = 0;
  int lip_depth = 0, max_lip_depth = 0;  // nesting of the position maps that stretch (interval stack of D_LIP_PUSH / _POP)
  int lip_push() { const int d = lip_depth++; if (lip_depth > max_lip_depth) max_lip_depth = lip_depth; op(D_LIP_PUSH, d); return d; }
  void lip_pop(int d) { op(D_LIP_POP, d); lip_depth--; }
  // Brick masks (dev_ops.h: D_SKIP / D_LIP_DOM). discont > 0: inside a position map that jumps (array cell, circular sector, screw
  // sawtooth) -- no numbers there. The lowering runs twice: the first run lists the candidate operand subtrees in emission order
  // with their costs, the 16 most expensive get a number, the second run emits them.
  int discont = 0;
  bool numbering = false;          // second run
  std::vector<double> cand_cost;   // first run: cost of candidate i
  std::vector<int> cand_id;        // second run: number of candidate i, or -1
  std::vector<uint64_t> cand_key;  // first run: (node, child slot) of candidate i; the second run must meet the same subtree at the same place
  size_t cand_next = 0;
  bool cand_mismatch = false;      // second run: a candidate came up out of the first run's order (its numbers would name other subtrees)
  int candidate(double cost, uint32_t node, uint32_t child_slot) {
    const size_t i = cand_next++;
    const uint64_t key = ((uint64_t)node << 32) | child_slot;
    if (!numbering) { cand_cost.push_back(cost); cand_key.push_back(key); return -1; }
    if (i >= cand_id.size() || cand_key[i] != key) { cand_mismatch = true; return -1; }
    return cand_id[i];
  }
  size_t max_code;
  std::vector<int8_t> clob;  // memo: -1 unknown, 0/1
  struct Table { size_t patch; float angle; int n; };
  std::vector<Table> tables;  // D_CIRC_PRE sin/cos tables, emitted after D_END
  // Static tracking of "which P.xy value is live" so that hypot(P.x,P.y) can be shared by sibling
  // primitives (three cylinders and the screw of npt-flange all see the same x,y): every instruction
  // that rewrites x or y starts a new version; saved positions remember theirs.
  uint32_t xyver = 1, next_ver = 2, hxyver = 0;
  std::vector<uint32_t> slotver = std::vector<uint32_t>(1 << 16, 0);
  void bump() { xyver = next_ver++; }
  // Full-position version (x, y and z): a new one after every position-rewriting instruction (see op()).
  // Saved positions that are still live are remembered with their versions, so a combine that needs the
  // position it was entered with does not store a second copy of a value some enclosing frame already
  // holds (deep CSG trees otherwise spend most of their LDS slots on copies of the root position, and the
  // slot count decides how many points per lane the interpreter can batch).
  uint32_t pver = 1;
  std::vector<uint32_t> slotpver = std::vector<uint32_t>(1 << 16, 0);
  // Which ENTRY coordinates (bit 0 x, bit 1 y, bit 2 z) each component of the current position depends on: lets
  // the mesher's leaf kernels share f(P.x,P.y) / g(P.z) between cube corners that enter with equal x,y / equal z
  // (D_FLAG_SHXY / D_FLAG_SHZ in dev_ops.h). Conservative: anything not known to be component-wise sets all bits.
  uint8_t dep[3] = {1, 2, 4};
  std::vector<uint32_t> slotdep = std::vector<uint32_t>(1 << 16, 0x070707u);
  uint32_t shxy_flag() const { return ((dep[0] | dep[1]) & 4) ? 0u : D_FLAG_SHXY; }
  uint32_t shz_flag() const { return (dep[2] & 3) ? 0u : D_FLAG_SHZ; }
  void dep_after(uint32_t base) {
    switch (base) {
      case D_TRANSLATE: case D_SCALE_PRE: case D_SYMMETRY: case D_ELONGATE_PRE: case D_ELONGATE2D_PRE: case D_EXTRUDE_PRE:
        break;  // component-wise (or position untouched)
      case D_ROT2D: case D_CIRC_PRE: dep[0] = dep[1] = (uint8_t)(dep[0] | dep[1]); break;
      case D_TWIST: dep[0] = dep[1] = (uint8_t)(dep[0] | dep[1] | dep[2]); break;
      default: dep[0] = dep[1] = dep[2] = 7; break;
    }
  }
  // Context handed from a (smooth) difference to its MINUEND when the subtrahend c was evaluated first: values of the
  // minuend that are <= -c - ok (and negative) are discarded by that difference, see D_GATE* in dev_ops.h.
  struct Outer { bool valid = false; int slot = 0; float ok = 0.f; } outer;
  struct Saved { int slot; bool is2d; };
  std::vector<Saved> live;
  int find_saved(bool is2d) const {
    for (size_t k = live.size(); k-- > 0;) {
      const Saved& s = live[k];
      if (is2d ? (slotver[(size_t)s.slot] == xyver) : (!s.is2d && slotpver[(size_t)s.slot] == pver)) return s.slot;
    }
    return -1;
  }
  void mark_saved(int slot, bool is2d) {
    slotver[(size_t)slot] = xyver;
    slotpver[(size_t)slot] = pver;
    slotdep[(size_t)slot] = dep[0] | (dep[1] << 8) | (dep[2] << 16);
    live.push_back({slot, is2d});
  }
  void load_saved(int slot, bool is2d) {
    op(is2d ? D_LOADP2 : D_LOADP3, slot);
    xyver = slotver[(size_t)slot];
    pver = is2d ? next_ver++ : slotpver[(size_t)slot];  // LOADP2 leaves z as it was: treat as a new position
    const uint32_t d = slotdep[(size_t)slot];
    dep[0] = (uint8_t)d; dep[1] = (uint8_t)(d >> 8);
    if (!is2d) dep[2] = (uint8_t)(d >> 16);
  }
  uint32_t hxy_flag() {  // call when emitting a consumer of hypot(P.x,P.y)
    uint32_t f = (hxyver == xyver) ? D_FLAG_HXY : 0u;
    hxyver = xyver;
    return f;
  }

  int alloc(int n) {
    int s = slots;
    slots += n;
    if (slots > max_slots) max_slots = slots;
    if (slots > 0xffff) throw std::runtime_error("too many scratch slots");
    return s;
  }
  void release(int n) { slots -= n; }
  void op(uint32_t o, int slot = 0) {
    if (code.size() > max_code) throw std::runtime_error("program too large after unrolling multi-evaluation nodes");
    code.push_back(o | ((uint32_t)slot << 16));
    const uint32_t base = o & D_OP_MASK;
    if (base >= D_TRANSLATE && base <= D_LOADP2_SUB) { pver = next_ver++; dep_after(base); }
  }
  void f(float v) { uint32_t u; std::memcpy(&u, &v, 4); code.push_back(u); }
  void u(uint32_t v) { code.push_back(v); }
  const gsdf_node& node(uint32_t i) const { return t->nodes[i]; }
  uint32_t child(const gsdf_node& n, uint32_t k) const { return t->links[n.link_off + k]; }
};

// ---- interval mode of the octree's centre tests (dev_ops.h: D_LIP_*): constants of the stretch factors. The oracle
// computes the same numbers with the same operation sequences (oracle/orc_eval.c: lip_norm3, lip_norm2, lip_screw_seam).
// Largest singular value of the 3x3 linear part of a row-major 4x4 (cyclic Jacobi on A^T A, fixed sweeps, double).
float lip_norm3(const float* m) {
  double b[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double acc = 0;
      for (int k = 0; k < 3; k++) acc += (double)m[4 * k + i] * (double)m[4 * k + j];
      b[i][j] = acc;
    }
  for (int sweep = 0; sweep < 12; sweep++)
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        double apq = b[p][q];
        if (apq == 0.0) continue;
        double theta = (b[q][q] - b[p][p]) / (2.0 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < 3; k++) { double x = b[k][p], y = b[k][q]; b[k][p] = c * x - sn * y; b[k][q] = sn * x + c * y; }
        for (int k = 0; k < 3; k++) { double x = b[p][k], y = b[q][k]; b[p][k] = c * x - sn * y; b[q][k] = sn * x + c * y; }
      }
  double e = b[0][0] > b[1][1] ? b[0][0] : b[1][1];
  if (b[2][2] > e) e = b[2][2];
  return (float)std::sqrt(e);
}
float lip_norm2(float a, float b, float c, float d) {  // 2x2 [[a b][c d]]
  double S = (double)a * a + (double)b * b + (double)c * c + (double)d * d, D = (double)a * d - (double)b * c;
  double disc = S * S - 4.0 * D * D;
  if (disc < 0) disc = 0;
  return (float)std::sqrt(0.5 * (S + std::sqrt(disc)));
}
constexpr float kLipRigidTol = 1.0f;       // a linear map of norm <= 1 keeps the ball's radius; anything above (also the 1.0000001 of a
constexpr float kLipRoundUp = 1.000001f;  // rotation built from rounded sines) stretches it, and the factor is rounded up
// How much a screw's field may jump across a seam of its sawtooth: the profile is evaluated at ONE period of the axial
// coordinate, so its field at (+pitch/2, y) meets its field at (-pitch/2, y) there. They differ by less than the distance
// between the two points, the pitch; if the profile is a polygon that is its own mirror image in x up to d (vertex i of the
// mirror image within d, per coordinate, of vertex k - i of the polygon for some k: the same boundary the other way round),
// the boundaries lie within sqrt2 d of each other and so do their distance fields: 3 d. ISO / NPT forms come out of
// PolygonBuilder.Smooth symmetric to a few 1e-7; buttress forms are not symmetric at all.
float lip_screw_seam(const gsdf_tree& t, uint32_t child, float pitch) {
  const gsdf_node& c = t.nodes[child];
  const float ap = std::fabs(pitch);
  if (c.op != GSDF_POLY2D) return ap;
  const float* v = &t.aux[c.aux_off];
  const uint32_t nv = c.aux_len / 2;
  float best = ap;
  for (uint32_t k = 0; k < nv; k++) {
    float dk = 0.0f;
    for (uint32_t i = 0; i < nv; i++) {
      const uint32_t w = (k + nv - i) % nv;
      const float dx = std::fabs(-v[2 * i] - v[2 * w]), dy = std::fabs(v[2 * i + 1] - v[2 * w + 1]);
      if (dx > dk) dk = dx;
      if (dy > dk) dk = dy;
    }
    if (dk < best) best = dk;
  }
  return best <= 1e-3f * ap ? 3.0f * best : ap;
}

// Does evaluating node i overwrite the position register?
bool clobbers(Ctx& c, uint32_t i) {
  if (c.clob[i] >= 0) return c.clob[i];
  const gsdf_node& n = c.node(i);
  bool r;
  switch (n.op) {
    case GSDF_OFFSET: case GSDF_OFFSET2D: case GSDF_ANNULUS2D:
      r = clobbers(c, c.child(n, 0));
      break;
    case GSDF_UNION: case GSDF_INTERSECT: case GSDF_DIFF: case GSDF_XOR: case GSDF_SMOOTH_UNION:
    case GSDF_SMOOTH_DIFF: case GSDF_SMOOTH_INTERSECT: case GSDF_UNION2D: case GSDF_INTERSECT2D:
    case GSDF_DIFF2D: case GSDF_XOR2D: {
      r = false;
      for (uint32_t k = 0; k < n.nchild; k++) r = r || clobbers(c, c.child(n, k));
      break;
    }
    default:
      r = n.nchild > 0;  // every other operator rewrites P; primitives do not
  }
  c.clob[i] = r;
  return r;
}

void gen(Ctx& c, uint32_t i, int depth);

// ---------------------------------------------------------------------------------------------------------------------
// Lower-bound regions. For a subtree S, lower_region() answers with a region G such that, for every point p OUTSIDE G,
//   S(p) >= LB_G(p) > 0,        LB_BOX(p)  = max over axes of (mn - p, p - mx)              (Chebyshev distance)
//                               LB_ZCYL(p) = max(z0 - z, z - z1, rs * (rho - r), rs * (rin - rho)), rho = hypot(x - cx, y - cy)
// (the device uses a cheaper lower estimate of the hypot, see region_lb in interp.h). Both are lower bounds of the
// Euclidean distance to G, which in turn bounds exact-distance fields from below; the screw and the smooth combines
// are not exact, their rules are derived one by one below. The claim is about values > 0 only: inside G nothing is said.
// Users: the gates of gen_combine (a child whose lower bound proves it cannot influence its parent's combine for any
// point of a wave is skipped, D_GATE*), and Program::exact_bb (dual contouring's origin pass).
//   *solid (optional): additionally the shape is non-empty inside the box, so the field is also <= the distance to the
//   box's farthest corner (the upper bound D_UBOUND* uses); only claimed for exact primitives under rigid motions,
//   uniform scaling and plain unions.
// ---------------------------------------------------------------------------------------------------------------------
struct Region {
  enum Kind { NONE = 0, BOX = 1, ZCYL = 2, OBOX = 3 } kind = NONE;
  float b[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // BOX: min xyz, max xyz (2-D: z = 0) | ZCYL: cx cy r z0 z1 rs rin | OBOX: cx cy c s hx hy z0 z1
  // OBOX: a box turned about z -- centre (cx, cy), own x axis (c, s), half extents hx, hy, z range; what a rotation about z makes
  // of a BOX (3-D only). Bound outside: the Chebyshev distance in the box's frame. Everything that cannot keep the orientation
  // (hulls, mirrors, cylinders) goes through its axis-aligned hull, obox_aabb.
  // ZCYL's rin (0: solid): additionally S(p) >= rs * (rin - rho) for rho < rin at ANY z -- the shape keeps at least rin away
  // from the axis (an annulus: boxes on a circle around the axis; lets a gate fire for points near the axis)
};

// Axis-aligned hull of a turned box (padded by the rounding of its corners).
Region obox_aabb(const Region& g) {
  if (g.kind != Region::OBOX) return g;
  const double ex = std::fabs((double)g.b[2]) * g.b[4] + std::fabs((double)g.b[3]) * g.b[5], ey = std::fabs((double)g.b[3]) * g.b[4] + std::fabs((double)g.b[2]) * g.b[5];
  const double pad = (std::fabs((double)g.b[0]) + std::fabs((double)g.b[1]) + ex + ey) * 1e-6 + 1e-30;
  Region o;
  o.kind = Region::BOX;
  o.b[0] = (float)(g.b[0] - ex - pad); o.b[3] = (float)(g.b[0] + ex + pad);
  o.b[1] = (float)(g.b[1] - ey - pad); o.b[4] = (float)(g.b[1] + ey + pad);
  o.b[2] = g.b[6]; o.b[5] = g.b[7];
  return o;
}

// The z-axis cylinder (about the axis through (cx, cy)) that encloses a region: what survives a rotation about z.
Region enclose_zcyl(const Region& gin, float cx, float cy, bool is2d) {
  Region o;
  o.kind = Region::ZCYL;
  o.b[0] = cx; o.b[1] = cy;
  if (gin.kind == Region::OBOX) {
    // farthest corner and nearest point of the turned rectangle, in its own frame (the axis at q = R^T (axis - centre))
    const double dx = (double)cx - gin.b[0], dy = (double)cy - gin.b[1];
    const double qx = gin.b[2] * dx + gin.b[3] * dy, qy = gin.b[2] * dy - gin.b[3] * dx;
    const double fx = std::fabs(qx) + gin.b[4], fy = std::fabs(qy) + gin.b[5];
    const double nx = std::fmax(std::fabs(qx) - gin.b[4], 0.0), ny = std::fmax(std::fabs(qy) - gin.b[5], 0.0);
    const double ext = std::fabs(dx) + std::fabs(dy) + gin.b[4] + gin.b[5];
    o.b[2] = (float)(std::sqrt(fx * fx + fy * fy) * (1 + 1e-6) + ext * 4e-6 + 1e-30);
    o.b[3] = gin.b[6]; o.b[4] = gin.b[7]; o.b[5] = 1.0f;
    o.b[6] = (float)std::fmax(0.0, std::sqrt(nx * nx + ny * ny) * (1 - 1e-6) - ext * 4e-6 - 1e-30);
    return o;
  }
  const Region& g = gin;
  if (g.kind == Region::BOX) {
    double r = 0;
    for (int k = 0; k < 4; k++) {
      const double dx = (double)g.b[(k & 1) ? 3 : 0] - cx, dy = (double)g.b[(k & 2) ? 4 : 1] - cy;
      r = std::fmax(r, std::sqrt(dx * dx + dy * dy));
    }
    o.b[2] = (float)(r * (1 + 1e-6) + 1e-30);
    o.b[3] = is2d ? -3.0e38f : g.b[2]; o.b[4] = is2d ? 3.0e38f : g.b[5]; o.b[5] = 1.0f;
    // nearest point of the box's xy rectangle to the axis (0 if the axis passes through it)
    const double nx = std::fmax(std::fmax((double)g.b[0] - cx, (double)cx - g.b[3]), 0.0), ny = std::fmax(std::fmax((double)g.b[1] - cy, (double)cy - g.b[4]), 0.0);
    o.b[6] = (float)std::fmax(0.0, std::sqrt(nx * nx + ny * ny) * (1 - 1e-6) - 1e-30);
  } else {
    const double dx = (double)g.b[0] - cx, dy = (double)g.b[1] - cy, dc = std::sqrt(dx * dx + dy * dy);
    o.b[2] = (float)((dc + (double)g.b[2]) * (1 + 1e-6) + 1e-30);
    o.b[3] = g.b[3]; o.b[4] = g.b[4]; o.b[5] = g.b[5];
    // the other axis lies outside that cylinder by dc - r, or inside its hole by rin - dc; the inner claim carries rs
    o.b[6] = (float)std::fmax(0.0, std::fmax(dc - (double)g.b[2], (double)g.b[6] - dc) * (1 - 1e-6) - 1e-30);
    if (dc == 0.0) o.b[6] = g.b[6];
  }
  return o;
}

// Smallest region of a common kind holding both (what a minimum of the two fields is bounded by).
bool hull(const Region& a_in, const Region& b_in, bool is2d, Region& o) {
  if (a_in.kind == Region::NONE || b_in.kind == Region::NONE) return false;
  const Region a = obox_aabb(a_in), b = obox_aabb(b_in);
  if (a.kind == Region::BOX && b.kind == Region::BOX) {
    o.kind = Region::BOX;
    for (int j = 0; j < 3; j++) { o.b[j] = std::fmin(a.b[j], b.b[j]); o.b[j + 3] = std::fmax(a.b[j + 3], b.b[j + 3]); }
    return true;
  }
  // at least one cylinder: both about the axis of the (first) cylinder
  const Region& zc = a.kind == Region::ZCYL ? a : b;
  const Region ea = enclose_zcyl(a, zc.b[0], zc.b[1], is2d), eb = enclose_zcyl(b, zc.b[0], zc.b[1], is2d);
  o.kind = Region::ZCYL;
  o.b[0] = zc.b[0]; o.b[1] = zc.b[1];
  o.b[2] = std::fmax(ea.b[2], eb.b[2]);
  o.b[3] = std::fmin(ea.b[3], eb.b[3]); o.b[4] = std::fmax(ea.b[4], eb.b[4]);
  o.b[5] = std::fmin(ea.b[5], eb.b[5]);
  o.b[6] = std::fmin(ea.b[6], eb.b[6]);
  return true;
}

// field - d (d >= 0): the region grows by d (ZCYL: radially by d / rs, since the radial term carries the factor rs)
void inflate(Region& g, float d, bool is2d) {
  if (g.kind == Region::OBOX) {
    g.b[4] += d; g.b[5] += d; g.b[6] -= d; g.b[7] += d;
  } else if (g.kind == Region::BOX) {
    for (int j = 0; j < (is2d ? 2 : 3); j++) { g.b[j] -= d; g.b[j + 3] += d; }
  } else if (g.kind == Region::ZCYL) {
    g.b[2] += d / g.b[5];
    g.b[3] -= d; g.b[4] += d;
    g.b[6] = std::fmax(0.f, g.b[6] - d / g.b[5]);
  }
}

bool lower_region(Ctx& c, uint32_t i, Region& out, int depth = 0, bool* solid = nullptr) {
  bool dummy = true;
  if (!solid) solid = &dummy;
  if (depth == 0) *solid = true;
  out.kind = Region::NONE;
  if (depth > 64) return false;
  const gsdf_node& n = c.node(i);
  const float* P = n.p;
  const bool is2d = gsdf_op_is2d(n.op);
  auto box = [&](float x0, float y0, float z0, float x1, float y1, float z1) {
    out.kind = Region::BOX;
    out.b[0] = x0; out.b[1] = y0; out.b[2] = z0; out.b[3] = x1; out.b[4] = y1; out.b[5] = z1;
    return true;
  };
  auto child_region = [&](uint32_t k, Region& g) { return n.nchild > k && lower_region(c, c.child(n, k), g, depth + 1, solid); };
  switch (n.op) {
    // ---- exact-distance primitives: the field is the Euclidean distance to the shape, which lies inside its box
    case GSDF_SPHERE: return box(-P[0], -P[0], -P[0], P[0], P[0], P[0]);
    case GSDF_BOX: return box(-0.5f * P[0], -0.5f * P[1], -0.5f * P[2], 0.5f * P[0], 0.5f * P[1], 0.5f * P[2]);
    case GSDF_CYLINDER: return box(-P[0], -P[0], -0.5f * P[1], P[0], P[0], 0.5f * P[1]);
    case GSDF_CIRCLE2D: return box(-P[0], -P[0], 0, P[0], P[0], 0);
    case GSDF_RECT2D: return box(-0.5f * P[0], -0.5f * P[1], 0, 0.5f * P[0], 0.5f * P[1], 0);
    case GSDF_POLY2D: {
      const uint32_t nv = n.aux_len / 2;
      if (nv < 3) return false;
      const float* v = &c.t->aux[n.aux_off];
      float x0 = v[0], y0 = v[1], x1 = v[0], y1 = v[1];
      for (uint32_t k = 1; k < nv; k++) {
        x0 = std::fmin(x0, v[2 * k]); x1 = std::fmax(x1, v[2 * k]);
        y0 = std::fmin(y0, v[2 * k + 1]); y1 = std::fmax(y1, v[2 * k + 1]);
      }
      return box(x0, y0, 0, x1, y1, 0);
    }
    // ---- rigid motions and uniform scaling carry the claim along
    case GSDF_TRANSLATE: case GSDF_TRANSLATE2D: {
      if (n.nchild != 1 || !child_region(0, out)) return false;
      const float tz = n.op == GSDF_TRANSLATE ? P[2] : 0.f;
      if (out.kind == Region::BOX) { out.b[0] += P[0]; out.b[3] += P[0]; out.b[1] += P[1]; out.b[4] += P[1]; out.b[2] += tz; out.b[5] += tz; }
      else if (out.kind == Region::OBOX) { out.b[0] += P[0]; out.b[1] += P[1]; out.b[6] += tz; out.b[7] += tz; }
      else { out.b[0] += P[0]; out.b[1] += P[1]; out.b[3] += tz; out.b[4] += tz; }
      return true;
    }
    case GSDF_SCALE: case GSDF_SCALE2D: {  // f(p) = s * g(p / s): LB_f(p) = s * LB_g(p / s) = LB of the scaled region
      if (n.nchild != 1 || !(P[0] > 0) || !child_region(0, out)) return false;
      if (out.kind == Region::OBOX) {
        out.b[0] *= P[0]; out.b[1] *= P[0];
        for (int j = 4; j < 8; j++) out.b[j] *= P[0];
        return true;
      }
      for (int j = 0; j < (out.kind == Region::BOX ? 6 : 5); j++)
        if (std::fabs(out.b[j]) < 3.0e38f) out.b[j] *= P[0];  // (not the +-3e38 "unbounded" sentinels of 2-D regions)
      if (out.kind == Region::ZCYL) out.b[6] *= P[0];
      return true;
    }
    case GSDF_TRANSFORM: case GSDF_ROTATION2D: {  // rigid motions only: p_local = A p + b with A orthonormal
      if (n.nchild != 1) return false;
      float A[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, b[3] = {0, 0, 0};
      if (n.op == GSDF_TRANSFORM) {
        if (n.aux_len < 16) return false;
        const float* m = &c.t->aux[n.aux_off];
        for (int r = 0; r < 3; r++) { for (int q = 0; q < 3; q++) A[r][q] = m[4 * r + q]; b[r] = m[4 * r + 3]; }
      } else {
        A[0][0] = P[0]; A[0][1] = P[1]; A[1][0] = P[2]; A[1][1] = P[3];
      }
      for (int r = 0; r < 3; r++)
        for (int q = r; q < 3; q++) {
          const double dot = (double)A[r][0] * A[q][0] + (double)A[r][1] * A[q][1] + (double)A[r][2] * A[q][2];
          if (std::fabs(dot - (r == q ? 1.0 : 0.0)) > 1e-5) return false;  // scaling / shearing matrix: no claim
        }
      Region lr;
      if (!child_region(0, lr)) return false;
      if (lr.kind == Region::ZCYL) {
        // a cylinder about z stays one only under a rotation about z (and any translation)
        if (std::fabs((double)A[2][2] - 1.0) > 1e-6 || std::fabs(A[0][2]) > 1e-6 || std::fabs(A[1][2]) > 1e-6 || std::fabs(A[2][0]) > 1e-6 ||
            std::fabs(A[2][1]) > 1e-6)
          return false;
        const double lx = (double)lr.b[0] - b[0], ly = (double)lr.b[1] - b[1];
        out = lr;
        out.b[0] = (float)(A[0][0] * lx + A[1][0] * ly); out.b[1] = (float)(A[0][1] * lx + A[1][1] * ly);
        // (a 2-D region's z range is the sentinel +-3e38 = "unbounded": it is neither part of the padding nor moved)
        const bool zfree = lr.b[3] <= -3.0e38f && lr.b[4] >= 3.0e38f;
        double ext = std::fabs((double)out.b[0]) + std::fabs((double)out.b[1]) + std::fabs((double)lr.b[2]) + std::fabs((double)b[2]);
        if (!zfree) ext += std::fabs((double)lr.b[3]) + std::fabs((double)lr.b[4]);
        const float pad = (float)(ext * 4e-5 + 1e-30);
        out.b[2] += pad;
        if (!zfree) { out.b[3] = lr.b[3] - b[2] - pad; out.b[4] = lr.b[4] - b[2] + pad; }
        out.b[6] = std::fmax(0.f, lr.b[6] - pad);
        return true;
      }
      const bool about_z = n.op == GSDF_TRANSFORM && std::fabs((double)A[2][2] - 1.0) <= 1e-6 && std::fabs(A[0][2]) <= 1e-6 && std::fabs(A[1][2]) <= 1e-6 &&
                           std::fabs(A[2][0]) <= 1e-6 && std::fabs(A[2][1]) <= 1e-6;
      if (about_z && std::fabs((double)A[0][0] * A[0][1]) > 1e-3) {
        // A rotation about z (not by a multiple of 90 degrees: those stay axis-aligned below) keeps a box a box, turned: world =
        // A^T (local - b). The local box (or turned box) centre cl, own axis ul -> centre A^T (cl - b), axis A^T ul.
        double clx, cly, ulx, uly, hx, hy, z0, z1;
        if (lr.kind == Region::OBOX) { clx = lr.b[0]; cly = lr.b[1]; ulx = lr.b[2]; uly = lr.b[3]; hx = lr.b[4]; hy = lr.b[5]; z0 = lr.b[6]; z1 = lr.b[7]; }
        else { clx = 0.5 * ((double)lr.b[0] + lr.b[3]); cly = 0.5 * ((double)lr.b[1] + lr.b[4]); ulx = 1; uly = 0; hx = 0.5 * ((double)lr.b[3] - lr.b[0]); hy = 0.5 * ((double)lr.b[4] - lr.b[1]); z0 = lr.b[2]; z1 = lr.b[5]; }
        const double dx = clx - b[0], dy = cly - b[1];
        double wx = A[0][0] * dx + A[1][0] * dy, wy = A[0][1] * dx + A[1][1] * dy;
        double ux = A[0][0] * ulx + A[1][0] * uly, uy = A[0][1] * ulx + A[1][1] * uly;
        const double un = std::sqrt(ux * ux + uy * uy);
        if (un > 0.5) {
          ux /= un; uy /= un;
          const double ext = std::fabs(wx) + std::fabs(wy) + hx + hy + std::fabs(z0) + std::fabs(z1) + std::fabs((double)b[2]);
          const double pad = ext * 4e-5 + 1e-30;  // the matrix is orthonormal to 1e-5 only
          out.kind = Region::OBOX;
          out.b[0] = (float)wx; out.b[1] = (float)wy; out.b[2] = (float)ux; out.b[3] = (float)uy;
          out.b[4] = (float)(hx + pad); out.b[5] = (float)(hy + pad);
          out.b[6] = (float)(z0 - b[2] - pad); out.b[7] = (float)(z1 - b[2] + pad);
          return true;
        }
      }
      if (lr.kind == Region::OBOX) lr = obox_aabb(lr);
      const float* lb = lr.b;
      float bb[6];
      for (int j = 0; j < 3; j++) { bb[j] = 3.0e38f; bb[j + 3] = -3.0e38f; }
      for (int k = 0; k < 8; k++) {  // world corner = A^T (corner_local - b); pad by the matrix' deviation from orthonormal
        const double cl[3] = {(double)lb[(k & 1) ? 3 : 0] - b[0], (double)lb[(k & 2) ? 4 : 1] - b[1], (double)lb[(k & 4) ? 5 : 2] - b[2]};
        for (int j = 0; j < 3; j++) {
          const double w = A[0][j] * cl[0] + A[1][j] * cl[1] + A[2][j] * cl[2];
          bb[j] = std::fmin(bb[j], (float)w);
          bb[j + 3] = std::fmax(bb[j + 3], (float)w);
        }
      }
      // an almost-orthonormal matrix (1e-5) moves points by a relative 1e-5 at most: grow the box accordingly
      double ext = 0;
      for (int j = 0; j < 6; j++) ext = std::fmax(ext, std::fabs((double)bb[j]));
      const float pad = (float)(ext * 4e-5 + 1e-30);
      for (int j = 0; j < 3; j++) { bb[j] -= pad; bb[j + 3] += pad; }
      if (n.op == GSDF_ROTATION2D) { bb[2] = 0; bb[5] = 0; }
      return box(bb[0], bb[1], bb[2], bb[3], bb[4], bb[5]);
    }
    case GSDF_SYMMETRY: case GSDF_SYMMETRY2D: {  // child(|p|) on the masked axes: the region and its mirror images
      if (n.nchild != 1 || !child_region(0, out)) return false;
      *solid = false;
      const int bits = (int)P[0];
      out = obox_aabb(out);
      if (out.kind == Region::ZCYL) {
        if ((bits & 3) && (out.b[0] != 0.f || out.b[1] != 0.f)) out = enclose_zcyl(out, 0.f, 0.f, is2d);
        if (bits & 4) { const float m = std::fmax(std::fabs(out.b[3]), std::fabs(out.b[4])); out.b[3] = -m; out.b[4] = m; }
        return true;
      }
      for (int j = 0; j < 3; j++)
        if (bits & (1 << j)) { const float lo = std::fmin(out.b[j], -out.b[j + 3]), hi = std::fmax(out.b[j + 3], -out.b[j]); out.b[j] = lo; out.b[j + 3] = hi; }
      return true;
    }
    // ---- minimum of fields: every child's bound holds outside the hull
    case GSDF_UNION: case GSDF_UNION2D: {
      Region a;
      for (uint32_t k = 0; k < n.nchild; k++) {
        if (!child_region(k, a)) return false;
        if (k == 0) out = a;
        else { Region h; if (!hull(out, a, is2d, h)) return false; out = h; }
      }
      return n.nchild > 0;
    }
    // smooth union = mix(b, a, h) - k h (1 - h) >= min(a, b) - k / 4 (cpu_evaluators.go:213-236): the hull, grown by k / 4
    case GSDF_SMOOTH_UNION: {
      Region a, b2;
      *solid = false;
      if (n.nchild != 2 || !(P[0] > 0) || !child_region(0, a) || !child_region(1, b2) || !hull(a, b2, is2d, out)) return false;
      inflate(out, 0.25f * P[0] * 1.0001f, is2d);
      return true;
    }
    // max(a, .) >= a, and the smooth maxima are >= max: the first operand's bound holds (for an intersection either one's)
    case GSDF_DIFF: case GSDF_DIFF2D: case GSDF_SMOOTH_DIFF:
      *solid = false;  // what is left after the cut may be anywhere in the region, or nothing
      return n.nchild == 2 && child_region(0, out);
    case GSDF_INTERSECT: case GSDF_INTERSECT2D: case GSDF_SMOOTH_INTERSECT:
      *solid = false;
      return n.nchild == 2 && (child_region(0, out) || child_region(1, out));
    case GSDF_EXTRUSION: {  // outside: hypot(max(d,0), max(w,0)) >= max(d, w), w = |z| - h/2, d >= the 2-D child's bound
      Region g;
      if (n.nchild != 1 || !child_region(0, g)) return false;
      out = g;
      if (g.kind == Region::BOX) { out.b[2] = -0.5f * P[0]; out.b[5] = 0.5f * P[0]; }
      else { out.b[3] = -0.5f * P[0]; out.b[4] = 0.5f * P[0]; }
      return true;
    }
    // screw = max(child(p0), |z| - L), p0 = (sawtooth, hypot(x,y) + z tanT) (threads.go:141-181). With [.., y1] the top of
    // the 2-D child's box: child(p0) >= p0.y - y1 >= rho - y1 - |z| t (t = |tanT|). For |z| <= L that is >= rho - r with
    // r = y1 + L t; for |z| > L, max(rho - y1 - |z| t, |z| - L) >= (rho - r) / (1 + t) (the two terms cross there). So
    // screw(p) >= max(|z| - L, (rho - r) / (1 + t)) everywhere outside the cylinder: a ZCYL with rs = 1 / (1 + t).
    case GSDF_SCREW: {
      Region g;
      *solid = false;
      if (n.nchild != 1 || !child_region(0, g) || g.kind != Region::BOX) return false;
      const double t = std::fabs((double)gsdf::tanf32(P[3]));
      if (!(t < 0.5) || !(P[2] > 0)) return false;
      out.kind = Region::ZCYL;
      out.b[0] = 0; out.b[1] = 0;
      out.b[2] = (float)(((double)g.b[4] + (double)P[2] * t) * (1 + 1e-6) + 1e-30);
      out.b[3] = -P[2]; out.b[4] = P[2];
      out.b[5] = (float)((1.0 - 1e-6) / (1.0 + t));
      out.b[6] = 0.f;
      return out.b[2] > 0;
    }
    // rotations about z that depend on the point (twist: by k z; circular array: by a multiple of the sector angle, two
    // candidates, minimum): rho and z of the evaluated position equal those of p, so a cylinder about the z axis holds
    case GSDF_TWIST: case GSDF_CIRCARRAY: case GSDF_CIRCARRAY2D: {
      Region g;
      *solid = false;
      if (n.nchild != 1 || !child_region(0, g)) return false;
      out = enclose_zcyl(g, 0.f, 0.f, is2d);
      return true;
    }
    case GSDF_OFFSET: case GSDF_OFFSET2D: {  // child + off: a negative offset grows the shape by |off|; a positive one only raises the field
      if (n.nchild != 1 || !child_region(0, out)) return false;
      *solid = false;
      if (P[0] < 0) inflate(out, -P[0] * 1.0001f, is2d);
      return true;
    }
    case GSDF_TRANSLATEMULTI2D: {  // min over displacements of child(p - d)
      Region g;
      const uint32_t nd = n.aux_len / 2;
      if (n.nchild != 1 || nd == 0 || !child_region(0, g)) return false;
      const float* d = &c.t->aux[n.aux_off];
      for (uint32_t k = 0; k < nd; k++) {
        Region s = g;
        if (s.kind == Region::BOX) { s.b[0] += d[2 * k]; s.b[3] += d[2 * k]; s.b[1] += d[2 * k + 1]; s.b[4] += d[2 * k + 1]; }
        else { s.b[0] += d[2 * k]; s.b[1] += d[2 * k + 1]; }
        if (k == 0) out = s;
        else { Region h; if (!hull(out, s, true, h)) return false; out = h; }
      }
      return true;
    }
    default: return false;
  }
}

// The box form only (Program::exact_bb, D_UBOUND*).
bool exact_box(Ctx& c, uint32_t i, float bb[6], bool* solid = nullptr) {
  Region g;
  if (!lower_region(c, i, g, 0, solid)) return false;
  g = obox_aabb(g);
  if (g.kind != Region::BOX) return false;
  std::memcpy(bb, g.b, 6 * sizeof(float));
  return true;
}

// Rough VALU instructions per point of a subtree: decides which children are worth a gate and which is evaluated last.
double subtree_cost(Ctx& c, uint32_t i, int depth = 0) {
  if (depth > 64) return 1e9;
  const gsdf_node& n = c.node(i);
  double kids = 0;
  for (uint32_t k = 0; k < n.nchild; k++) kids += subtree_cost(c, c.child(n, k), depth + 1);
  switch (n.op) {
    case GSDF_POLY2D: return 20.0 + 24.0 * (n.aux_len / 2);
    case GSDF_LINES2D: return 20.0 + 20.0 * (n.aux_len / 4);
    case GSDF_SCREW: return 150.0 + kids;
    case GSDF_TWIST: return 130.0 + kids;
    case GSDF_CIRCARRAY: case GSDF_CIRCARRAY2D: return 140.0 + 2 * kids;
    case GSDF_ARRAY: return 8 * (40.0 + kids);
    case GSDF_ARRAY2D: return 4 * (30.0 + kids);
    case GSDF_TRANSLATEMULTI2D: return (n.aux_len / 2) * (4.0 + kids);
    case GSDF_ELLIPSE2D: case GSDF_QUADBEZIER2D: return 250.0;
    case GSDF_BOX: case GSDF_BOXFRAME: case GSDF_HEX: return 45.0;
    default: return 25.0 + kids;
  }
}
constexpr double kGateMinCost = 60.0;  // cheaper children are evaluated rather than tested (a gate costs ~12 per point); 80 until the end of round 2, which left the vents of knurled-cylinder (translate + rotation + cylinder = 75) ungated; 40 measured no better for knurled-cylinder and bolt and 6 % worse for npt-flange (gates on cheap primitives that rarely fire)

// n-ary / binary combine frame (cpu_evaluators.go:124-286, 821-912). Children that do not rewrite the
// position are evaluated first (min/max are order-independent; for the asymmetric binary ops the
// combine gets D_FLAG_SWAP), so the position only has to be saved when two or more children rewrite it.
//
// Gates (D_GATE*). With `a` the value of the children evaluated so far (LDS slot) and B the next child, bounded from below
// outside its region by L(p) (lower_region): if for EVERY point of the wave
//     union         min(a, b)            : L > a           -> b >= L > a: the result is a
//     difference    max(a, -b)           : L > -a          -> -b <= -L < a: the result is a
//     smooth union  (either operand order): L - a > 1.002 k -> |b - a| / 2 >= 0.501 k: the blend weight clamps to exactly 0
//                                                           or 1 whatever b is, and b only enters multiplied by 0
//     smooth diff.  (b the subtrahend)   : L + a > 1.002 k -> likewise
// (each with a safety margin, see region_lb / gate_far in interp.h), then B is not evaluated: R <- L (positive, finite,
// on the same side of every comparison as the true b) and the combine instruction runs unchanged on it -- the same
// float operations on operands that yield the same bits. Intersections, xor and the minuend of a difference need b's
// value itself: no gate. Children worth a gate (cost >= kGateMinCost, region known) are evaluated last, cheapest first.
void gen_combine(Ctx& c, const gsdf_node& n, uint32_t comb, bool has_k, int depth) {
  bool is2d = gsdf_op_is2d(n.op);
  const Ctx::Outer outer = c.outer;  // meant for this frame only (set by the enclosing difference right before gen())
  c.outer.valid = false;
  const bool asym = comb == D_COMBINE_DIFF || comb == D_COMBINE_SUNION || comb == D_COMBINE_SDIFF || comb == D_COMBINE_SINTER;
  if (n.nchild != 2 && asym) throw std::runtime_error("asymmetric combine needs exactly 2 children");
  // which children could be gated if evaluated after another one
  const bool wide = comb == D_COMBINE_MIN && n.nchild >= 4;  // wide unions gate every child with a region, cheap or not
  std::vector<Region> reg(n.nchild);
  std::vector<double> cost(n.nchild, 0.0);
  std::vector<char> worth(n.nchild, 0);
  const bool gate_kind = comb == D_COMBINE_MIN || comb == D_COMBINE_DIFF || (comb == D_COMBINE_SUNION && n.p[0] > 0) ||
                         (comb == D_COMBINE_SDIFF && n.p[0] > 0);
  for (uint32_t k = 0; k < n.nchild; k++) {
    cost[k] = subtree_cost(c, c.child(n, k));
    if (!gate_kind) continue;
    if ((comb == D_COMBINE_DIFF || comb == D_COMBINE_SDIFF) && k == 0) continue;  // the minuend's value is the result
    if (lower_region(c, c.child(n, k), reg[k]) && (wide || cost[k] >= kGateMinCost)) worth[k] = 1;
  }
  std::vector<uint32_t> order;
  for (uint32_t k = 0; k < n.nchild; k++) if (!worth[k] && !clobbers(c, c.child(n, k))) order.push_back(k);
  for (uint32_t k = 0; k < n.nchild; k++) if (!worth[k] && clobbers(c, c.child(n, k))) order.push_back(k);
  if (wide) {
    for (uint32_t k = 0; k < n.nchild; k++) if (worth[k] && !clobbers(c, c.child(n, k))) order.push_back(k);
    for (uint32_t k = 0; k < n.nchild; k++) if (worth[k] && clobbers(c, c.child(n, k))) order.push_back(k);
  } else {
    std::vector<uint32_t> w;
    for (uint32_t k = 0; k < n.nchild; k++) if (worth[k]) w.push_back(k);
    std::stable_sort(w.begin(), w.end(), [&](uint32_t x, uint32_t y) { return cost[x] < cost[y]; });
    for (uint32_t k : w) order.push_back(k);
  }
  // a frame whose every child is worth a gate still evaluates the first one unconditionally: prefer one that leaves the position alone
  const bool swapped = n.nchild == 2 && order[0] != 0;
  bool need_save = false;
  for (uint32_t k = 0; k + 1 < n.nchild; k++) need_save = need_save || clobbers(c, c.child(n, order[k]));
  int slotP = -1;
  bool own_save = false;
  if (need_save) {
    slotP = c.find_saved(is2d);  // an enclosing frame already holds exactly this position
    if (slotP < 0) {
      own_save = true;
      slotP = c.alloc(is2d ? 2 : 3);
      c.op(is2d ? D_SAVEP2 : D_SAVEP3, slotP);
      c.mark_saved(slotP, is2d);
    }
  }
  // Wide union with >= 3 exact-boxed children: start the running minimum at an upper bound of the union (D_UBOUND*),
  // so that far children are skipped from the first one on, wherever the point is.
  std::vector<float> ub;
  if (wide) {
    float cb[6];
    for (uint32_t k = 0; k < n.nchild; k++) {
      bool solid = true;
      if (exact_box(c, c.child(n, k), cb, &solid) && solid) {
        if (is2d) { ub.push_back(cb[0]); ub.push_back(cb[1]); ub.push_back(cb[3]); ub.push_back(cb[4]); }
        else for (int j = 0; j < 6; j++) ub.push_back(cb[j]);
      }
    }
    if (ub.size() < (size_t)(is2d ? 12 : 18)) ub.clear();
  }
  const bool bounded = !ub.empty();
  // The partial result gets its slot only when the first child is done: while that child (in left-deep trees, the
  // whole rest of the tree) runs, this frame holds no distance slot.
  int slotD = -1;
  if (bounded) {
    slotD = c.alloc(1);
    c.op(is2d ? D_UBOUND2D : D_UBOUND3D, slotD);
    c.u((uint32_t)(ub.size() / (is2d ? 4 : 6)));
    for (float v : ub) c.f(v);
  }
  // Brick masks: which operands of this frame may carry a number (dev_ops.h: D_SKIP). kind = D_LIP_DOM's.
  int dom_kind = -1;
  switch (comb) {
    case D_COMBINE_MIN: dom_kind = 0; break;
    case D_COMBINE_MAX: dom_kind = 1; break;
    case D_COMBINE_DIFF: dom_kind = 2; break;
    case D_COMBINE_SUNION: if (n.p[0] > 0) dom_kind = 3; break;
    case D_COMBINE_SDIFF: if (n.p[0] > 0) dom_kind = 4; break;
    case D_COMBINE_SINTER: if (n.p[0] > 0) dom_kind = 5; break;
    default: break;
  }
  if (c.discont > 0) dom_kind = -1;
  std::vector<int> sid(n.nchild, -1);  // by child number
  if (dom_kind >= 0)
    for (uint32_t k = 0; k < n.nchild; k++) sid[order[k]] = c.candidate(cost[order[k]], (uint32_t)(&n - c.t->nodes), order[k]);  // (emission order)
  bool any_id = false;
  for (int v : sid) any_id = any_id || v >= 0;
  // the frame's entry position magnitude for D_LIP_DOM's padding: interval stack[depth], second column (D_LIP_PUSH stores it)
  int dom_depth = -1;
  if (any_id) { dom_depth = c.lip_depth++; if (c.lip_depth > c.max_lip_depth) c.max_lip_depth = c.lip_depth; c.op(D_LIP_PUSH, dom_depth); }
  auto subst_of = [&](uint32_t ck) -> float {  // the side of the combine on which a dominated operand lies
    const bool a_role = ck == 0;
    switch (dom_kind) {
      case 0: case 3: return GSDF_SKIP_BIG;
      case 1: case 5: return -GSDF_SKIP_BIG;
      default: return a_role ? -GSDF_SKIP_BIG : GSDF_SKIP_BIG;  // (smooth) difference: minuend far below, subtrahend far above
    }
  };
  bool dirty = false;
  for (uint32_t k = 0; k < n.nchild; k++) {
    if (k > 0 && dirty) { c.load_saved(slotP, is2d); dirty = false; }
    const uint32_t ck = order[k];
    uint32_t ch = c.child(n, ck);
    long skip_at = -1, gate_pc = -1;
    long bskip_pc = -1, bskip_at = -1;
    const uint32_t bs_xyver = c.xyver, bs_hxy = c.hxyver;
    if (sid[ck] >= 0) {
      bskip_pc = (long)c.code.size();
      c.op(D_SKIP | c.shxy_flag(), 0);
      c.u((uint32_t)sid[ck]);
      c.f(subst_of(ck));
      bskip_at = (long)c.code.size();
      c.u(0);  // patched below: words to the end of the subtree
    }
    if (worth[ck] && (k > 0 || bounded)) {
      const Region& g = reg[ck];
      const bool minus = comb == D_COMBINE_DIFF || comb == D_COMBINE_SDIFF;  // compare with -a
      const float kk = (comb == D_COMBINE_SUNION || comb == D_COMBINE_SDIFF) ? 1.002f * n.p[0] : 0.0f;
      gate_pc = (long)c.code.size();
      if (g.kind == Region::BOX) {
        c.op(is2d ? D_GATE2D : D_GATE3D, slotD);
        if (is2d) { c.f(g.b[0]); c.f(g.b[1]); c.f(g.b[3]); c.f(g.b[4]); }
        else { for (int j = 0; j < 6; j++) c.f(g.b[j]); }
      } else if (g.kind == Region::OBOX) {
        c.op(D_GATEOB, slotD);
        for (int j = 0; j < 8; j++) c.f(g.b[j]);
      } else {
        // hypot(P.x, P.y) already in the register (and the cylinder about the origin): the exact radius instead of the estimate
        const bool centred = g.b[0] == 0.f && g.b[1] == 0.f;
        c.op(D_GATEZC | ((centred && c.hxyver == c.xyver) ? D_FLAG_HXY : 0u), slotD);
        for (int j = 0; j < 7; j++) c.f(g.b[j]);
      }
      c.f(minus ? -1.0f : 1.0f);
      c.f(kk);
      // context of the enclosing difference: this frame is its minuend and B its own subtrahend (unswapped)
      const bool ctx = outer.valid && minus && !swapped && ck == 1;
      c.u(ctx ? (uint32_t)outer.slot : 0xffffu);
      c.f(ctx ? outer.ok : 0.0f);
      c.f(ctx && comb == D_COMBINE_SDIFF ? 0.25f * n.p[0] * 1.0001f : 0.0f);
      skip_at = (long)c.code.size();
      c.u(0);  // patched below: words from this instruction to the child's combine instruction | (brick-mask number + 1) << 24
    }
    const uint32_t hxy_before = c.hxyver;
    // a (smooth) difference whose subtrahend is already in slotD hands its minuend the context (k == 1: evaluated second)
    if ((comb == D_COMBINE_DIFF || (comb == D_COMBINE_SDIFF && n.p[0] > 0)) && swapped && k == 1 && slotD >= 0) {
      c.outer.valid = true;
      c.outer.slot = slotD;
      c.outer.ok = comb == D_COMBINE_SDIFF ? 1.002f * n.p[0] : 0.0f;
    }
    gen(c, ch, depth + 1);
    c.outer.valid = false;
    // a child that may be skipped at run time may not have refreshed the hypot(x,y) register: forget what it cached
    if (skip_at >= 0 && c.hxyver != hxy_before) c.hxyver = 0;
    dirty = dirty || clobbers(c, ch);
    if (skip_at >= 0) c.code[(size_t)skip_at] = (uint32_t)((long)c.code.size() - gate_pc) | ((uint32_t)(sid[ck] + 1) << 24);
    if (bskip_at >= 0) {
      c.code[(size_t)bskip_at] = (uint32_t)((long)c.code.size() - bskip_pc);
      // the subtree left hypot(x, y) of ITS ENTRY position in the register (a primitive right under the frame): the skip path
      // computes the same value; anything else it may have cached is forgotten
      if (c.hxyver != bs_hxy) {
        if (c.hxyver == bs_xyver) c.code[(size_t)bskip_pc] |= D_FLAG_HXY;
        else c.hxyver = 0;
      }
    }
    if (k > 0 || bounded) {
      const uint32_t swapf = (asym && swapped) ? D_FLAG_SWAP : 0u;
      const int first_id = (k == 1 && !bounded) ? sid[order[0]] : -1, second_id = sid[ck];
      if (first_id >= 0 || second_id >= 0) {
        const int ida = swapf ? second_id : first_id, idb = swapf ? first_id : second_id;
        c.op(D_LIP_DOM | swapf, slotD);
        c.u((uint32_t)dom_kind);
        c.u(ida >= 0 ? (uint32_t)ida : 0xffu);
        c.u(idb >= 0 ? (uint32_t)idb : 0xffu);
        c.f(has_k ? 1.002f * n.p[0] : 0.0f);
        c.u((uint32_t)dom_depth | (is2d ? 0x10000u : 0u));
      }
      c.op(comb | swapf, slotD);
      if (has_k) { c.f(n.p[0]); c.f(recip_for(n.p[0])); }
    }
    if (k + 1 < n.nchild) {
      if (slotD < 0) slotD = c.alloc(1);
      c.op(D_SAVER, slotD);
    }
  }
  if (dom_depth >= 0) c.lip_depth--;
  if (slotD >= 0) c.release(1);
  if (own_save) { c.live.pop_back(); c.release(is2d ? 2 : 3); }
}

void gen(Ctx& c, uint32_t i, int depth) {
  if (depth > 256) throw std::runtime_error("tree too deep (cycle?)");
  const gsdf_node& n = c.node(i);
  // the enclosing difference's context is about THIS node's value: only a difference consumes it (gen_combine); any
  // other node between the two (a scale multiplies the value, an offset shifts it ...) ends it
  if (n.op != GSDF_DIFF && n.op != GSDF_SMOOTH_DIFF && n.op != GSDF_DIFF2D) c.outer.valid = false;
  const float* P = n.p;
  auto need_children = [&](uint32_t k) { if (n.nchild != k) throw std::runtime_error("bad child count for op " + std::to_string(n.op)); };
  auto child_dim = [&](bool want2d) {
    for (uint32_t k = 0; k < n.nchild; k++)
      if ((bool)gsdf_op_is2d(c.node(c.child(n, k)).op) != want2d) throw std::runtime_error("child dimension mismatch for op " + std::to_string(n.op));
  };
  switch (n.op) {
    // ------------------------------- 3D primitives -------------------------------
    case GSDF_SPHERE: c.op(D_SPHERE); c.f(P[0]); break;                                              // :20-26
    case GSDF_BOX: c.op(D_BOX); c.f(0.5f * P[0]); c.f(0.5f * P[1]); c.f(0.5f * P[2]); c.f(P[3]); break;  // :28-36
    case GSDF_BOXFRAME: {                                                                          // :38-57, primitives.go:292-297
      float e = P[3];
      c.op(D_BOXFRAME); c.f(e);
      c.f(0.5f * P[0] + (-2 * e)); c.f(0.5f * P[1] + (-2 * e)); c.f(0.5f * P[2] + (-2 * e));
      break;
    }
    case GSDF_TORUS: c.op(D_TORUS | c.hxy_flag() | c.shxy_flag()); c.f(P[0]); c.f(P[1]); break;                                   // :59-68
    case GSDF_CYLINDER: {                                                                          // :70-88, primitives.go:147-149
      float r = P[0], h = (P[1] - 2 * P[2]) / 2, round = P[2];
      if (round == 0) { c.op(D_CYL0 | c.hxy_flag() | c.shxy_flag()); c.f(r); c.f(h); }
      else { c.op(D_CYLR | c.hxy_flag() | c.shxy_flag()); c.f(r); c.f(h); c.f(round); }
      break;
    }
    case GSDF_HEX: c.op(D_HEX); c.f(P[0]); c.f(P[1]); c.f(0.57735f * P[0]); break;                  // :90-105
    // ------------------------------- 3D booleans -------------------------------
    case GSDF_UNION: if (n.nchild < 2) throw std::runtime_error("OpUnion must have at least 2 elements"); child_dim(false); gen_combine(c, n, D_COMBINE_MIN, false, depth); break;
    case GSDF_INTERSECT: need_children(2); child_dim(false); gen_combine(c, n, D_COMBINE_MAX, false, depth); break;
    case GSDF_DIFF: need_children(2); child_dim(false); gen_combine(c, n, D_COMBINE_DIFF, false, depth); break;
    case GSDF_XOR: need_children(2); child_dim(false); gen_combine(c, n, D_COMBINE_XOR, false, depth); break;
    case GSDF_SMOOTH_UNION: need_children(2); child_dim(false); gen_combine(c, n, D_COMBINE_SUNION, true, depth); break;
    case GSDF_SMOOTH_DIFF: need_children(2); child_dim(false); gen_combine(c, n, D_COMBINE_SDIFF, true, depth); break;
    case GSDF_SMOOTH_INTERSECT: need_children(2); child_dim(false); gen_combine(c, n, D_COMBINE_SINTER, true, depth); break;
    // ------------------------------- 3D unary -------------------------------
    case GSDF_SCALE: need_children(1); child_dim(false);                                           // :288-312
      { const int ld = c.lip_push(); c.op(D_SCALE_PRE); c.f(1.f / P[0]); c.bump(); gen(c, c.child(n, 0), depth + 1); c.op(D_MULR); c.f(P[0]); c.lip_pop(ld); } break;  // (push / pop: the outer radius comes back exactly, not as (R / s) * s)
    case GSDF_SYMMETRY: need_children(1); child_dim(false);                                        // :314-343
      c.op(D_SYMMETRY); c.u((uint32_t)(int)P[0]); if ((int)P[0] & 3) c.bump(); gen(c, c.child(n, 0), depth + 1); break;
    case GSDF_ARRAY: {                                                                             // :345-397
      need_children(1); child_dim(false);
      int slotP = c.alloc(3), slotD = c.alloc(1);
      c.op(D_SAVEP3, slotP);
      c.op(D_SETSLOT, slotD); c.f(1e20f);
      for (int k = 0; k < 2; k++) for (int j = 0; j < 2; j++) for (int ii = 0; ii < 2; ii++) {
        c.op(D_ARRAY_PRE, slotP);
        c.bump();
        c.f((float)ii); c.f((float)j); c.f((float)k);
        c.f(P[0]); c.f(P[1]); c.f(P[2]);
        c.f(P[3] + -1); c.f(P[4] + -1); c.f(P[5] + -1);
        c.discont++; gen(c, c.child(n, 0), depth + 1); c.discont--;
        c.op(D_COMBINE_MIN, slotD);
        if (!(k == 1 && j == 1 && ii == 1)) c.op(D_SAVER, slotD);
      }
      c.release(4);
      break;
    }
    case GSDF_ELONGATE: {                                                                          // :399-426
      need_children(1); child_dim(false);
      int s = c.alloc(1);
      c.op(D_ELONGATE_PRE, s); c.f(0.5f * P[0]); c.f(0.5f * P[1]); c.f(0.5f * P[2]);
      c.bump();
      gen(c, c.child(n, 0), depth + 1);
      c.op(D_ADDR_SLOT, s);
      c.release(1);
      break;
    }
    case GSDF_SHELL: need_children(1); child_dim(false);                                           // :428-452
      { const int ld = c.lip_push(); c.op(D_SCALE_PRE); c.f(1 / P[0]); c.bump(); gen(c, c.child(n, 0), depth + 1); c.op(D_SHELL_POST); c.f(P[0]); c.lip_pop(ld); } break;
    case GSDF_OFFSET: need_children(1); child_dim(false);                                          // :454-468
      gen(c, c.child(n, 0), depth + 1); c.op(D_ADDR); c.f(P[0]); break;
    case GSDF_TRANSLATE: need_children(1); child_dim(false);                                       // :470-486
      c.op(D_TRANSLATE); c.f(P[0]); c.f(P[1]); c.f(P[2]);
      if (!(P[0] == 0.f && P[1] == 0.f)) c.bump();  // x - 0.0f == x bitwise: a z-only translate keeps hypot(x,y)
      gen(c, c.child(n, 0), depth + 1); break;
    case GSDF_TRANSFORM: {                                                                         // :488-504
      need_children(1); child_dim(false);
      if (n.aux_len < 16) throw std::runtime_error("transform needs 16 aux floats");
      int lipd = -1;
      {  // a map that is not rigid stretches the cube's image: interval mode scales its radius (prune_kernel only)
        float f = lip_norm3(&c.t->aux[n.aux_off]);
        if (f > kLipRigidTol) { lipd = c.lip_push(); c.op(D_LIP_MUL); c.f(f * kLipRoundUp); }
      }
      c.op(D_TRANSFORM);
      for (int k = 0; k < 12; k++) c.f(c.t->aux[n.aux_off + k]);
      c.bump();
      gen(c, c.child(n, 0), depth + 1);
      if (lipd >= 0) c.lip_pop(lipd);
      break;
    }
    case GSDF_CIRCARRAY: case GSDF_CIRCARRAY2D: {                                                  // :1042-1143
      need_children(1);
      bool is2d = n.op == GSDF_CIRCARRAY2D;
      child_dim(is2d);
      int slotP = c.alloc(is2d ? 2 : 3), slotD = c.alloc(1);
      if (!is2d) c.op(D_SAVEP3, slotP);  // keeps z at slotP+2; CIRC_PRE overwrites slotP..+1 with p0.xy
      c.op(D_CIRC_PRE | c.shxy_flag(), slotP);
      c.bump();
      c.slotver[(size_t)slotP] = c.xyver;  // p0
      c.slotpver[(size_t)slotP] = c.next_ver++;
      c.slotdep[(size_t)slotP] = c.dep[0] | (c.dep[1] << 8) | (c.dep[2] << 16);  // p0: same dependencies as p1
      c.live.push_back({slotP, is2d});       // frames of the second pass find p0 here
      c.bump();                              // P = p1
      c.f((float)(2 * gsdf::kPi) / P[1]); c.f(P[1]); c.f((float)((int)P[0] - 1));
      // the two rotations use sincos(angle * i) for an integer i < circleDiv: a table instead of two polynomial
      // evaluations per point (same float32 routine and the same float32 product, computed here)
      c.tables.push_back({c.code.size(), (float)(2 * gsdf::kPi) / P[1], (int)P[1]});
      c.u(0);
      // The two copies are a union: the farther one cannot matter where the child's bound outside its region exceeds the
      // nearer one's value (gen_combine's rule for min). Worth it for a child that costs and has a (turned) box for a
      // region: the wave starts with the copy nearer that box (D_CIRC_ORDER) and tests the other (D_GATEOB).
      Region cg;
      static const bool sector_off = [] { const char* e = getenv("GSDF_HIP_NO_SECTOR_GATE"); return e && atoi(e) != 0; }();  // developer knob (A/B timing)
      // kSectorGateMinCost: order + gate cost ~26 instructions per point and fire for ~40 % of knurled-cylinder's surface bricks
      // (on the lands between its grooves both copies are equally near): at that scene's 95-instruction child the gate costs
      // what it saves (3.375 vs 3.370 ms, profiles/r3b_*), so it is reserved for children that cost several times the test
      constexpr double kSectorGateMinCost = 150.0;
      const bool sector_gate = !sector_off && !is2d && subtree_cost(c, c.child(n, 0)) >= kSectorGateMinCost && lower_region(c, c.child(n, 0), cg) &&
                               (cg.kind == Region::BOX || cg.kind == Region::OBOX);
      if (sector_gate && cg.kind == Region::BOX) {
        Region o;
        o.kind = Region::OBOX;
        o.b[0] = 0.5f * (cg.b[0] + cg.b[3]); o.b[1] = 0.5f * (cg.b[1] + cg.b[4]); o.b[2] = 1.f; o.b[3] = 0.f;
        o.b[4] = 0.5f * (cg.b[3] - cg.b[0]) * 1.000001f; o.b[5] = 0.5f * (cg.b[4] - cg.b[1]) * 1.000001f; o.b[6] = cg.b[2]; o.b[7] = cg.b[5];
        cg = o;
      }
      if (sector_gate) { c.op(D_CIRC_ORDER, slotP); for (int j = 0; j < 6; j++) c.f(cg.b[j]); }
      c.discont++;  // a sector's copy of the child is evaluated where the point's fold puts it: no brick-level numbers inside
      gen(c, c.child(n, 0), depth + 1);  // pos1 first (or whichever D_CIRC_ORDER put there)
      c.op(D_SAVER, slotD);
      c.load_saved(slotP, is2d);
      if (is2d) c.slotpver[(size_t)slotP] = c.pver;
      long skip_at = -1, gate_pc = -1;
      const uint32_t hxy_before = c.hxyver;
      if (sector_gate) {
        gate_pc = (long)c.code.size();
        c.op(D_GATEOB, slotD);
        for (int j = 0; j < 8; j++) c.f(cg.b[j]);
        c.f(1.0f); c.f(0.0f); c.u(0xffffu); c.f(0.0f); c.f(0.0f);
        skip_at = (long)c.code.size();
        c.u(0);
      }
      gen(c, c.child(n, 0), depth + 1);  // pos0
      c.discont--;
      if (skip_at >= 0) {
        if (c.hxyver != hxy_before) c.hxyver = 0;  // a child that may be skipped may not have refreshed the hypot register
        c.code[(size_t)skip_at] = (uint32_t)((long)c.code.size() - gate_pc);
      }
      c.op(D_COMBINE_MIN, slotD);
      c.live.pop_back();
      c.release(is2d ? 3 : 4);
      break;
    }
    case GSDF_TWIST: {                                                                             // :1257-1274
      need_children(1); child_dim(false);
      const int lipd = c.lip_push();  // D_TWIST stretches the interval radius by the twist's shear (interp.h, LIP)
      c.op(D_TWIST | c.shz_flag()); c.f(P[0]); c.bump(); gen(c, c.child(n, 0), depth + 1);
      c.lip_pop(lipd);
      break;
    }
    // ------------------------------- 2D -> 3D -------------------------------
    case GSDF_EXTRUSION: {                                                                         // :506-531
      need_children(1); child_dim(true);
      int s = c.alloc(1);
      c.op(D_EXTRUDE_PRE, s); c.f(P[0] / 2);
      gen(c, c.child(n, 0), depth + 1);
      c.op(D_EXTRUDE_POST, s);
      c.release(1);
      break;
    }
    case GSDF_REVOLUTION: need_children(1); child_dim(true);                                       // :533-549
      c.op(D_REVOLVE_PRE); c.f(P[0]); c.bump(); gen(c, c.child(n, 0), depth + 1); break;
    case GSDF_SCREW: {                                                                             // threads.go:141-181
      need_children(1); child_dim(true);
      int s = c.alloc(1);
      const int lipd = c.lip_push();  // D_SCREW_PRE stretches the interval radius by the helix map (interp.h, LIP)
      c.op(D_SCREW_PRE | c.hxy_flag() | c.shxy_flag(), s);
      c.bump();
      c.f(P[0]); c.f(P[1]); c.f(P[2]); c.f(gsdf::tanf32(P[3])); c.f(P[0] / 2); c.f(recip_for(P[0]));
      c.op(D_LIP_WRAP); c.f(P[0] / 2); c.f(lip_screw_seam(*c.t, c.child(n, 0), P[0]));
      c.discont++; gen(c, c.child(n, 0), depth + 1); c.discont--;  // (the profile coordinate is a sawtooth of the axial one)
      c.lip_pop(lipd);
      c.op(D_MAXR_SLOT, s);
      c.release(1);
      break;
    }
    // ------------------------------- 2D primitives -------------------------------
    case GSDF_LINE2D: {                                                                            // :551-562
      float bax = P[2] - P[0], bay = P[3] - P[1];
      c.op(D_LINE2D); c.f(P[0]); c.f(P[1]); c.f(bax); c.f(bay); c.f(bax * bax + bay * bay); c.f(P[4] / 2);
      break;
    }
    case GSDF_ARC2D: {                                                                             // :564-579
      float s, cs;
      gsdf::sincosf32(P[1] / 2, s, cs);
      c.op(D_ARC2D); c.f(P[0]); c.f(P[2] / 2); c.f(s); c.f(cs); c.f(P[0] * s); c.f(P[0] * cs);
      break;
    }
    case GSDF_QUADBEZIER2D: {                                                                      // :581-593
      float Ax = P[0], Ay = P[1], Bx = P[2], By = P[3], Cx = P[4], Cy = P[5];
      float ax = Bx - Ax, ay = By - Ay;
      float a2 = ax * ax + ay * ay;
      float bx = Ax + (Cx - 2 * Bx), by = Ay + (Cy - 2 * By);
      float cx = 2 * ax, cy = 2 * ay;
      float kk = 1.f / (bx * bx + by * by);
      float kx = kk * (ax * bx + ay * by);
      float kx2 = kx * kx;
      c.op(D_QUADBEZIER2D);
      c.f(Ax); c.f(Ay); c.f(ax); c.f(ay); c.f(a2); c.f(bx); c.f(by); c.f(cx); c.f(cy); c.f(kk); c.f(kx); c.f(kx2); c.f(P[6] / 2);
      break;
    }
    case GSDF_CIRCLE2D: c.op(D_CIRCLE2D | c.hxy_flag() | c.shxy_flag()); c.f(P[0]); break;                                        // :661-667
    case GSDF_EQTRI2D: { float r = P[0] / SQRT3; c.op(D_EQTRI2D); c.f(r); c.f(r / SQRT3); break; }  // :669-683
    case GSDF_RECT2D: c.op(D_RECT2D); c.f(0.5f * P[0]); c.f(0.5f * P[1]); break;                    // :685-692
    case GSDF_DIAMOND2D: {                                                                         // :694-703
      float bx = 0.5f * P[0], by = 0.5f * P[1];
      c.op(D_DIAMOND2D); c.f(bx); c.f(by); c.f(bx * bx + by * by); c.f(0.5f * bx); c.f(0.5f * by); c.f(bx * by);
      break;
    }
    case GSDF_X2D: c.op(D_X2D); c.f(P[0]); c.f(P[1]); break;                                       // :705-716
    case GSDF_HEX2D: c.op(D_HEX2D); c.f(P[0]); c.f(0.577350269f * P[0]); break;                     // :718-729
    case GSDF_OCT2D: c.op(D_OCT2D); c.f(P[0]); c.f(0.4142135623f * P[0]); break;                    // :731-748
    case GSDF_ELLIPSE2D: c.op(D_ELLIPSE2D); c.f(P[0]); c.f(P[1]); break;                           // :750-791
    case GSDF_POLY2D: {                                                                            // :793-818
      uint32_t nv = n.aux_len / 2;
      if (nv < 3) throw std::runtime_error("polygon needs at least 3 vertices");
      const float* v = &c.t->aux[n.aux_off];
      // all edge divisors eligible for the exact reciprocal form? (bit 31 of the vertex-count word)
      bool all_recip = true;
      {
        uint32_t jv0 = nv - 1;
        for (uint32_t iv = 0; iv < nv; iv++) {
          float ex = v[2 * jv0] - v[2 * iv], ey = v[2 * jv0 + 1] - v[2 * iv + 1];
          if (recip_for(ex * ex + ey * ey) == 0.f) all_recip = false;
          jv0 = iv;
        }
      }
      c.op(D_POLY2D); c.u(nv | (all_recip ? 0x80000000u : 0u)); c.f(v[0]); c.f(v[1]);
      while (c.code.size() % 8 != 0) c.u(0);  // edge records: 8 dwords, 32-byte aligned (two s_load_dwordx4 each)
      uint32_t jv = nv - 1;
      for (uint32_t iv = 0; iv < nv; iv++) {
        float v1x = v[2 * iv], v1y = v[2 * iv + 1], v2x = v[2 * jv], v2y = v[2 * jv + 1];
        float ex = v2x - v1x, ey = v2y - v1y;
        const float n2e = ex * ex + ey * ey;
        c.f(v1x); c.f(v1y); c.f(ex); c.f(ey); c.f(n2e); c.f(v2y); c.f(recip_for(n2e)); c.u(0);
        jv = iv;
      }
      break;
    }
    case GSDF_LINES2D: {                                                                           // :1145-1160
      uint32_t ns = n.aux_len / 4;
      const float* sg = &c.t->aux[n.aux_off];
      c.op(D_LINES2D); c.u(ns); c.f(P[0] / 2);
      for (uint32_t k = 0; k < ns; k++) {
        float ax = sg[4 * k], ay = sg[4 * k + 1], bax = sg[4 * k + 2] - ax, bay = sg[4 * k + 3] - ay;
        c.f(ax); c.f(ay); c.f(bax); c.f(bay); c.f(bax * bax + bay * bay);
      }
      break;
    }
    // ------------------------------- 2D ops -------------------------------
    case GSDF_UNION2D: if (n.nchild < 2) throw std::runtime_error("OpUnion2D must have at least 2 elements"); child_dim(true); gen_combine(c, n, D_COMBINE_MIN, false, depth); break;
    case GSDF_INTERSECT2D: need_children(2); child_dim(true); gen_combine(c, n, D_COMBINE_MAX, false, depth); break;
    case GSDF_DIFF2D: need_children(2); child_dim(true); gen_combine(c, n, D_COMBINE_DIFF, false, depth); break;
    case GSDF_XOR2D: need_children(2); child_dim(true); gen_combine(c, n, D_COMBINE_XOR, false, depth); break;
    case GSDF_ARRAY2D: {                                                                           // :914-962
      need_children(1); child_dim(true);
      int slotP = c.alloc(2), slotD = c.alloc(1);
      c.op(D_SAVEP2, slotP);
      c.op(D_SETSLOT, slotD); c.f(1e20f);
      for (int j = 0; j < 2; j++) for (int ii = 0; ii < 2; ii++) {
        c.op(D_ARRAY2D_PRE, slotP);
        c.bump();
        c.f((float)ii); c.f((float)j); c.f(P[0]); c.f(P[1]); c.f(P[2] + -1); c.f(P[3] + -1);
        c.discont++; gen(c, c.child(n, 0), depth + 1); c.discont--;
        c.op(D_COMBINE_MIN, slotD);
        if (!(j == 1 && ii == 1)) c.op(D_SAVER, slotD);
      }
      c.release(3);
      break;
    }
    case GSDF_OFFSET2D: need_children(1); child_dim(true); gen(c, c.child(n, 0), depth + 1); c.op(D_ADDR); c.f(P[0]); break;  // :964-978
    case GSDF_TRANSLATE2D: need_children(1); child_dim(true);                                      // :980-996
      c.op(D_TRANSLATE); c.f(P[0]); c.f(P[1]); c.f(0.f); if (!(P[0] == 0.f && P[1] == 0.f)) c.bump(); gen(c, c.child(n, 0), depth + 1); break;
    case GSDF_SYMMETRY2D: need_children(1); child_dim(true);                                       // :998-1024
      c.op(D_SYMMETRY); c.u((uint32_t)(int)P[0] & 3u); if ((int)P[0] & 3) c.bump(); gen(c, c.child(n, 0), depth + 1); break;
    case GSDF_ANNULUS2D: need_children(1); child_dim(true); gen(c, c.child(n, 0), depth + 1); c.op(D_ANNULUS); c.f(P[0]); break;  // :1026-1040
    case GSDF_TRANSLATEMULTI2D: {                                                                  // :1162-1184
      need_children(1); child_dim(true);
      uint32_t nd = n.aux_len / 2;
      const float* d = &c.t->aux[n.aux_off];
      int slotP = c.alloc(2), slotD = c.alloc(1);
      c.op(D_SAVEP2, slotP);
      c.op(D_SETSLOT, slotD); c.f(3.40282346638528859811704183484516925440e+38f);
      if (nd == 0) { c.op(D_SETR); c.f(3.40282346638528859811704183484516925440e+38f); }
      for (uint32_t k = 0; k < nd; k++) {
        c.op(D_LOADP2_SUB, slotP); c.f(d[2 * k]); c.f(d[2 * k + 1]);
        c.bump();
        gen(c, c.child(n, 0), depth + 1);
        c.op(D_COMBINE_MIN, slotD);
        if (k + 1 < nd) c.op(D_SAVER, slotD);
      }
      c.release(3);
      break;
    }
    case GSDF_ROTATION2D: {                                                                        // :1186-1203
      need_children(1); child_dim(true);
      int lipd = -1;
      {
        float f = lip_norm2(P[0], P[1], P[2], P[3]);
        if (f > kLipRigidTol) { lipd = c.lip_push(); c.op(D_LIP_MUL); c.f(f * kLipRoundUp); }
      }
      c.op(D_ROT2D); c.f(P[0]); c.f(P[1]); c.f(P[2]); c.f(P[3]); c.bump(); gen(c, c.child(n, 0), depth + 1);
      if (lipd >= 0) c.lip_pop(lipd);
      break;
    }
    case GSDF_SCALE2D: need_children(1); child_dim(true);                                          // :1205-1226
      { const int ld = c.lip_push(); c.op(D_SCALE_PRE); c.f(1.f / P[0]); c.bump(); gen(c, c.child(n, 0), depth + 1); c.op(D_MULR); c.f(P[0]); c.lip_pop(ld); } break;
    case GSDF_ELONGATE2D: {                                                                        // :1228-1255
      need_children(1); child_dim(true);
      int s = c.alloc(1);
      c.op(D_ELONGATE2D_PRE, s); c.f(0.5f * P[0]); c.f(0.5f * P[1]);
      c.bump();
      gen(c, c.child(n, 0), depth + 1);
      c.op(D_ADDR_SLOT, s);
      c.release(1);
      break;
    }
    default:
      throw std::runtime_error("unknown op " + std::to_string(n.op));
  }
}

}  // namespace

static void validate(const gsdf_tree& t) {
  if (!t.nodes || t.n_nodes == 0 || t.root >= t.n_nodes) throw std::runtime_error("malformed tree: bad root");
  for (uint32_t i = 0; i < t.n_nodes; i++) {
    const gsdf_node& nd = t.nodes[i];
    if (nd.op == GSDF_OP_INVALID || nd.op >= GSDF_OP_COUNT) throw std::runtime_error("malformed tree: bad op");
    if ((uint64_t)nd.link_off + nd.nchild > t.n_links) throw std::runtime_error("malformed tree: links out of range");
    if ((uint64_t)nd.aux_off + nd.aux_len > t.n_aux) throw std::runtime_error("malformed tree: aux out of range");
    for (uint32_t k = 0; k < nd.nchild; k++)
      if (t.links[nd.link_off + k] >= t.n_nodes) throw std::runtime_error("malformed tree: child out of range");
  }
  // The node graph must be acyclic (shared subtrees are fine): iterative DFS with in-progress marks, so that a blob like
  // node 0 = UNION{0, 0} is refused here instead of recursing without bound in clobbers() / lower_region().
  std::vector<uint8_t> state(t.n_nodes, 0);  // 0 new, 1 on the stack, 2 done
  std::vector<std::pair<uint32_t, uint32_t>> stack;
  stack.push_back({t.root, 0});
  state[t.root] = 1;
  while (!stack.empty()) {
    auto& top = stack.back();
    const gsdf_node& nd = t.nodes[top.first];
    if (top.second < nd.nchild) {
      const uint32_t ch = t.links[nd.link_off + top.second++];
      if (state[ch] == 1) throw std::runtime_error("malformed tree: cycle through node " + std::to_string(ch));
      if (state[ch] == 0) {
        if (stack.size() > 4096) throw std::runtime_error("tree too deep");
        state[ch] = 1;
        stack.push_back({ch, 0});
      }
    } else {
      state[top.first] = 2;
      stack.pop_back();
    }
  }
}

int region_of(const gsdf_tree& t, uint32_t node, float params[8]) {
  validate(t);
  if (node >= t.n_nodes) throw std::runtime_error("node out of range");
  Ctx c;
  c.t = &t;
  c.max_code = 0;
  c.clob.assign(t.n_nodes, -1);
  Region g;
  if (!lower_region(c, node, g)) return 0;
  std::memcpy(params, g.b, sizeof g.b);  // 8 floats
  return (int)g.kind;
}

Program compile(const gsdf_tree& t, size_t max_code_words) {
  validate(t);
  // first run: the candidate operand subtrees of the brick masks, in emission order, with their costs
  std::vector<int> cand_id;
  std::vector<uint64_t> cand_key;
  {
    Ctx c0;
    c0.t = &t;
    c0.max_code = max_code_words;
    c0.clob.assign(t.n_nodes, -1);
    gen(c0, t.root, 0);
    const bool masks_off = [] { const char* e = getenv("GSDF_HIP_NO_BRICK_MASKS"); return e && atoi(e) != 0; }();  // developer knob (A/B timing, cross-check in the tests), read at every compile
    cand_id.assign(c0.cand_cost.size(), -1);
    cand_key = c0.cand_key;
    if (!masks_off) {
      std::vector<size_t> idx(c0.cand_cost.size());
      for (size_t i = 0; i < idx.size(); i++) idx[i] = i;
      std::stable_sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return c0.cand_cost[x] > c0.cand_cost[y]; });
      std::vector<char> take(idx.size(), 0);
      for (size_t i = 0; i < idx.size() && i < 16; i++) take[idx[i]] = 1;
      int next = 0;
      for (size_t i = 0; i < take.size(); i++) if (take[i]) cand_id[i] = next++;
    }
  }
  Ctx c;
  c.t = &t;
  c.max_code = max_code_words;
  c.clob.assign(t.n_nodes, -1);
  c.numbering = true;
  c.cand_id = cand_id;
  c.cand_key = cand_key;
  gen(c, t.root, 0);
  // The numbers were dealt by emission index: both runs must have met the same operand subtrees in the same order. If a decision
  // of the lowering ever came to depend on what the second run emits (a D_SKIP moves the hypot cache's version), the numbers would
  // name other subtrees -- a wrong mesh, silently. Then: lower once more without masks (always right, a third slower).
  if (c.cand_mismatch || c.cand_next != cand_id.size()) {
    bool any = false;
    for (int v : cand_id) any = any || v >= 0;
    if (any) {
      Ctx c2;
      c2.t = &t;
      c2.max_code = max_code_words;
      c2.clob.assign(t.n_nodes, -1);
      c2.numbering = true;
      c2.cand_id.assign(cand_id.size(), -1);
      c2.cand_key = cand_key;
      gen(c2, t.root, 0);
      if (c2.cand_mismatch || c2.cand_next != cand_id.size()) throw std::runtime_error("lowering is not repeatable: candidate subtrees differ between runs");
      c = std::move(c2);
      std::fill(cand_id.begin(), cand_id.end(), -1);
    }
  }
  c.op(D_END);
  for (const Ctx::Table& tb : c.tables) {
    if (tb.n < 1 || tb.n > 4096) continue;  // offset stays 0: the device computes
    c.code[tb.patch] = (uint32_t)c.code.size();
    for (int i = 0; i < tb.n; i++) {
      float sn, cs;
      gsdf::sincosf32(tb.angle * (float)i, sn, cs);
      c.f(sn); c.f(cs);
    }
  }
  Program p;
  p.code = std::move(c.code);
  p.nslots = c.max_slots;
  p.lip_depth = c.max_lip_depth;
  p.n_skip_ids = 0;
  for (int v : cand_id) if (v >= 0) p.n_skip_ids++;
  p.is2d = gsdf_op_is2d(t.nodes[t.root].op);
  p.has_exact_bb = exact_box(c, t.root, p.exact_bb);
  std::memcpy(p.bb, t.bb, sizeof(p.bb));
  return p;
}

}  // namespace gsdf_dev
