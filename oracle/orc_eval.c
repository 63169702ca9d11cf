/*
 * orc_eval.c -- ORACLE (test infrastructure only; see oracle/README.md). PARITY UNPINNED for the
 * un-vendored math32/geometry externals (see orc_math.h); pinned at count level by tests/.
 *
 * CPU restatement of the reference's batch-recursive evaluators:
 *   /root/reference/cpu_evaluators.go:14-1274   one Evaluate method per node type
 *   /root/reference/forge/threads/threads.go:141-202   screw.Evaluate, sawTooth
 *   /root/reference/gleval/cpu.go:92-118,275-317       SDF3CPU.Evaluate, bufPool Acquire/Release
 * Same structure as the reference: every node makes whole-batch passes, temporaries come from a
 * pooled scratch allocator (first free buffer with len >= n, else allocate max(n,minAlloc)).
 */
#include "orc_eval.h"
#include <stdio.h>
#include <stdlib.h>

#include "orc_math.h"

typedef struct { float x, y, z; } V3;
typedef struct { float x, y; } V2;

/* ------------------------------------------------------------------------------------------
 * VecPool  (gleval/cpu.go:188-390)
 * ------------------------------------------------------------------------------------------ */
#define ORC_POOL_MAXBUF 64
typedef struct {
  void* buf[ORC_POOL_MAXBUF];
  size_t len[ORC_POOL_MAXBUF];
  int used[ORC_POOL_MAXBUF];
  int n;
  size_t elem;
} bufpool;

/* Interval bookkeeping of the octree's centre tests (NOT part of the reference). With a context
 * attached, a batch holds every point TWICE: entry 2j carries a lower bound and entry 2j+1 an upper bound of the field
 * over the ball of radius h around point j (the cube whose centre it is). Primitives are exact distances (1-Lipschitz),
 * so they yield value -+ e with e = the radius of the ball's image in their frame; the monotone nodes (min, max, smooth
 * union / intersection, offset, scale ...) act on the two entries unchanged; the others (difference, xor, |d|) cross
 * them, see binop_combine_lip / lip_abs. R[j] = radius of a ball that holds the image of ball j in the frame of the node
 * being evaluated: h at the root, scaled by the scale nodes, stretched by the position maps that are not 1-Lipschitz
 * (twist, screw, non-rigid transform), widened where the image may straddle a seam of a screw's sawtooth.
 * The device evaluates the same thing with two "points" per lane (interp.h: LIP); same float32 operation sequences. */
typedef struct {
  float* R; /* per pair */
} lipctx;
#define LIP_BIG 1e18f /* cap of R: "no bound" (a ball that reaches a screw's axis) without Inf - Inf = NaN downstream */

struct orc_pool {
  bufpool v3, v2, f;
  size_t min_alloc;
  lipctx* lip;
};

orc_pool* orc_pool_create(size_t min_alloc) {
  orc_pool* p = (orc_pool*)calloc(1, sizeof(orc_pool));
  p->v3.elem = sizeof(V3);
  p->v2.elem = sizeof(V2);
  p->f.elem = sizeof(float);
  p->min_alloc = min_alloc;
  return p;
}
static void bufpool_free(bufpool* b) {
  for (int i = 0; i < b->n; i++) free(b->buf[i]);
  b->n = 0;
}
void orc_pool_destroy(orc_pool* p) {
  if (!p) return;
  bufpool_free(&p->v3);
  bufpool_free(&p->v2);
  bufpool_free(&p->f);
  free(p);
}
/* cpu.go:275-293 */
static void* pool_acquire(orc_pool* vp, bufpool* b, size_t n) {
  for (int i = 0; i < b->n; i++) {
    if (!b->used[i] && b->len[i] >= n) { b->used[i] = 1; return b->buf[i]; }
  }
  if (b->n == ORC_POOL_MAXBUF) { fprintf(stderr, "orc: pool exhausted\n"); abort(); }
  size_t m = n > vp->min_alloc ? n : vp->min_alloc;
  void* p = malloc(m * b->elem);
  b->buf[b->n] = p; b->len[b->n] = m; b->used[b->n] = 1; b->n++;
  return p;
}
/* cpu.go:304-317 */
static void pool_release(bufpool* b, void* p) {
  for (int i = 0; i < b->n; i++) {
    if (b->buf[i] == p) {
      if (!b->used[i]) { fprintf(stderr, "orc: double release\n"); abort(); }
      b->used[i] = 0;
      return;
    }
  }
  fprintf(stderr, "orc: release of unknown buffer\n");
  abort();
}
/* cpu.go:218-232 AssertAllReleased */
static int pool_all_released(const orc_pool* vp) {
  const bufpool* bs[3] = {&vp->v3, &vp->v2, &vp->f};
  for (int k = 0; k < 3; k++)
    for (int i = 0; i < bs[k]->n; i++)
      if (bs[k]->used[i]) return 0;
  return 1;
}
#define ACQ_V3(n) ((V3*)pool_acquire(vp, &vp->v3, (n)))
#define ACQ_V2(n) ((V2*)pool_acquire(vp, &vp->v2, (n)))
#define ACQ_F(n) ((float*)pool_acquire(vp, &vp->f, (n)))
#define REL_V3(p) pool_release(&vp->v3, (p))
#define REL_V2(p) pool_release(&vp->v2, (p))
#define REL_F(p) pool_release(&vp->f, (p))

/* ------------------------------------------------------------------------------------------
 * tree
 * ------------------------------------------------------------------------------------------ */
struct orc_sdf {
  gsdf_node* nodes;
  uint32_t n_nodes;
  uint32_t* links;
  uint32_t n_links;
  float* aux;
  uint32_t n_aux;
  uint32_t root;
  float bb[6];
};

orc_sdf* orc_sdf_create(const gsdf_tree* t) {
  if (!t || !t->nodes || t->n_nodes == 0 || t->root >= t->n_nodes) return NULL;
  for (uint32_t i = 0; i < t->n_nodes; i++) {
    const gsdf_node* nd = &t->nodes[i];
    if (nd->op == GSDF_OP_INVALID || nd->op >= GSDF_OP_COUNT) return NULL;
    if ((uint64_t)nd->link_off + nd->nchild > t->n_links) return NULL;
    if ((uint64_t)nd->aux_off + nd->aux_len > t->n_aux) return NULL;
    for (uint32_t c = 0; c < nd->nchild; c++)
      if (t->links[nd->link_off + c] >= t->n_nodes) return NULL;
  }
  orc_sdf* s = (orc_sdf*)calloc(1, sizeof(orc_sdf));
  s->n_nodes = t->n_nodes; s->n_links = t->n_links; s->n_aux = t->n_aux; s->root = t->root;
  s->nodes = (gsdf_node*)malloc(sizeof(gsdf_node) * t->n_nodes);
  memcpy(s->nodes, t->nodes, sizeof(gsdf_node) * t->n_nodes);
  s->links = (uint32_t*)malloc(sizeof(uint32_t) * (t->n_links ? t->n_links : 1));
  if (t->n_links) memcpy(s->links, t->links, sizeof(uint32_t) * t->n_links);
  s->aux = (float*)malloc(sizeof(float) * (t->n_aux ? t->n_aux : 1));
  if (t->n_aux) memcpy(s->aux, t->aux, sizeof(float) * t->n_aux);
  memcpy(s->bb, t->bb, sizeof(s->bb));
  return s;
}
void orc_sdf_destroy(orc_sdf* s) {
  if (!s) return;
  free(s->nodes); free(s->links); free(s->aux); free(s);
}
void orc_sdf_bounds(const orc_sdf* s, float bb[6]) { memcpy(bb, s->bb, sizeof(float) * 6); }
int orc_sdf_root_is2d(const orc_sdf* s) { return gsdf_op_is2d(s->nodes[s->root].op); }

/* ms3/ms2 helpers [external soypat/geometry, restated] */
static inline float norm3(V3 p) { return go_hypotf(p.x, go_hypotf(p.y, p.z)); }
static inline float norm2(V2 p) { return go_hypotf(p.x, p.y); }
static inline float dot2(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
static inline float norm2sq(V2 a) { return a.x * a.x + a.y * a.y; }
static inline V2 sub2(V2 a, V2 b) { V2 r = {a.x - b.x, a.y - b.y}; return r; }
static inline V2 add2(V2 a, V2 b) { V2 r = {a.x + b.x, a.y + b.y}; return r; }
static inline V2 scale2(float f, V2 a) { V2 r = {f * a.x, f * a.y}; return r; }
static inline float cross2(V2 a, V2 b) { return a.x * b.y - a.y * b.x; }
static inline float ms1_clamp(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline float ms1_sign(float a) { return a == 0 ? 0.0f : go_copysignf(1.0f, a); }

/* ---- stretch factors: rho = distance of the centre from the node's z axis, rl = radius of the ball's image there.
 *   twist  (x,y) -> R(k z)(x,y): in the (radial, tangential, axial) frame the Jacobian is a shear by s = |k| rho in the
 *          (tangential, axial) plane; its spectral norm is (s + sqrt(s^2 + 4)) / 2, largest at the largest rho.
 *   screw  (threads.go:141-181) x' = saw(z + lead theta / 2pi), y' = rho + z tanT: rows (0, a, 1) and (1, 0, t) in that
 *          frame with a = |lead| / (2 pi rho); J J^T = [[1 + a^2, t], [t, 1 + t^2]], largest eigenvalue
 *          ((2 + a^2 + t^2) + sqrt((a^2 - t^2)^2 + 4 t^2)) / 2, largest at the smallest rho; unbounded on the axis. */
static inline float lip_twist(float rho, float rl, float ak) {
  float s = ak * (rho + rl);
  return 0.5f * (s + sqrtf(s * s + 4.0f));
}
static inline float lip_screw(float rho, float rl, float alead, float t) {
  float rmin = rho - rl;
  if (!(rmin > 0.0f)) return LIP_BIG;
  float a = alead / (6.2831855f * rmin);
  float a2 = a * a, t2 = t * t, dd = a2 - t2;
  return sqrtf(0.5f * ((2.0f + a2 + t2) + sqrtf(dd * dd + 4.0f * t2)));
}
/* largest singular value of the 3x3 linear part of a row-major 4x4 (cyclic Jacobi on A^T A, fixed sweeps, double) */
static float lip_norm3(const float* m) {
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
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < 3; k++) { double x = b[k][p], y = b[k][q]; b[k][p] = c * x - sn * y; b[k][q] = sn * x + c * y; }
        for (int k = 0; k < 3; k++) { double x = b[p][k], y = b[q][k]; b[p][k] = c * x - sn * y; b[q][k] = sn * x + c * y; }
      }
  double e = b[0][0] > b[1][1] ? b[0][0] : b[1][1];
  if (b[2][2] > e) e = b[2][2];
  return (float)sqrt(e);
}
static float lip_norm2(float a, float b, float c, float d) { /* 2x2 [[a b][c d]] */
  double S = (double)a * a + (double)b * b + (double)c * c + (double)d * d, D = (double)a * d - (double)b * c;
  double disc = S * S - 4.0 * D * D;
  if (disc < 0) disc = 0;
  return (float)sqrt(0.5 * (S + sqrt(disc)));
}
/* a linear map whose largest singular value is at most 1 shrinks or keeps the ball (factor 1); anything above -- also the
 * 1.0000001 of a rotation built from rounded sines -- stretches it, and the factor is rounded up */
#define LIP_RIGID_TOL 1.0f
#define LIP_ROUND_UP 1.000001f

/* radius of the ball's image in the current frame, pair j */
static inline float lip_radius(const lipctx* lc, size_t j) { return lc->R[j]; }
/* (a negative factor mirrors the ball: its radius scales by the magnitude) */
static void lip_scale(lipctx* lc, size_t n, float f) { for (size_t j = 0; j < n / 2; j++) lc->R[j] = lc->R[j] * fabsf(f); }
/* value -+ radius: an exact distance (or any 1-Lipschitz term of the current frame) over the ball */
static void lip_widen(const lipctx* lc, float* d, size_t n) {
  for (size_t j = 0; j < n / 2; j++) {
    float e = lip_radius(lc, j);
    d[2 * j] = d[2 * j] - e;
    d[2 * j + 1] = d[2 * j + 1] + e;
  }
}
/* |x| of an interval */
static inline void lip_abs(float lo, float hi, float* alo, float* ahi) {
  *alo = go_maxf(go_maxf(lo, -hi), 0.0f);
  *ahi = go_maxf(-lo, hi);
}
/* enter / leave a node whose position map stretches by f around pair j */
static float* lip_enter(orc_pool* vp, size_t n) {
  float* save = (float*)pool_acquire(vp, &vp->f, n / 2);
  memcpy(save, vp->lip->R, (n / 2) * sizeof(float));
  return save;
}
static inline void lip_stretch(lipctx* lc, size_t j, float f) { lc->R[j] = go_minf(lc->R[j] * f, LIP_BIG); }
static void lip_leave(orc_pool* vp, float* save, size_t n) {
  memcpy(vp->lip->R, save, (n / 2) * sizeof(float));
  pool_release(&vp->f, save);
}

/* the scale nodes' d * f on an interval (f of either sign) */
static void lip_mul(float* d, size_t n, float f) {
  for (size_t j = 0; j < n / 2; j++) {
    float x0 = d[2 * j] * f, x1 = d[2 * j + 1] * f;
    d[2 * j] = go_minf(x0, x1);
    d[2 * j + 1] = go_maxf(x0, x1);
  }
}

static const float TRIBISECT = 0.8660254037844386467637231707529361834714026269051903140279034897f;
static const float SQRT3 = 1.7320508075688772935274463415058723669428052538103806280558069794f;
static const float LARGENUM = 1e20f;

/* How much a screw's field may jump across a seam of its sawtooth. The profile's field at (+pitch/2, y) and at
 * (-pitch/2, y) differ by less than the distance between the two points, the pitch. If the profile is a polygon that is
 * its own mirror image in x up to d (vertex i of the mirror image within d, per coordinate, of vertex k - i of the
 * polygon, for some k: the same boundary traversed the other way round), the two boundaries lie within sqrt2 d of each
 * other and so do their distance fields: 3 d. The thread forms with rounded roots (ISO, NPT) come out of PolygonBuilder.Smooth
 * symmetric to a few 1e-7; buttress forms are not symmetric at all. Same float32 sequence as compile.cpp: lip_screw_seam. */
static float lip_screw_seam(const orc_sdf* s, uint32_t child, float pitch) {
  const gsdf_node* c = &s->nodes[child];
  const float ap = go_absf(pitch);
  if (c->op != GSDF_POLY2D) return ap;
  const float* v = &s->aux[c->aux_off];
  const uint32_t nv = c->aux_len / 2;
  float best = ap;
  for (uint32_t k = 0; k < nv; k++) {
    float dk = 0.0f;
    for (uint32_t i = 0; i < nv; i++) {
      const uint32_t w = (k + nv - i) % nv;
      const float dx = go_absf(-v[2 * i] - v[2 * w]), dy = go_absf(v[2 * i + 1] - v[2 * w + 1]);
      if (dx > dk) dk = dx;
      if (dy > dk) dk = dy;
    }
    if (dk < best) best = dk;
  }
  return best <= 1e-3f * ap ? 3.0f * best : ap;
}
static int eval3_node(const orc_sdf* s, uint32_t ni, const V3* pos, float* dist, size_t n, orc_pool* vp);
static int eval2_node(const orc_sdf* s, uint32_t ni, const V2* pos, float* dist, size_t n, orc_pool* vp);
/* the reference's Evaluate of node ni; with an interval context attached, a primitive's distances become bounds */
static int eval3(const orc_sdf* s, uint32_t ni, const V3* pos, float* dist, size_t n, orc_pool* vp) {
  int err = eval3_node(s, ni, pos, dist, n, vp);
  if (!err && vp->lip && s->nodes[ni].op >= GSDF_SPHERE && s->nodes[ni].op <= GSDF_HEX) lip_widen(vp->lip, dist, n);
  return err;
}
static int eval2(const orc_sdf* s, uint32_t ni, const V2* pos, float* dist, size_t n, orc_pool* vp) {
  int err = eval2_node(s, ni, pos, dist, n, vp);
  if (!err && vp->lip && s->nodes[ni].op >= GSDF_LINE2D && s->nodes[ni].op <= GSDF_LINES2D) lip_widen(vp->lip, dist, n);
  return err;
}

#define CHILD(nd, k) (s->links[(nd)->link_off + (k)])

/* cpu_evaluators.go:14-18 */
static void min_reduce(float* d1_and_dst, const float* d2, size_t n) {
  for (size_t i = 0; i < n; i++) d1_and_dst[i] = go_minf(d1_and_dst[i], d2[i]);
}

/* generic binary op frame: cpu_evaluators.go:146-286 (3D) and :847-912 (2D) */
static inline float smooth_diff(float a, float b, float k) {
  float h = orc_clampf(0.5f - 0.5f * (b + a) / k, 0, 1);
  return orc_mixf(a, -b, h) + k * h * (1 - h);
}
/* interval forms of the combines that DEcrease in their second operand (entries 2j / 2j+1 = lower / upper bound) */
static void binop_combine_lip(int op, float k, float* dist, const float* d2, size_t n) {
  for (size_t j = 0; j < n / 2; j++) {
    float alo = dist[2 * j], ahi = dist[2 * j + 1], blo = d2[2 * j], bhi = d2[2 * j + 1];
    switch (op) {
      case GSDF_DIFF: case GSDF_DIFF2D:
        dist[2 * j] = go_maxf(alo, -bhi);
        dist[2 * j + 1] = go_maxf(ahi, -blo);
        break;
      case GSDF_XOR: case GSDF_XOR2D:
        dist[2 * j] = go_maxf(go_minf(alo, blo), -go_maxf(ahi, bhi));
        dist[2 * j + 1] = go_maxf(go_minf(ahi, bhi), -go_maxf(alo, blo));
        break;
      default: /* GSDF_SMOOTH_DIFF */
        dist[2 * j] = smooth_diff(alo, bhi, k);
        dist[2 * j + 1] = smooth_diff(ahi, blo, k);
        break;
    }
  }
}
static void binop_combine(int op, float k, float* dist, const float* d2, size_t n) {
  switch (op) {
    case GSDF_INTERSECT: case GSDF_INTERSECT2D:
      for (size_t i = 0; i < n; i++) dist[i] = go_maxf(dist[i], d2[i]);
      break;
    case GSDF_DIFF: case GSDF_DIFF2D:
      for (size_t i = 0; i < n; i++) dist[i] = go_maxf(dist[i], -d2[i]);
      break;
    case GSDF_XOR: case GSDF_XOR2D:
      for (size_t i = 0; i < n; i++) {
        float a = dist[i], b = d2[i];
        dist[i] = go_maxf(go_minf(a, b), -go_maxf(a, b));
      }
      break;
    case GSDF_SMOOTH_UNION:
      for (size_t i = 0; i < n; i++) {
        float a = dist[i], b = d2[i];
        float h = orc_clampf(0.5f + 0.5f * (b - a) / k, 0, 1);
        dist[i] = orc_mixf(b, a, h) - k * h * (1 - h);
      }
      break;
    case GSDF_SMOOTH_DIFF:
      for (size_t i = 0; i < n; i++) {
        float a = dist[i], b = d2[i];
        float h = orc_clampf(0.5f - 0.5f * (b + a) / k, 0, 1);
        dist[i] = orc_mixf(a, -b, h) + k * h * (1 - h);
      }
      break;
    case GSDF_SMOOTH_INTERSECT:
      for (size_t i = 0; i < n; i++) {
        float a = dist[i], b = d2[i];
        float h = orc_clampf(0.5f - 0.5f * (b - a) / k, 0, 1);
        dist[i] = orc_mixf(b, a, h) + k * h * (1 - h);
      }
      break;
  }
}

static int eval3_node(const orc_sdf* s, uint32_t ni, const V3* pos, float* dist, size_t n, orc_pool* vp) {
  const gsdf_node* nd = &s->nodes[ni];
  const float* P = nd->p;
  int err = 0;
  switch (nd->op) {
    case GSDF_SPHERE: { /* :20-26 */
      float r = P[0];
      for (size_t i = 0; i < n; i++) dist[i] = norm3(pos[i]) - r;
      return 0;
    }
    case GSDF_BOX: { /* :28-36 */
      V3 d = {0.5f * P[0], 0.5f * P[1], 0.5f * P[2]};
      float r = P[3];
      for (size_t i = 0; i < n; i++) {
        V3 p = pos[i];
        V3 q = {(go_absf(p.x) - d.x) + r, (go_absf(p.y) - d.y) + r, (go_absf(p.z) - d.z) + r};
        V3 qm = {go_maxf(q.x, 0), go_maxf(q.y, 0), go_maxf(q.z, 0)};
        dist[i] = norm3(qm) + go_minf(go_maxf(q.x, go_maxf(q.y, q.z)), 0.0f) - r;
      }
      return 0;
    }
    case GSDF_BOXFRAME: { /* :38-57 ; args(): primitives.go:292-297 */
      float e = P[3];
      V3 b = {0.5f * P[0] + (-2 * e), 0.5f * P[1] + (-2 * e), 0.5f * P[2] + (-2 * e)};
      for (size_t i = 0; i < n; i++) {
        V3 p = pos[i];
        p.x = go_absf(p.x) - b.x; p.y = go_absf(p.y) - b.y; p.z = go_absf(p.z) - b.z;
        V3 q = {go_absf(p.x + e) + (-e), go_absf(p.y + e) + (-e), go_absf(p.z + e) + (-e)};
        float s1 = go_minf(0, go_maxf(p.x, go_maxf(q.y, q.z)));
        V3 a1 = {go_maxf(p.x, 0), go_maxf(q.y, 0), go_maxf(q.z, 0)};
        float n1 = norm3(a1) + s1;
        float s2 = go_minf(0, go_maxf(q.x, go_maxf(p.y, q.z)));
        V3 a2 = {go_maxf(q.x, 0), go_maxf(p.y, 0), go_maxf(q.z, 0)};
        float n2 = norm3(a2) + s2;
        float s3 = go_minf(0, go_maxf(q.x, go_maxf(q.y, p.z)));
        V3 a3 = {go_maxf(q.x, 0), go_maxf(q.y, 0), go_maxf(p.z, 0)};
        float n3 = norm3(a3) + s3;
        dist[i] = go_minf(n1, go_minf(n2, n3));
      }
      return 0;
    }
    case GSDF_TORUS: { /* :59-68 */
      float t1 = P[0], t2 = P[1];
      for (size_t i = 0; i < n; i++) {
        V3 p = {pos[i].x, pos[i].z, pos[i].y};
        V2 q = {go_hypotf(p.x, p.z) - t1, p.y};
        dist[i] = norm2(q) - t2;
      }
      return 0;
    }
    case GSDF_CYLINDER: { /* :70-88 ; args(): primitives.go:147-149 */
      float r = P[0], h = (P[1] - 2 * P[2]) / 2, round = P[2];
      if (round == 0) {
        for (size_t i = 0; i < n; i++) {
          V3 p = {pos[i].x, pos[i].z, pos[i].y};
          float dx = go_hypotf(p.x, p.z) - r;
          float dy = go_absf(p.y) - h;
          dist[i] = go_minf(0, go_maxf(dx, dy)) + go_hypotf(go_maxf(0, dx), go_maxf(0, dy));
        }
      } else {
        for (size_t i = 0; i < n; i++) {
          V3 p = {pos[i].x, pos[i].z, pos[i].y};
          float dx = go_hypotf(p.x, p.z) - r + round;
          float dy = go_absf(p.y) - h;
          dist[i] = go_minf(go_maxf(dx, dy), 0) + go_hypotf(go_maxf(dx, 0), go_maxf(dy, 0)) - round;
        }
      }
      return 0;
    }
    case GSDF_HEX: { /* :90-105 */
      const float k1 = -TRIBISECT, k2 = 0.5f, k3 = 0.57735f;
      const float twok1 = (float)(2 * -0.8660254037844386467637231707529361834714026269051903140279034897);
      float h1 = P[0], h2 = P[1];
      float clm = k3 * h1;
      for (size_t i = 0; i < n; i++) {
        V3 p = {go_absf(pos[i].x), go_absf(pos[i].y), go_absf(pos[i].z)};
        float pm = go_minf(k1 * p.x + k2 * p.y, 0);
        p.x -= twok1 * pm;
        p.y -= 1.0f * pm;
        float d1 = go_hypotf(p.x - orc_clampf(p.x, -clm, clm), p.y - h1) * orc_signf(p.y - h1);
        float d2 = p.z - h2;
        dist[i] = go_minf(go_maxf(d1, d2), 0) + go_hypotf(go_maxf(d1, 0), go_maxf(d2, 0));
      }
      return 0;
    }
    case GSDF_UNION: { /* :124-144 */
      if (nd->nchild < 2) return -2;
      float* aux = ACQ_F(n);
      err = eval3(s, CHILD(nd, 0), pos, dist, n, vp);
      for (uint32_t c = 1; c < nd->nchild && !err; c++) {
        err = eval3(s, CHILD(nd, c), pos, aux, n, vp);
        if (!err) min_reduce(dist, aux, n);
      }
      REL_F(aux);
      return err;
    }
    case GSDF_INTERSECT: case GSDF_DIFF: case GSDF_XOR:
    case GSDF_SMOOTH_UNION: case GSDF_SMOOTH_DIFF: case GSDF_SMOOTH_INTERSECT: { /* :146-286 */
      if (nd->nchild != 2) return -2;
      float* d2 = ACQ_F(n);
      err = eval3(s, CHILD(nd, 0), pos, dist, n, vp);
      if (!err) err = eval3(s, CHILD(nd, 1), pos, d2, n, vp);
      if (!err) {
        if (vp->lip && (nd->op == GSDF_DIFF || nd->op == GSDF_XOR || nd->op == GSDF_SMOOTH_DIFF)) binop_combine_lip(nd->op, P[0], dist, d2, n);
        else binop_combine(nd->op, P[0], dist, d2, n);
      }
      REL_F(d2);
      return err;
    }
    case GSDF_SCALE: { /* :288-312 */
      V3* sc = ACQ_V3(n);
      float factor = P[0];
      float inv = 1.f / P[0];
      for (size_t i = 0; i < n; i++) { sc[i].x = inv * pos[i].x; sc[i].y = inv * pos[i].y; sc[i].z = inv * pos[i].z; }
      float* lsave = vp->lip ? lip_enter(vp, n) : NULL; /* the outer radius comes back exactly (not (R * inv) * factor) */
      if (vp->lip) lip_scale(vp->lip, n, inv);
      err = eval3(s, CHILD(nd, 0), sc, dist, n, vp);
      if (vp->lip) {
        lip_leave(vp, lsave, n);
        if (!err) lip_mul(dist, n, factor);
      } else
      if (!err) for (size_t i = 0; i < n; i++) dist[i] *= factor;
      REL_V3(sc);
      return err;
    }
    case GSDF_SYMMETRY: { /* :314-343 */
      V3* t = ACQ_V3(n);
      memcpy(t, pos, n * sizeof(V3));
      int bits = (int)P[0];
      for (size_t i = 0; i < n; i++) {
        if (bits & 1) t[i].x = go_absf(t[i].x);
        if (bits & 2) t[i].y = go_absf(t[i].y);
        if (bits & 4) t[i].z = go_absf(t[i].z);
      }
      err = eval3(s, CHILD(nd, 0), t, dist, n, vp);
      REL_V3(t);
      return err;
    }
    case GSDF_ARRAY: { /* :345-397 */
      V3* t = ACQ_V3(n);
      float* aux = ACQ_F(n);
      V3 sp = {P[0], P[1], P[2]};
      V3 nn = {P[3] + -1, P[4] + -1, P[5] + -1};
      for (size_t i = 0; i < n; i++) dist[i] = LARGENUM;
      for (int k = 0; k < 2 && !err; k++)
        for (int j = 0; j < 2 && !err; j++)
          for (int ii = 0; ii < 2 && !err; ii++) {
            V3 ijk = {(float)ii, (float)j, (float)k};
            for (size_t ip = 0; ip < n; ip++) {
              V3 p = pos[ip];
              V3 id = {go_roundf(p.x / sp.x), go_roundf(p.y / sp.y), go_roundf(p.z / sp.z)};
              V3 o = {ms1_sign(p.x - sp.x * id.x), ms1_sign(p.y - sp.y * id.y), ms1_sign(p.z - sp.z * id.z)};
              V3 rid = {id.x + ijk.x * o.x, id.y + ijk.y * o.y, id.z + ijk.z * o.z};
              rid.x = ms1_clamp(rid.x, 0, nn.x); rid.y = ms1_clamp(rid.y, 0, nn.y); rid.z = ms1_clamp(rid.z, 0, nn.z);
              t[ip].x = p.x - sp.x * rid.x; t[ip].y = p.y - sp.y * rid.y; t[ip].z = p.z - sp.z * rid.z;
            }
            err = eval3(s, CHILD(nd, 0), t, aux, n, vp);
            if (!err) for (size_t i = 0; i < n; i++) dist[i] = go_minf(dist[i], aux[i]);
          }
      REL_F(aux);
      REL_V3(t);
      return err;
    }
    case GSDF_ELONGATE: { /* :399-426 */
      V3* t = ACQ_V3(n);
      float* aux = ACQ_F(n);
      V3 h = {0.5f * P[0], 0.5f * P[1], 0.5f * P[2]};
      for (size_t i = 0; i < n; i++) {
        V3 q = {go_absf(pos[i].x) - h.x, go_absf(pos[i].y) - h.y, go_absf(pos[i].z) - h.z};
        aux[i] = go_minf(go_maxf(q.x, go_maxf(q.y, q.z)), 0);
        t[i].x = go_maxf(q.x, 0); t[i].y = go_maxf(q.y, 0); t[i].z = go_maxf(q.z, 0);
      }
      if (vp->lip) lip_widen(vp->lip, aux, n); /* min(max3(q), 0) is 1-Lipschitz in this frame */
      err = eval3(s, CHILD(nd, 0), t, dist, n, vp);
      if (!err) for (size_t i = 0; i < n; i++) dist[i] += aux[i];
      REL_F(aux);
      REL_V3(t);
      return err;
    }
    case GSDF_SHELL: { /* :428-452 */
      V3* t = ACQ_V3(n);
      float th = P[0];
      for (size_t i = 0; i < n; i++) { float f = 1 / th; t[i].x = f * pos[i].x; t[i].y = f * pos[i].y; t[i].z = f * pos[i].z; }
      float* lsave = vp->lip ? lip_enter(vp, n) : NULL;
      if (vp->lip) lip_scale(vp->lip, n, 1 / th);
      err = eval3(s, CHILD(nd, 0), t, dist, n, vp);
      if (vp->lip) {
        lip_leave(vp, lsave, n);
        if (!err)
          for (size_t j = 0; j < n / 2; j++) {
            float alo, ahi;
            lip_abs(dist[2 * j], dist[2 * j + 1], &alo, &ahi);
            float x0 = th * (alo - th), x1 = th * (ahi - th);
            dist[2 * j] = go_minf(x0, x1);
            dist[2 * j + 1] = go_maxf(x0, x1);
          }
      } else
      if (!err) for (size_t i = 0; i < n; i++) dist[i] = th * (go_absf(dist[i]) - th);
      REL_V3(t);
      return err;
    }
    case GSDF_OFFSET: { /* :454-468 */
      err = eval3(s, CHILD(nd, 0), pos, dist, n, vp);
      if (!err) for (size_t i = 0; i < n; i++) dist[i] = dist[i] + P[0];
      return err;
    }
    case GSDF_TRANSLATE: { /* :470-486 */
      V3* t = ACQ_V3(n);
      for (size_t i = 0; i < n; i++) { t[i].x = pos[i].x - P[0]; t[i].y = pos[i].y - P[1]; t[i].z = pos[i].z - P[2]; }
      err = eval3(s, CHILD(nd, 0), t, dist, n, vp);
      REL_V3(t);
      return err;
    }
    case GSDF_TRANSFORM: { /* :488-504 ; Mat4.MulPosition [external] */
      if (nd->aux_len < 16) return -2;
      const float* m = &s->aux[nd->aux_off];
      V3* t = ACQ_V3(n);
      for (size_t i = 0; i < n; i++) {
        V3 v = pos[i];
        t[i].x = m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3];
        t[i].y = m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7];
        t[i].z = m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11];
      }
      float* vsave = NULL;
      if (vp->lip) {
        float f = lip_norm3(m);
        if (f > LIP_RIGID_TOL) {
          f = f * LIP_ROUND_UP;
          vsave = lip_enter(vp, n);
          for (size_t j = 0; j < n / 2; j++) lip_stretch(vp->lip, j, f);
        }
      }
      err = eval3(s, CHILD(nd, 0), t, dist, n, vp);
      if (vsave) lip_leave(vp, vsave, n);
      REL_V3(t);
      return err;
    }
    case GSDF_CIRCARRAY: { /* :1042-1092 */
      V3* pos0 = ACQ_V3(n);
      V3* pos1 = ACQ_V3(n);
      float angle = (float)(2 * ORC_PI) / P[1];
      float ncirc = P[1];
      float ninsm1 = (float)((int)P[0] - 1);
      for (size_t i = 0; i < n; i++) {
        V3 p = pos[i];
        float pangle = go_atan2f(p.y, p.x);
        float id = go_floorf(pangle / angle);
        if (id < 0) id += ncirc;
        float i0, i1;
        if (id >= ninsm1) { i0 = ninsm1; i1 = 0; } else { i0 = id; i1 = id + 1; }
        float s0, c0, s1, c1;
        go_sincosf(angle * i0, &s0, &c0);
        go_sincosf(angle * i1, &s1, &c1);
        /* MulMatVecTrans(RotationMat2(a), p): x = c*x + s*y ; y = -s*x + c*y */
        pos0[i].x = c0 * p.x + s0 * p.y; pos0[i].y = (-s0) * p.x + c0 * p.y; pos0[i].z = p.z;
        pos1[i].x = c1 * p.x + s1 * p.y; pos1[i].y = (-s1) * p.x + c1 * p.y; pos1[i].z = p.z;
      }
      float* dist1 = ACQ_F(n);
      err = eval3(s, CHILD(nd, 0), pos1, dist1, n, vp);
      if (!err) err = eval3(s, CHILD(nd, 0), pos0, dist, n, vp);
      if (!err) min_reduce(dist, dist1, n);
      REL_F(dist1);
      REL_V3(pos1);
      REL_V3(pos0);
      return err;
    }
    case GSDF_TWIST: { /* :1257-1274 */
      V3* t = ACQ_V3(n);
      float k = P[0];
      for (size_t i = 0; i < n; i++) {
        V3 p = pos[i];
        float c = go_cosf(k * p.z);
        float sn = go_sinf(k * p.z);
        t[i].x = c * p.x - sn * p.y; t[i].y = sn * p.x + c * p.y; t[i].z = p.z;
      }
      float* vsave = NULL;
      if (vp->lip) {
        lipctx* lc = vp->lip;
        vsave = lip_enter(vp, n);
        for (size_t j = 0; j < n / 2; j++) lip_stretch(lc, j, lip_twist(go_hypotf(pos[2 * j].x, pos[2 * j].y), lip_radius(lc, j), go_absf(k)));
      }
      err = eval3(s, CHILD(nd, 0), t, dist, n, vp);
      if (vsave) lip_leave(vp, vsave, n);
      REL_V3(t);
      return err;
    }
    case GSDF_EXTRUSION: { /* :506-531 */
      V2* p2 = ACQ_V2(n);
      for (size_t i = 0; i < n; i++) { p2[i].x = pos[i].x; p2[i].y = pos[i].y; }
      err = eval2(s, CHILD(nd, 0), p2, dist, n, vp);
      if (!err) {
        float h = P[0] / 2;
        for (size_t i = 0; i < n; i++) {
          float d = dist[i];
          float wy = go_absf(pos[i].z) - h;
          if (vp->lip) { float e = lip_radius(vp->lip, i / 2); wy = (i & 1) ? wy + e : wy - e; } /* increasing in d and in wy */
          dist[i] = go_minf(0, go_maxf(d, wy)) + go_hypotf(go_maxf(d, 0), go_maxf(wy, 0));
        }
      }
      REL_V2(p2);
      return err;
    }
    case GSDF_REVOLUTION: { /* :533-549 */
      V2* p2 = ACQ_V2(n);
      float o = P[0];
      for (size_t i = 0; i < n; i++) { p2[i].x = go_hypotf(pos[i].x, pos[i].z) - o; p2[i].y = pos[i].y; }
      err = eval2(s, CHILD(nd, 0), p2, dist, n, vp);
      REL_V2(p2);
      return err;
    }
    case GSDF_SCREW: { /* forge/threads/threads.go:141-181, sawTooth :198-202 */
      V2* tr = ACQ_V2(n);
      float pitch = P[0], lead = P[1], L = P[2], taper = P[3];
      float tanTaper = go_tanf(taper);
      const float twopi = (float)(2 * ORC_PI);
      for (size_t i = 0; i < n; i++) {
        V3 p = pos[i];
        V2 p0;
        p0.y = go_hypotf(p.x, p.y);
        p0.y += p.z * tanTaper;
        float theta = go_atan2f(p.y, p.x);
        float z = p.z + lead * theta / twopi;
        float x = z + pitch / 2;
        float t = x / pitch;
        p0.x = pitch * (t - go_floorf(t)) - pitch / 2;
        tr[i] = p0;
      }
      float* vsave = NULL;
      if (vp->lip) {
        lipctx* lc = vp->lip;
        vsave = lip_enter(vp, n);
        /* The profile is evaluated at the sawtooth of the axial coordinate: ONE period of it. Across a seam of the
         * sawtooth the field is continuous only if the profile's field is the same at x = -pitch/2 and x = +pitch/2 (a
         * profile symmetric in x: ISO, NPT, the knurl); an asymmetric one (buttress threads) jumps there, by less than the
         * distance between the two points, i.e. the pitch. A ball whose image may reach a seam is widened by that much. */
        float seam = lip_screw_seam(s, CHILD(nd, 0), pitch);
        for (size_t j = 0; j < n / 2; j++) {
          lip_stretch(lc, j, lip_screw(go_hypotf(pos[2 * j].x, pos[2 * j].y), lip_radius(lc, j), go_absf(lead), tanTaper));
          if (go_absf(tr[2 * j].x) + lc->R[j] >= pitch / 2) lc->R[j] = lc->R[j] + seam;
        }
      }
      err = eval2(s, CHILD(nd, 0), tr, dist, n, vp);
      if (vsave) lip_leave(vp, vsave, n);
      if (!err)
        for (size_t i = 0; i < n; i++) {
          float d0 = dist[i];
          float d1 = go_absf(pos[i].z) - L;
          if (vp->lip) { float e = lip_radius(vp->lip, i / 2); d1 = (i & 1) ? d1 + e : d1 - e; } /* |z| - L in the screw's own frame */
          dist[i] = go_maxf(d0, d1);
        }
      REL_V2(tr);
      return err;
    }
    default:
      return -3; /* 2D node where a 3D node is required, or unknown op */
  }
}

static int eval2_node(const orc_sdf* s, uint32_t ni, const V2* pos, float* dist, size_t n, orc_pool* vp) {
  const gsdf_node* nd = &s->nodes[ni];
  const float* P = nd->p;
  int err = 0;
  switch (nd->op) {
    case GSDF_LINE2D: { /* :551-562 */
      V2 a = {P[0], P[1]}, b = {P[2], P[3]};
      V2 ba = sub2(b, a);
      float dotba = dot2(ba, ba);
      float w = P[4] / 2;
      for (size_t i = 0; i < n; i++) {
        V2 pa = sub2(pos[i], a);
        float h = ms1_clamp(dot2(pa, ba) / dotba, 0, 1);
        dist[i] = norm2(sub2(pa, scale2(h, ba))) - w;
      }
      return 0;
    }
    case GSDF_ARC2D: { /* :564-579 */
      float r = P[0], t = P[2] / 2;
      float sn, cs;
      go_sincosf(P[1] / 2, &sn, &cs);
      V2 sc = {sn, cs};
      V2 scr = scale2(r, sc);
      for (size_t i = 0; i < n; i++) {
        V2 p = pos[i];
        p.x = go_absf(p.x);
        if (sc.y * p.x > sc.x * p.y) dist[i] = norm2(sub2(p, scr)) - t;
        else dist[i] = go_absf(norm2(p) - r) - t;
      }
      return 0;
    }
    case GSDF_QUADBEZIER2D: { /* :581-659 */
      float thick = P[6] / 2;
      V2 A = {P[0], P[1]}, B = {P[2], P[3]}, C = {P[4], P[5]};
      V2 a = sub2(B, A);
      float a2 = dot2(a, a);
      V2 b = add2(A, sub2(C, scale2(2, B)));
      V2 c = scale2(2, a);
      float kk = 1.f / dot2(b, b);
      float kx = kk * dot2(a, b);
      float kx2 = kx * kx;
      for (size_t i = 0; i < n; i++) {
        V2 p = pos[i];
        V2 d = sub2(A, p);
        float ky = kk * (2 * a2 + dot2(d, b)) / 3;
        float kz = kk * dot2(d, a);
        float g = ky - kx2;
        float q = kx * (2 * kx2 - 3 * ky) + kz;
        float g3 = g * g * g;
        float q2 = q * q;
        float h = q2 + 4 * g3;
        float res;
        if (h >= 0) {
          h = go_sqrtf(h);
          V2 x = {0.5f * (h + -q), 0.5f * (-h + -q)};
          if (go_absf(g) < 0.001f) {
            float k = (1.0f - g3 / q2) * g3 / q;
            x.x = k; x.y = -k - q;
          }
          V2 uv = {ms1_sign(x.x) * go_pow13f(go_absf(x.x)), ms1_sign(x.y) * go_pow13f(go_absf(x.y))};
          float t = uv.x + uv.y;
          t -= (t * (t * t + 3.0f * g) + q) / (3.0f * t * t + 3.0f * g);
          t = ms1_clamp(t - kx, 0, 1);
          V2 w = add2(d, scale2(t, add2(c, scale2(t, b))));
          res = dot2(w, w);
        } else {
          float z = go_sqrtf(-g);
          float xx = q / (2 * g * z);
          xx = go_sqrtf(0.5f + 0.5f * xx); /* cos_acos_3, gsdf.go:186-189 */
          float m = xx * (xx * (xx * (xx * -0.008972f + 0.039071f) - 0.107074f) + 0.576975f) + 0.5f;
          float nn = go_sqrtf(1 - m * m);
          nn *= SQRT3;
          float tx = ms1_clamp((m + m) * z - kx, 0, 1);
          float ty = ms1_clamp((-nn - m) * z - kx, 0, 1);
          V2 qx = add2(d, scale2(tx, add2(c, scale2(tx, b))));
          V2 qy = add2(d, scale2(ty, add2(c, scale2(ty, b))));
          float dx = dot2(qx, qx), dy = dot2(qy, qy);
          res = dx < dy ? dx : dy;
        }
        dist[i] = go_sqrtf(res) - thick;
      }
      return 0;
    }
    case GSDF_CIRCLE2D: { /* :661-667 */
      for (size_t i = 0; i < n; i++) dist[i] = norm2(pos[i]) - P[0];
      return 0;
    }
    case GSDF_EQTRI2D: { /* :669-683 */
      const float k = SQRT3;
      float r = P[0] / SQRT3;
      for (size_t i = 0; i < n; i++) {
        V2 p = pos[i];
        p.x = go_absf(p.x) - r;
        p.y += r / k;
        if (p.x + k * p.y > 0) {
          V2 t = {p.x - k * p.y, -k * p.x - p.y};
          p = scale2(0.5f, t);
        }
        p.x -= orc_clampf(p.x, -2 * r, 0);
        dist[i] = -norm2(p) * orc_signf(p.y);
      }
      return 0;
    }
    case GSDF_RECT2D: { /* :685-692 */
      V2 b = {0.5f * P[0], 0.5f * P[1]};
      for (size_t i = 0; i < n; i++) {
        V2 d = {go_absf(pos[i].x) - b.x, go_absf(pos[i].y) - b.y};
        V2 dm = {go_maxf(d.x, 0), go_maxf(d.y, 0)};
        dist[i] = norm2(dm) + go_minf(0, go_maxf(d.x, d.y));
      }
      return 0;
    }
    case GSDF_DIAMOND2D: { /* :694-703 */
      V2 b = {0.5f * P[0], 0.5f * P[1]};
      for (size_t i = 0; i < n; i++) {
        V2 p = {go_absf(pos[i].x), go_absf(pos[i].y)};
        V2 t = sub2(b, scale2(2, p));
        float h = ms1_clamp((t.x * b.x - t.y * b.y) / dot2(b, b), -1, 1);
        V2 hb = scale2(0.5f, b);
        V2 m = {hb.x * (1 - h), hb.y * (1 + h)};
        float d = norm2(sub2(p, m));
        dist[i] = d * ms1_sign(p.x * b.y + p.y * b.x - b.x * b.y);
      }
      return 0;
    }
    case GSDF_X2D: { /* :705-716 */
      float w = P[0], r = P[1];
      for (size_t i = 0; i < n; i++) {
        V2 p = {go_absf(pos[i].x), go_absf(pos[i].y)};
        float sub = 0.5f * go_minf(p.x + p.y, w);
        p.x -= sub; p.y -= sub;
        dist[i] = norm2(p) - r;
      }
      return 0;
    }
    case GSDF_HEX2D: { /* :718-729 */
      float r = P[0];
      V2 k = {-TRIBISECT, 0.5f};
      const float kz = 0.577350269f;
      for (size_t i = 0; i < n; i++) {
        V2 p = {go_absf(pos[i].x), go_absf(pos[i].y)};
        p = sub2(p, scale2(2 * go_minf(dot2(k, p), 0), k));
        V2 c = {orc_clampf(p.x, -kz * r, kz * r), r};
        p = sub2(p, c);
        dist[i] = orc_signf(p.y) * norm2(p);
      }
      return 0;
    }
    case GSDF_OCT2D: { /* :731-748 */
      const float kx = -0.9238795325f, ky = 0.3826834323f, kz = 0.4142135623f;
      float r = P[0];
      float kzr = kz * r, nkzr = -kzr;
      V2 v1 = {kx, ky}, v2 = {-kx, ky};
      for (size_t i = 0; i < n; i++) {
        V2 p = {go_absf(pos[i].x), go_absf(pos[i].y)};
        p = sub2(p, scale2(2 * go_minf(dot2(v1, p), 0), v1));
        p = sub2(p, scale2(2 * go_minf(dot2(v2, p), 0), v2));
        V2 c = {ms1_clamp(p.x, nkzr, kzr), r};
        p = sub2(p, c);
        dist[i] = orc_signf(p.y) * norm2(p);
      }
      return 0;
    }
    case GSDF_ELLIPSE2D: { /* :750-791 */
      for (size_t i = 0; i < n; i++) {
        float a = P[0], b = P[1];
        V2 p = {go_absf(pos[i].x), go_absf(pos[i].y)};
        if (p.x > p.y) { float t = p.x; p.x = p.y; p.y = t; t = a; a = b; b = t; }
        float l = b * b - a * a;
        float m = a * p.x / l;
        float m2 = m * m;
        float nn = b * p.y / l;
        float n2 = nn * nn;
        float c = (m2 + n2 - 1) / 3;
        float c3 = c * c * c;
        float q = c3 + 2 * m2 * n2;
        float d = c3 + m2 * n2;
        float g = m + m * n2;
        float co;
        if (d < 0) {
          float h = go_acosf(q / c3) / 3;
          float sh, ch;
          go_sincosf(h, &sh, &ch);
          float t = SQRT3 * sh;
          float rx = go_sqrtf(-c * (ch + t + 2) + m2);
          float ry = go_sqrtf(-c * (ch - t + 2) + m2);
          co = (ry + orc_signf(l) * rx + go_absf(g) / (rx * ry) - m) / 2;
        } else {
          float h = 2 * m * nn * go_sqrtf(d);
          float sv = orc_signf(q + h) * go_cbrtf(go_absf(q + h));
          float u = orc_signf(q - h) * go_cbrtf(go_absf(q - h));
          float rx = -sv - u - 4 * c + 2 * m2;
          float ry = SQRT3 * (sv - u);
          float rm = go_hypotf(rx, ry);
          co = (ry / go_sqrtf(rm - rx) + 2 * g / rm - m) / 2;
        }
        V2 r = {a * co, b * go_sqrtf(1 - co * co)};
        dist[i] = norm2(sub2(r, p)) * orc_signf(p.y - r.y);
      }
      return 0;
    }
    case GSDF_POLY2D: { /* :793-818 */
      uint32_t nv = nd->aux_len / 2;
      if (nv < 3) return -2;
      const V2* verts = (const V2*)&s->aux[nd->aux_off];
      for (size_t i = 0; i < n; i++) {
        V2 p = pos[i];
        float d = norm2sq(sub2(p, verts[0]));
        float sg = 1.0f;
        uint32_t jv = nv - 1;
        for (uint32_t iv = 0; iv < nv; iv++) {
          V2 v1 = verts[iv], v2 = verts[jv];
          V2 e = sub2(v2, v1);
          V2 w = sub2(p, v1);
          V2 b = sub2(w, scale2(ms1_clamp(dot2(w, e) / norm2sq(e), 0, 1), e));
          d = go_minf(d, norm2sq(b));
          int b1 = p.y >= v1.y, b2 = p.y < v2.y, b3 = e.x * w.y > e.y * w.x;
          if ((b1 && b2 && b3) || (!b1 && !b2 && !b3)) sg = -sg;
          jv = iv;
        }
        dist[i] = sg * go_sqrtf(d);
      }
      return 0;
    }
    case GSDF_LINES2D: { /* :1145-1160 */
      uint32_t ns = nd->aux_len / 4;
      const float* seg = &s->aux[nd->aux_off];
      float w = P[0] / 2;
      for (size_t i = 0; i < n; i++) {
        V2 p = pos[i];
        float d = 1e23f;
        for (uint32_t k = 0; k < ns; k++) {
          V2 a = {seg[4 * k], seg[4 * k + 1]}, b = {seg[4 * k + 2], seg[4 * k + 3]};
          V2 pa = sub2(p, a), ba = sub2(b, a);
          float dotba = dot2(ba, ba);
          float h = ms1_clamp(dot2(pa, ba) / dotba, 0, 1);
          d = go_minf(d, norm2sq(sub2(pa, scale2(h, ba))));
        }
        dist[i] = go_sqrtf(d) - w;
      }
      return 0;
    }
    case GSDF_UNION2D: { /* :821-845 */
      if (nd->nchild < 2) return -2;
      float* aux = ACQ_F(n);
      err = eval2(s, CHILD(nd, 0), pos, dist, n, vp);
      for (uint32_t c = 1; c < nd->nchild && !err; c++) {
        err = eval2(s, CHILD(nd, c), pos, aux, n, vp);
        if (!err) for (size_t i = 0; i < n; i++) dist[i] = go_minf(dist[i], aux[i]);
      }
      REL_F(aux);
      return err;
    }
    case GSDF_INTERSECT2D: case GSDF_DIFF2D: case GSDF_XOR2D: { /* :847-912 */
      if (nd->nchild != 2) return -2;
      float* d2 = ACQ_F(n);
      err = eval2(s, CHILD(nd, 0), pos, dist, n, vp);
      if (!err) err = eval2(s, CHILD(nd, 1), pos, d2, n, vp);
      if (!err) {
        if (vp->lip && nd->op != GSDF_INTERSECT2D) binop_combine_lip(nd->op, 0, dist, d2, n);
        else binop_combine(nd->op, 0, dist, d2, n);
      }
      REL_F(d2);
      return err;
    }
    case GSDF_ARRAY2D: { /* :914-962 */
      V2* t = ACQ_V2(n);
      float* aux = ACQ_F(n);
      V2 sp = {P[0], P[1]};
      V2 nn = {P[2] + -1, P[3] + -1};
      for (size_t i = 0; i < n; i++) dist[i] = LARGENUM;
      for (int j = 0; j < 2 && !err; j++)
        for (int ii = 0; ii < 2 && !err; ii++) {
          V2 ij = {(float)ii, (float)j};
          for (size_t ip = 0; ip < n; ip++) {
            V2 p = pos[ip];
            V2 id = {go_roundf(p.x / sp.x), go_roundf(p.y / sp.y)};
            V2 o = {ms1_sign(p.x - sp.x * id.x), ms1_sign(p.y - sp.y * id.y)};
            V2 rid = {id.x + ij.x * o.x, id.y + ij.y * o.y};
            rid.x = ms1_clamp(rid.x, 0, nn.x); rid.y = ms1_clamp(rid.y, 0, nn.y);
            t[ip].x = p.x - sp.x * rid.x; t[ip].y = p.y - sp.y * rid.y;
          }
          err = eval2(s, CHILD(nd, 0), t, aux, n, vp);
          if (!err) for (size_t i = 0; i < n; i++) dist[i] = go_minf(dist[i], aux[i]);
        }
      REL_F(aux);
      REL_V2(t);
      return err;
    }
    case GSDF_OFFSET2D: { /* :964-978 */
      err = eval2(s, CHILD(nd, 0), pos, dist, n, vp);
      if (!err) for (size_t i = 0; i < n; i++) dist[i] = dist[i] + P[0];
      return err;
    }
    case GSDF_TRANSLATE2D: { /* :980-996 */
      V2* t = ACQ_V2(n);
      for (size_t i = 0; i < n; i++) { t[i].x = pos[i].x - P[0]; t[i].y = pos[i].y - P[1]; }
      err = eval2(s, CHILD(nd, 0), t, dist, n, vp);
      REL_V2(t);
      return err;
    }
    case GSDF_SYMMETRY2D: { /* :998-1024 */
      V2* t = ACQ_V2(n);
      memcpy(t, pos, n * sizeof(V2));
      int bits = (int)P[0];
      for (size_t i = 0; i < n; i++) {
        if (bits & 1) t[i].x = go_absf(t[i].x);
        if (bits & 2) t[i].y = go_absf(t[i].y);
      }
      err = eval2(s, CHILD(nd, 0), t, dist, n, vp);
      REL_V2(t);
      return err;
    }
    case GSDF_ANNULUS2D: { /* :1026-1040 */
      err = eval2(s, CHILD(nd, 0), pos, dist, n, vp);
      if (!err && vp->lip) {
        for (size_t j = 0; j < n / 2; j++) {
          float alo, ahi;
          lip_abs(dist[2 * j], dist[2 * j + 1], &alo, &ahi);
          dist[2 * j] = alo - P[0];
          dist[2 * j + 1] = ahi - P[0];
        }
      } else
      if (!err) for (size_t i = 0; i < n; i++) dist[i] = go_absf(dist[i]) - P[0];
      return err;
    }
    case GSDF_CIRCARRAY2D: { /* :1094-1143 */
      V2* pos0 = ACQ_V2(n);
      V2* pos1 = ACQ_V2(n);
      float angle = (float)(2 * ORC_PI) / P[1];
      float ncirc = P[1];
      float ninsm1 = (float)((int)P[0] - 1);
      for (size_t i = 0; i < n; i++) {
        V2 p = pos[i];
        float pangle = go_atan2f(p.y, p.x);
        float id = go_floorf(pangle / angle);
        if (id < 0) id += ncirc;
        float i0, i1;
        if (id >= ninsm1) { i0 = ninsm1; i1 = 0; } else { i0 = id; i1 = id + 1; }
        float s0, c0, s1, c1;
        go_sincosf(angle * i0, &s0, &c0);
        go_sincosf(angle * i1, &s1, &c1);
        pos0[i].x = c0 * p.x + s0 * p.y; pos0[i].y = (-s0) * p.x + c0 * p.y;
        pos1[i].x = c1 * p.x + s1 * p.y; pos1[i].y = (-s1) * p.x + c1 * p.y;
      }
      float* dist1 = ACQ_F(n);
      err = eval2(s, CHILD(nd, 0), pos1, dist1, n, vp);
      if (!err) err = eval2(s, CHILD(nd, 0), pos0, dist, n, vp);
      if (!err) for (size_t i = 0; i < n; i++) dist[i] = go_minf(dist[i], dist1[i]);
      REL_F(dist1);
      REL_V2(pos1);
      REL_V2(pos0);
      return err;
    }
    case GSDF_TRANSLATEMULTI2D: { /* :1162-1184 */
      uint32_t nd2 = nd->aux_len / 2;
      const float* disp = &s->aux[nd->aux_off];
      for (size_t i = 0; i < n; i++) dist[i] = 3.40282346638528859811704183484516925440e+38f;
      float* d1 = ACQ_F(n);
      for (uint32_t k = 0; k < nd2 && !err; k++) {
        V2* t = ACQ_V2(n); /* translate2D.Evaluate :980-996 */
        for (size_t i = 0; i < n; i++) { t[i].x = pos[i].x - disp[2 * k]; t[i].y = pos[i].y - disp[2 * k + 1]; }
        err = eval2(s, CHILD(nd, 0), t, d1, n, vp);
        REL_V2(t);
        if (!err) min_reduce(dist, d1, n);
      }
      REL_F(d1);
      return err;
    }
    case GSDF_ROTATION2D: { /* :1186-1203 ; MulMatVec(tInv,p) */
      V2* t = ACQ_V2(n);
      for (size_t i = 0; i < n; i++) {
        V2 p = pos[i];
        t[i].x = P[0] * p.x + P[1] * p.y;
        t[i].y = P[2] * p.x + P[3] * p.y;
      }
      float* vsave = NULL;
      if (vp->lip) {
        float f = lip_norm2(P[0], P[1], P[2], P[3]);
        if (f > LIP_RIGID_TOL) {
          f = f * LIP_ROUND_UP;
          vsave = lip_enter(vp, n);
          for (size_t j = 0; j < n / 2; j++) lip_stretch(vp->lip, j, f);
        }
      }
      err = eval2(s, CHILD(nd, 0), t, dist, n, vp);
      if (vsave) lip_leave(vp, vsave, n);
      REL_V2(t);
      return err;
    }
    case GSDF_SCALE2D: { /* :1205-1226 */
      V2* t = ACQ_V2(n);
      float inv = 1.f / P[0];
      for (size_t i = 0; i < n; i++) { t[i].x = inv * pos[i].x; t[i].y = inv * pos[i].y; }
      float* lsave = vp->lip ? lip_enter(vp, n) : NULL;
      if (vp->lip) lip_scale(vp->lip, n, inv);
      err = eval2(s, CHILD(nd, 0), t, dist, n, vp);
      if (vp->lip) {
        lip_leave(vp, lsave, n);
        if (!err) lip_mul(dist, n, P[0]);
      } else
      if (!err) for (size_t i = 0; i < n; i++) dist[i] = dist[i] * P[0];
      REL_V2(t);
      return err;
    }
    case GSDF_ELONGATE2D: { /* :1228-1255 */
      V2* t = ACQ_V2(n);
      float* aux = ACQ_F(n);
      V2 h = {0.5f * P[0], 0.5f * P[1]};
      for (size_t i = 0; i < n; i++) {
        V2 q = {go_absf(pos[i].x) - h.x, go_absf(pos[i].y) - h.y};
        aux[i] = go_minf(go_maxf(q.x, q.y), 0);
        t[i].x = go_maxf(q.x, 0); t[i].y = go_maxf(q.y, 0);
      }
      if (vp->lip) lip_widen(vp->lip, aux, n);
      err = eval2(s, CHILD(nd, 0), t, dist, n, vp);
      if (!err) for (size_t i = 0; i < n; i++) dist[i] += aux[i];
      REL_F(aux);
      REL_V2(t);
      return err;
    }
    default:
      return -3;
  }
}

/* gleval/cpu.go:92-118 SDF3CPU.Evaluate */
int orc_eval3(const orc_sdf* s, orc_pool* vp, const float* pos, float* dist, size_t n) {
  if (n == 0) return -1; /* errEmptyBuffers */
  int err = eval3(s, s->root, (const V3*)pos, dist, n, vp);
  if (err) return err;
  if (!pool_all_released(vp)) return -4;
  return 0;
}
/* Bounds lo[i] <= field <= hi[i] over the ball of radius h around pos[i] (the cube whose centre pos[i] is and whose half
 * diagonal h is): interval evaluation of the tree, see lipctx. Not a reference function; what the octree's centre tests
 * need in order to stay surface-preserving for fields that are not 1-Lipschitz. For a tree of
 * exact-distance primitives under rigid motions and min / max it returns d -+ h exactly: the reference's predicate.
 * Sector and cell seams of (circular) arrays are taken as continuous, as those nodes' own Bounds() assume. */
int orc_eval3_bounds(const orc_sdf* s, orc_pool* vp, const float* pos, float* lo, float* hi, size_t n, float h) {
  if (n == 0) return -1;
  lipctx lc;
  lc.R = (float*)malloc(sizeof(float) * n);
  float* p2 = (float*)malloc(sizeof(float) * 6 * n);
  float* d2 = (float*)malloc(sizeof(float) * 2 * n);
  for (size_t i = 0; i < n; i++) {
    lc.R[i] = h;
    for (int k = 0; k < 3; k++) p2[6 * i + k] = p2[6 * i + 3 + k] = pos[3 * i + k];
  }
  vp->lip = &lc;
  int err = eval3(s, s->root, (const V3*)p2, d2, 2 * n, vp);
  vp->lip = NULL;
  for (size_t i = 0; i < n && !err; i++) { lo[i] = d2[2 * i]; hi[i] = d2[2 * i + 1]; }
  free(lc.R); free(p2); free(d2);
  if (err) return err;
  if (!pool_all_released(vp)) return -4;
  return 0;
}
int orc_eval2(const orc_sdf* s, orc_pool* vp, const float* pos, float* dist, size_t n) {
  if (n == 0) return -1;
  int err = eval2(s, s->root, (const V2*)pos, dist, n, vp);
  if (err) return err;
  if (!pool_all_released(vp)) return -4;
  return 0;
}
