/*
 * orc_render.c -- ORACLE (test infrastructure only). CPU restatement of the reference meshers:
 *   /root/reference/glrender/marchcubes.go:8-98        marchCubes, mcToTriangles, mcInterpolate
 *   /root/reference/glrender/flatrenderer.go:36-256    FlatRenderer (Reset, evalGrid, evalKRange, ReadTriangles)
 *   /root/reference/glrender/octreerenderer.go:71-284  Octree (Reset, makeICube, prune predicate, leaf corners)
 *   /root/reference/glrender/stl.go:15-62              WriteBinarySTL
 *   /root/reference/gleval/gleval.go:53-108            NormalsCentralDiff
 * ms3.Octree / i3.Cube / ms3.Box are external (soypat/geometry, not vendored): restated from the
 * call sites -- PARITY UNPINNED for those; pinned at count level by the reference's
 * known answers (41072 sphere triangles, glrender_test.go:91; 423,852 npt-flange@400, README.md:116,130).
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "mc_tables.h"
#include "orc_eval.h"
#include "orc_math.h"

typedef struct { float x, y, z; } V3;

static inline float ms1_clamp_r(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

const uint16_t* orc_mc_edge_table(void) { return ORC_MC_EDGE; }
const int8_t* orc_mc_tri_table(void) { return &ORC_MC_TRI[0][0]; }

void orc_mesh_free(orc_mesh* m) {
  if (m && m->tris) { free(m->tris); m->tris = NULL; }
}
static void mesh_reserve(orc_mesh* m, uint64_t extra) {
  if (m->n_tris + extra <= m->cap) return;
  uint64_t nc = m->cap ? m->cap * 2 : 4096;
  while (nc < m->n_tris + extra) nc *= 2;
  m->tris = (float*)realloc(m->tris, nc * 9 * sizeof(float));
  m->cap = nc;
}

/* ---------------- marching cubes (marchcubes.go) ---------------- */
static const float GLRENDER_SQRT3 = 1.73205080757f; /* glrender.go:9 */

/* marchcubes.go:76-98 */
static V3 mc_interpolate(V3 p1, V3 p2, float v1, float v2, float x) {
  const float eps = 1e-12f;
  int c1 = go_absf(x - v1) < eps;
  int c2 = go_absf(x - v2) < eps;
  if (c1 && !c2) return p1;
  if (c2 && !c1) return p2;
  float t = 0.5f;
  if (!c1 || !c2) t = (x - v1) / (v2 - v1);
  V3 r = {p1.x + t * (p2.x - p1.x), p1.y + t * (p2.y - p1.y), p1.z + t * (p2.z - p1.z)};
  return r;
}

/* marchcubes.go:34-73 ; dst must have room for 5 triangles (45 floats). */
static int mc_to_triangles(float* dst, const V3 p[8], const float v[8], float x) {
  int index = 0;
  for (int i = 0; i < 8; i++)
    if (v[i] < x) index |= 1 << i;
  int edges = ORC_MC_EDGE[index];
  if (edges == 0) return 0;
  V3 pts[12];
  for (int i = 0; i < 12; i++) {
    if (edges & (1 << i)) {
      int a = ORC_MC_PAIR[i][0], b = ORC_MC_PAIR[i][1];
      pts[i] = mc_interpolate(p[a], p[b], v[a], v[b], x);
    }
  }
  const int8_t* table = ORC_MC_TRI[index];
  int nt = 0;
  for (int i = 0; i < 16 && table[i] >= 0; i += 3) {
    V3 a = pts[table[i + 2]], b = pts[table[i + 1]], c = pts[table[i + 0]];
    float* t = dst + 9 * nt;
    t[0] = a.x; t[1] = a.y; t[2] = a.z;
    t[3] = b.x; t[4] = b.y; t[5] = b.z;
    t[6] = c.x; t[7] = c.y; t[8] = c.z;
    nt++;
  }
  return nt;
}

/* marchcubes.go:14-32 over explicit cubes. */
uint64_t orc_march_cubes(const float* pos, const float* dist, uint64_t ncubes, float res, float* tris) {
  float cubeDiag = 2 * GLRENDER_SQRT3 * res;
  uint64_t nt = 0;
  for (uint64_t c = 0; c < ncubes; c++) {
    const float* d = dist + 8 * c;
    if (go_absf(d[0]) <= cubeDiag) nt += mc_to_triangles(tris + 9 * nt, (const V3*)(pos + 24 * c), d, 0);
  }
  return nt;
}

/* ---------------- ms3.Box helpers [external, restated] ---------------- */
typedef struct { V3 min, max; } Box3;
static Box3 box_scale_centered(Box3 a, float sx, float sy, float sz) {
  /* ScaleCentered(scale) = NewCenteredBox(a.Center(), MulElem(scale, a.Size())) */
  V3 c = {0.5f * (a.min.x + a.max.x), 0.5f * (a.min.y + a.max.y), 0.5f * (a.min.z + a.max.z)};
  V3 sz3 = {a.max.x - a.min.x, a.max.y - a.min.y, a.max.z - a.min.z};
  V3 s = {go_maxf(sx, 0) * sz3.x, go_maxf(sy, 0) * sz3.y, go_maxf(sz, 0) * sz3.z};
  V3 h = {0.5f * go_maxf(s.x, 0), 0.5f * go_maxf(s.y, 0), 0.5f * go_maxf(s.z, 0)};
  Box3 r = {{c.x - h.x, c.y - h.y, c.z - h.z}, {c.x + h.x, c.y + h.y, c.z + h.z}};
  return r;
}

/* ---------------- FlatRenderer (flatrenderer.go) ---------------- */
typedef struct {
  const orc_sdf* s;
  float res;
  V3 origin;
  int nx, ny, nz;
  float* grid;
  int batch;
  int k0, k1;
  uint64_t evals;
  int err;
} flat_job;

/* flatrenderer.go:146-182 evalKRange */
static void* flat_eval_krange(void* arg) {
  flat_job* j = (flat_job*)arg;
  int bufSize = j->batch;
  float* posbuf = (float*)malloc(sizeof(float) * 3 * bufSize);
  float* distbuf = (float*)malloc(sizeof(float) * bufSize);
  orc_pool* vp = orc_pool_create((size_t)bufSize);
  size_t sz = (size_t)(j->nx + 1) * (j->ny + 1);
  size_t batchStart = (size_t)j->k0 * sz;
  int posIdx = 0;
  for (int k = j->k0; k < j->k1 && !j->err; k++)
    for (int jj = 0; jj <= j->ny && !j->err; jj++)
      for (int i = 0; i <= j->nx; i++) {
        posbuf[3 * posIdx + 0] = j->origin.x + (float)i * j->res;
        posbuf[3 * posIdx + 1] = j->origin.y + (float)jj * j->res;
        posbuf[3 * posIdx + 2] = j->origin.z + (float)k * j->res;
        posIdx++;
        if (posIdx == bufSize) {
          j->err = orc_eval3(j->s, vp, posbuf, distbuf, (size_t)bufSize);
          if (j->err) break;
          memcpy(j->grid + batchStart, distbuf, sizeof(float) * bufSize);
          j->evals += (uint64_t)bufSize;
          batchStart += (size_t)bufSize;
          posIdx = 0;
        }
      }
  if (posIdx > 0 && !j->err) {
    j->err = orc_eval3(j->s, vp, posbuf, distbuf, (size_t)posIdx);
    if (!j->err) {
      memcpy(j->grid + batchStart, distbuf, sizeof(float) * posIdx);
      j->evals += (uint64_t)posIdx;
    }
  }
  orc_pool_destroy(vp);
  free(posbuf);
  free(distbuf);
  return NULL;
}

int orc_render_flat(const orc_sdf* s, float res, int batch, int nthreads, orc_mesh* out) {
  memset(out, 0, sizeof(*out));
  if (!(res > 0) || batch < 8 || nthreads < 1) return -1; /* flatrenderer.go:37-45 */
  float bbf[6];
  orc_sdf_bounds(s, bbf);
  Box3 bb = {{bbf[0], bbf[1], bbf[2]}, {bbf[3], bbf[4], bbf[5]}};
  bb = box_scale_centered(bb, 1.01f, 1.01f, 1.01f); /* :47-48 */
  V3 sz = {bb.max.x - bb.min.x, bb.max.y - bb.min.y, bb.max.z - bb.min.z};
  int nx = (int)go_ceilf(sz.x / res), ny = (int)go_ceilf(sz.y / res), nz = (int)go_ceilf(sz.z / res);
  if (nx <= 0 || ny <= 0 || nz <= 0) return -2;
  size_t gridSize = (size_t)(nx + 1) * (ny + 1) * (nz + 1);
  float* grid = (float*)malloc(sizeof(float) * gridSize);
  if (!grid) return -5;
  out->nx = nx; out->ny = ny; out->nz = nz;

  /* evalGrid :103-141 */
  double t0 = now_s();
  int numG = nthreads;
  if (numG > nz + 1) numG = nz + 1;
  flat_job* jobs = (flat_job*)calloc((size_t)numG, sizeof(flat_job));
  pthread_t* th = (pthread_t*)calloc((size_t)numG, sizeof(pthread_t));
  for (int g = 0; g < numG; g++) {
    flat_job* j = &jobs[g];
    j->s = s; j->res = res; j->origin = bb.min; j->nx = nx; j->ny = ny; j->nz = nz;
    j->grid = grid; j->batch = batch;
    j->k0 = (int)((long long)g * (nz + 1) / numG);
    j->k1 = (int)((long long)(g + 1) * (nz + 1) / numG);
    if (numG == 1) flat_eval_krange(j);
    else pthread_create(&th[g], NULL, flat_eval_krange, j);
  }
  int err = 0;
  for (int g = 0; g < numG; g++) {
    if (numG > 1) pthread_join(th[g], NULL);
    if (jobs[g].err) err = jobs[g].err;
    out->evals += jobs[g].evals;
  }
  free(jobs);
  free(th);
  out->t_eval_s = now_s() - t0;
  if (err) { free(grid); return err; }

  /* ReadTriangles :186-256 */
  t0 = now_s();
  size_t sy = (size_t)nx + 1;
  size_t szz = sy * ((size_t)ny + 1);
  float cubeDiag = 2 * GLRENDER_SQRT3 * res;
  V3 o = bb.min;
  for (int cz = 0; cz < nz; cz++)
    for (int cy = 0; cy < ny; cy++)
      for (int cx = 0; cx < nx; cx++) {
        size_t base = (size_t)cx + (size_t)cy * sy + (size_t)cz * szz;
        if (go_absf(grid[base]) > cubeDiag) continue;
        float v[8] = {grid[base], grid[base + 1], grid[base + 1 + sy], grid[base + sy],
                      grid[base + szz], grid[base + 1 + szz], grid[base + 1 + sy + szz], grid[base + sy + szz]};
        float ox = o.x + (float)cx * res, oy = o.y + (float)cy * res, oz = o.z + (float)cz * res;
        float r = res;
        V3 p[8] = {{ox, oy, oz},         {ox + r, oy, oz},         {ox + r, oy + r, oz},         {ox, oy + r, oz},
                   {ox, oy, oz + r},     {ox + r, oy, oz + r},     {ox + r, oy + r, oz + r},     {ox, oy + r, oz + r}};
        mesh_reserve(out, 5);
        out->n_tris += (uint64_t)mc_to_triangles(out->tris + 9 * out->n_tris, p, v, 0);
      }
  out->t_march_s = now_s() - t0;
  free(grid);
  return 0;
}

/* ---------------- Octree (octreerenderer.go) ---------------- */
typedef struct { int32_t x, y, z; int32_t level; } Cube; /* x,y,z in leaf units; size = 2^(level-1) leaves */

/* makeICube :222-235 */
static int make_icube(Box3 bb, float res, int* levels) {
  if (!(res > 0) || res != res || isinf(res)) return -1;
  V3 sz = {bb.max.x - bb.min.x, bb.max.y - bb.min.y, bb.max.z - bb.min.z};
  float longAxis = go_maxf(sz.x, go_maxf(sz.y, sz.z));
  float l2 = go_log2f(longAxis / res);
  int lv = (int)go_ceilf(l2) + 1;
  if (lv <= 1) return -2;
  *levels = lv;
  return 0;
}
/* i3.Cube/ms3.Octree CubeSize, CubeOrigin [external]: size = float(2^(level-1))*res ;
 * origin = Origin + size*float(levelIndex) with levelIndex = leafcoord >> (level-1). */
static inline float cube_size(int level, float res) { return (float)(1 << (level - 1)) * res; }
static inline V3 cube_origin(Cube c, V3 origin, float size) {
  int sh = c.level - 1;
  V3 r = {origin.x + size * (float)(c.x >> sh), origin.y + size * (float)(c.y >> sh), origin.z + size * (float)(c.z >> sh)};
  return r;
}

typedef struct { Cube* v; size_t n, cap; } CubeVec;
static void cv_push(CubeVec* cv, Cube c) {
  if (cv->n == cv->cap) { cv->cap = cv->cap ? cv->cap * 2 : 1024; cv->v = (Cube*)realloc(cv->v, cv->cap * sizeof(Cube)); }
  cv->v[cv->n++] = c;
}

/* A surviving Level-3 cube as the device's share_corners options see it (gsdf_amd/csrc/kernels_octree.h: DZ, leaf_dense_kernel):
 * per axis the eight corner coordinates of its four leaves, A_k = O + res*float(i0+k) and A_k + res, of which (A_{k-1} + res) and
 * A_k are the same plane -- and one point to an evaluator iff they are the same float (bits: -0 and +0 differ). */
static void count_brick(orc_mesh* out, Cube c, V3 origin, float res) {
  const float org[3] = {origin.x, origin.y, origin.z};
  const int32_t i0[3] = {c.x, c.y, c.z};
  unsigned n[3];
  for (int ax = 0; ax < 3; ax++) {
    n[ax] = 5;
    for (int k = 1; k < 4; k++) {
      const float far = (org[ax] + res * (float)(i0[ax] + k - 1)) + res, near = org[ax] + res * (float)(i0[ax] + k);
      uint32_t a, b;
      memcpy(&a, &far, 4); memcpy(&b, &near, 4);
      if (a != b) n[ax]++;
    }
  }
  const unsigned N = n[0] * n[1] * n[2];
  out->evals_rows += 64u * n[2];
  out->evals_points += N;
  out->evals_points_256 += (N + 255u) & ~255u;
  unsigned t0 = 0, slots = 0;
  for (; t0 + 192u < N; t0 += 256u) slots += 256u;
  if (t0 < N) { const unsigned rem = N - t0; slots += rem <= 64u ? 64u : (rem <= 128u ? 128u : 256u); }
  out->evals_points_tails += slots;
}

int orc_render_octree(const orc_sdf* s, float res, int batch, int prune, orc_mesh* out) {
  memset(out, 0, sizeof(*out));
  if (batch < 64) return -1; /* :46-48 */
  if (!(res > 0)) return -1;
  float bbf[6];
  orc_sdf_bounds(s, bbf);
  Box3 bb = {{bbf[0], bbf[1], bbf[2]}, {bbf[3], bbf[4], bbf[5]}};
  bb = box_scale_centered(bb, 1.01f, 1.01f, 1.01f); /* :79-80 */
  int levels;
  int err = make_icube(bb, res, &levels);
  if (err) return err;
  if (levels > 21) return -6;
  out->levels = levels;
  V3 origin = bb.min;
  batch &= ~7;
  float* posbuf = (float*)malloc(sizeof(float) * 3 * (size_t)batch);
  float* distbuf = (float*)malloc(sizeof(float) * (size_t)batch);
  float* hibuf = (float*)malloc(sizeof(float) * (size_t)batch);
  orc_pool* vp = orc_pool_create((size_t)batch);
  const float szMult = GLRENDER_SQRT3 / 2; /* :182 */
  const int assume_sdf = (prune & ORC_PRUNE_ASSUME_SDF) != 0;
  int pmask = prune & ~ORC_PRUNE_ASSUME_SDF;
  if (assume_sdf && pmask == 0) pmask = 1;

  /* Level-synchronous descent. Every cube with Level >= minPrunableLvl(3) is centre-tested with the
   * reference predicate (:270-273) -- a superset of the capacity-dependent subset the reference
   * tests through DecomposeBFS (:140, table :94-105); identical surface for 1-Lipschitz fields. */
  CubeVec cur = {0}, nxt = {0};
  Cube top = {0, 0, 0, levels};
  cv_push(&cur, top);
  double t0 = now_s();
  for (int level = levels; level >= 2; level--) {
    nxt.n = 0;
    if (level >= 3 && (pmask == 1 || (pmask > 1 && ((pmask >> level) & 1)))) { /* prune: 0 none, 1 every level >= 3, else a bit mask of the levels to test */
      float size = cube_size(level, res);
      float maxDist = size * szMult;
      for (size_t b0 = 0; b0 < cur.n; b0 += (size_t)batch) {
        size_t nb = cur.n - b0 < (size_t)batch ? cur.n - b0 : (size_t)batch;
        for (size_t i = 0; i < nb; i++) {
          V3 o = cube_origin(cur.v[b0 + i], origin, size);
          V3 mx = {o.x + size, o.y + size, o.z + size};
          /* CubeCenter = box.Center() = Scale(0.5, Add(Min, Max)) [external] */
          posbuf[3 * i] = 0.5f * (o.x + mx.x); posbuf[3 * i + 1] = 0.5f * (o.y + mx.y); posbuf[3 * i + 2] = 0.5f * (o.z + mx.z);
        }
        if (assume_sdf) err = orc_eval3(s, vp, posbuf, distbuf, nb);
        else err = orc_eval3_bounds(s, vp, posbuf, distbuf, hibuf, nb, maxDist);
        if (err) goto done;
        out->evals += nb;
        for (size_t i = 0; i < nb; i++) {
          /* :270-273 |d| >= maxDist, i.e. d - maxDist >= 0 or d + maxDist <= 0: with the field's bounds over the cube
           * in place of d -+ maxDist (the same numbers when the field is a true distance) */
          int prunable = assume_sdf ? go_absf(distbuf[i]) >= maxDist : (distbuf[i] >= 0.0f || hibuf[i] <= 0.0f);
          if (!prunable) {
            Cube c = cur.v[b0 + i];
            int h = 1 << (level - 2); /* child size in leaves */
            /* children in corner order (i3.Cube octree decomposition [external]) */
            const int ox[8] = {0, 1, 1, 0, 0, 1, 1, 0}, oy[8] = {0, 0, 1, 1, 0, 0, 1, 1}, oz[8] = {0, 0, 0, 0, 1, 1, 1, 1};
            for (int k = 0; k < 8; k++) { Cube ch = {c.x + ox[k] * h, c.y + oy[k] * h, c.z + oz[k] * h, level - 1}; cv_push(&nxt, ch); }
            if (level == 3) count_brick(out, c, origin, res);
          } else {
            out->pruned += (uint64_t)1 << (3 * (level - 1)); /* DecomposesTo(1) = 8^(level-1) :279 */
          }
        }
      }
    } else {
      for (size_t i = 0; i < cur.n; i++) {
        Cube c = cur.v[i];
        int h = 1 << (level - 2);
        const int ox[8] = {0, 1, 1, 0, 0, 1, 1, 0}, oy[8] = {0, 0, 1, 1, 0, 0, 1, 1}, oz[8] = {0, 0, 0, 0, 1, 1, 1, 1};
        for (int k = 0; k < 8; k++) { Cube ch = {c.x + ox[k] * h, c.y + oy[k] * h, c.z + oz[k] * h, level - 1}; cv_push(&nxt, ch); }
        if (level == 3) count_brick(out, c, origin, res);
      }
    }
    CubeVec t = cur; cur = nxt; nxt = t;
  }
  /* cur = leaf cubes (level 1): 8 corners each (Box.Vertices order), evaluate + marchCubes :161-172 */
  {
    size_t cubesPerBatch = (size_t)batch / 8;
    for (size_t b0 = 0; b0 < cur.n; b0 += cubesPerBatch) {
      size_t nb = cur.n - b0 < cubesPerBatch ? cur.n - b0 : cubesPerBatch;
      for (size_t i = 0; i < nb; i++) {
        V3 o = cube_origin(cur.v[b0 + i], origin, res);
        V3 m = {o.x + res, o.y + res, o.z + res};
        float* p = posbuf + 24 * i;
        p[0] = o.x; p[1] = o.y; p[2] = o.z;
        p[3] = m.x; p[4] = o.y; p[5] = o.z;
        p[6] = m.x; p[7] = m.y; p[8] = o.z;
        p[9] = o.x; p[10] = m.y; p[11] = o.z;
        p[12] = o.x; p[13] = o.y; p[14] = m.z;
        p[15] = m.x; p[16] = o.y; p[17] = m.z;
        p[18] = m.x; p[19] = m.y; p[20] = m.z;
        p[21] = o.x; p[22] = m.y; p[23] = m.z;
      }
      err = orc_eval3(s, vp, posbuf, distbuf, nb * 8);
      if (err) goto done;
      out->evals += nb * 8;
      mesh_reserve(out, 5 * nb);
      out->n_tris += orc_march_cubes(posbuf, distbuf, nb, res, out->tris + 9 * out->n_tris);
    }
  }
  out->t_eval_s = now_s() - t0;
done:
  free(cur.v); free(nxt.v);
  orc_pool_destroy(vp);
  free(posbuf); free(distbuf); free(hibuf);
  return err;
}

/* ---------------- STL (stl.go:15-62) ---------------- */
size_t orc_stl_size(uint64_t n) { return 84 + 50 * (size_t)n; }
static void put_f32(uint8_t* b, float f) { uint32_t u = orc_f32bits(f); b[0] = u; b[1] = u >> 8; b[2] = u >> 16; b[3] = u >> 24; }
int orc_write_stl(const float* tris, uint64_t n, uint8_t* dst) {
  if (n == 0) return -1;           /* "empty triangle slice" */
  if (n > 0xffffffffull) return -2; /* exceeds STL design limits */
  memset(dst, 0, 84);
  uint32_t cnt = (uint32_t)n;
  dst[80] = cnt; dst[81] = cnt >> 8; dst[82] = cnt >> 16; dst[83] = cnt >> 24;
  for (uint64_t i = 0; i < n; i++) {
    const float* t = tris + 9 * i;
    uint8_t* b = dst + 84 + 50 * i;
    /* ms3.Triangle.Normal() = Cross(t1-t0, t2-t0); Unit(v) = Scale(1/Norm(v), v) [external] */
    V3 a = {t[3] - t[0], t[4] - t[1], t[5] - t[2]};
    V3 c = {t[6] - t[0], t[7] - t[1], t[8] - t[2]};
    V3 nrm = {a.y * c.z - a.z * c.y, a.z * c.x - a.x * c.z, a.x * c.y - a.y * c.x};
    float inv = 1 / go_hypotf(nrm.x, go_hypotf(nrm.y, nrm.z));
    put_f32(b, inv * nrm.x); put_f32(b + 4, inv * nrm.y); put_f32(b + 8, inv * nrm.z);
    for (int k = 0; k < 9; k++) put_f32(b + 12 + 4 * k, t[k]);
    b[48] = 0; b[49] = 0;
  }
  return 0;
}

/* ---------------- NormalsCentralDiff (gleval/gleval.go:53-108) ---------------- */
int orc_normals_central_diff(const orc_sdf* s, orc_pool* vp, const float* pos, float* normals, size_t n, float step) {
  step *= 0.5f;
  if (!(step > 0)) return -1;
  if (n == 0) return -1;
  float* d1 = (float*)malloc(sizeof(float) * n);
  float* d2 = (float*)malloc(sizeof(float) * n);
  float* aux = (float*)malloc(sizeof(float) * 3 * n);
  int err = 0;
  for (int dim = 0; dim < 3 && !err; dim++) {
    for (size_t i = 0; i < n; i++) { aux[3 * i] = pos[3 * i]; aux[3 * i + 1] = pos[3 * i + 1]; aux[3 * i + 2] = pos[3 * i + 2]; aux[3 * i + dim] = pos[3 * i + dim] + step; }
    err = orc_eval3(s, vp, aux, d1, n);
    if (err) break;
    for (size_t i = 0; i < n; i++) { aux[3 * i] = pos[3 * i]; aux[3 * i + 1] = pos[3 * i + 1]; aux[3 * i + 2] = pos[3 * i + 2]; aux[3 * i + dim] = pos[3 * i + dim] - step; }
    err = orc_eval3(s, vp, aux, d2, n);
    if (err) break;
    for (size_t i = 0; i < n; i++) normals[3 * i + dim] = d1[i] - d2[i];
  }
  free(d1); free(d2); free(aux);
  return err;
}

/* ---------------- DualContourRenderer (dual_contour.go, dual_contour_vertexplacement.go) ----------------
 * Reset :26-83 (bounds shifted by -res/2, full decomposition to level 1, prune by cube ORIGIN with
 * |d| < 2*size), RenderAll :85-219 (4 evaluations per cube, sign-bit edge activity, neighbour
 * accumulation, quads -> 2 triangles), DualContourLeastSquares.PlaceVertices (vertexplacement.go:26-143),
 * vertMean :145-150, leastSquaresMGS64 :152-223.
 * cubebuf order is ms3.Octree.DecomposeBFS order [external]; restated as lexicographic (z, y, x), which
 * fixes the (floating point) row order of the least-squares systems: PARITY UNPINNED at the value
 * level for this renderer; the reference's own tests for it are tolerance tests (dual_contour_test.go). */
typedef struct {
  int32_t x, y, z;
  float d0, dx, dy, dz;
  V3 fv;
  int nnb;
  int32_t nb_cube[12];
  int8_t nb_axis[12];
} DualCube;

static inline int dc_signbit(float f) { return (int)(orc_f32bits(f) >> 31); }
static inline float dc_isect(float o, float e) { return -o / (e - o); }

static void lsq_mgs64(const float (*A)[3], const float* b, int K, float x_out[3]) {
  x_out[0] = x_out[1] = x_out[2] = 0;
  if (K < 3) return;
  double Q[20][3], b64[20], R[3][3] = {{0}};
  for (int k = 0; k < K; k++) { Q[k][0] = A[k][0]; Q[k][1] = A[k][1]; Q[k][2] = A[k][2]; b64[k] = b[k]; }
  for (int j = 0; j < 3; j++) {
    for (int i = 0; i < j; i++) {
      double dot = 0;
      for (int k = 0; k < K; k++) dot += Q[k][i] * Q[k][j];
      R[i][j] = dot;
      for (int k = 0; k < K; k++) Q[k][j] -= dot * Q[k][i];
    }
    double nsq = 0;
    for (int k = 0; k < K; k++) nsq += Q[k][j] * Q[k][j];
    double norm = sqrt(nsq);
    R[j][j] = norm;
    if (norm > 1e-14) {
      double inv = 1.0 / norm;
      for (int k = 0; k < K; k++) Q[k][j] *= inv;
    }
  }
  double Qtb[3] = {0, 0, 0};
  for (int j = 0; j < 3; j++)
    for (int k = 0; k < K; k++) Qtb[j] += Q[k][j] * b64[k];
  double x[3];
  for (int i = 2; i >= 0; i--) {
    x[i] = Qtb[i];
    for (int k = i + 1; k < 3; k++) x[i] -= R[i][k] * x[k];
    if (R[i][i] > 1e-14) x[i] /= R[i][i];
    else x[i] = 0;
  }
  x_out[0] = (float)x[0]; x_out[1] = (float)x[1]; x_out[2] = (float)x[2];
}

int orc_render_dualcontour(const orc_sdf* s, float res, int chiseled, orc_mesh* out) {
  memset(out, 0, sizeof(*out));
  if (!(res > 0)) return -1;
  float bbf[6];
  orc_sdf_bounds(s, bbf);
  const float sub = res / 2;
  Box3 bb = {{bbf[0] + -sub, bbf[1] + -sub, bbf[2] + -sub}, {bbf[3] + -sub, bbf[4] + -sub, bbf[5] + -sub}};
  int levels;
  int err = make_icube(bb, res, &levels);
  if (err) return err;
  if (levels > 10) return -6; /* 8^(levels-1) cubes: the reference itself is infeasible beyond this */
  out->levels = levels;
  const V3 origin = bb.min;
  const int n = 1 << (levels - 1);
  const size_t N = (size_t)n * n * n;
  orc_pool* vp = orc_pool_create(4096);
  const size_t B = 1 << 16;
  float* posbuf = (float*)malloc(sizeof(float) * 3 * B * 4);
  float* distbuf = (float*)malloc(sizeof(float) * B * 4);
  int32_t* grid = (int32_t*)malloc(sizeof(int32_t) * N);
  DualCube* cubes = NULL;
  size_t ncubes = 0, capc = 0;
  /* Reset: prune by origin, keep iff |d| < 2*res (octreePrunea szMult=2, useOrigin=true) */
  for (size_t c0 = 0; c0 < N && !err; c0 += B) {
    size_t nb = N - c0 < B ? N - c0 : B;
    for (size_t i = 0; i < nb; i++) {
      size_t c = c0 + i;
      int x = (int)(c % n), y = (int)((c / n) % n), z = (int)(c / ((size_t)n * n));
      posbuf[3 * i] = origin.x + res * (float)x; posbuf[3 * i + 1] = origin.y + res * (float)y; posbuf[3 * i + 2] = origin.z + res * (float)z;
    }
    err = orc_eval3(s, vp, posbuf, distbuf, nb);
    out->evals += nb;
    for (size_t i = 0; i < nb && !err; i++) {
      size_t c = c0 + i;
      float maxDist = res * 2;
      if (go_absf(distbuf[i]) >= maxDist) { grid[c] = -1; continue; }
      if (ncubes == capc) { capc = capc ? capc * 2 : 4096; cubes = (DualCube*)realloc(cubes, capc * sizeof(DualCube)); }
      DualCube* q = &cubes[ncubes];
      memset(q, 0, sizeof(*q));
      q->x = (int)(c % n); q->y = (int)((c / n) % n); q->z = (int)(c / ((size_t)n * n));
      grid[c] = (int32_t)ncubes++;
    }
  }
  /* RenderAll: 4 evaluations per cube */
  for (size_t c0 = 0; c0 < ncubes && !err; c0 += B) {
    size_t nb = ncubes - c0 < B ? ncubes - c0 : B;
    for (size_t i = 0; i < nb; i++) {
      DualCube* q = &cubes[c0 + i];
      V3 o = {origin.x + res * (float)q->x, origin.y + res * (float)q->y, origin.z + res * (float)q->z};
      float* p = posbuf + 12 * i;
      p[0] = o.x; p[1] = o.y; p[2] = o.z;
      p[3] = o.x + res; p[4] = o.y + 0; p[5] = o.z + 0;
      p[6] = o.x + 0; p[7] = o.y + res; p[8] = o.z + 0;
      p[9] = o.x + 0; p[10] = o.y + 0; p[11] = o.z + res;
    }
    err = orc_eval3(s, vp, posbuf, distbuf, nb * 4);
    out->evals += nb * 4;
    for (size_t i = 0; i < nb && !err; i++) {
      DualCube* q = &cubes[c0 + i];
      q->d0 = distbuf[4 * i]; q->dx = distbuf[4 * i + 1]; q->dy = distbuf[4 * i + 2]; q->dz = distbuf[4 * i + 3];
      q->fv.x = origin.x + res * (float)q->x; q->fv.y = origin.y + res * (float)q->y; q->fv.z = origin.z + res * (float)q->z;
    }
  }
#define GRID(X, Y, Z) (((X) < 0 || (Y) < 0 || (Z) < 0 || (X) >= n || (Y) >= n || (Z) >= n) ? -1 : grid[((size_t)(Z) * n + (Y)) * n + (X)])
  /* neighbour accumulation :111-137 (contributors in cubebuf order, axes x,y,z) */
  static const int NBX[4][3] = {{0, -1, -1}, {0, 0, -1}, {0, 0, 0}, {0, -1, 0}};
  static const int NBY[4][3] = {{-1, 0, -1}, {-1, 0, 0}, {0, 0, 0}, {0, 0, -1}};
  static const int NBZ[4][3] = {{-1, -1, 0}, {0, -1, 0}, {0, 0, 0}, {-1, 0, 0}};
  for (size_t e = 0; e < ncubes && !err; e++) {
    DualCube* q = &cubes[e];
    int ax = dc_signbit(q->d0) != dc_signbit(q->dx), ay = dc_signbit(q->d0) != dc_signbit(q->dy), az = dc_signbit(q->d0) != dc_signbit(q->dz);
    for (int axis = 0; axis < 3; axis++) {
      if (!(axis == 0 ? ax : axis == 1 ? ay : az)) continue;
      const int(*NB)[3] = axis == 0 ? NBX : axis == 1 ? NBY : NBZ;
      for (int k = 0; k < 4; k++) {
        int idx = GRID(q->x + NB[k][0], q->y + NB[k][1], q->z + NB[k][2]);
        if (idx >= 0) {
          DualCube* t = &cubes[idx];
          if (t->nnb < 12) { t->nb_cube[t->nnb] = (int32_t)e; t->nb_axis[t->nnb] = (int8_t)axis; t->nnb++; }
        }
      }
    }
  }
  /* PlaceVertices: normals at the 3 edge intersections of every cube (only active edges are ever read) */
  {
    float step = chiseled ? (float)1e-4 : (float)2e-8;
    float* nrm = (float*)calloc(ncubes * 9 + 9, sizeof(float));
    float* ip = (float*)malloc(sizeof(float) * 9 * B);
    for (size_t c0 = 0; c0 < ncubes && !err; c0 += B) {
      size_t nb = ncubes - c0 < B ? ncubes - c0 : B;
      size_t m = 0;
      size_t* where = (size_t*)malloc(sizeof(size_t) * 3 * nb);
      for (size_t i = 0; i < nb; i++) {
        DualCube* q = &cubes[c0 + i];
        V3 o = {origin.x + res * (float)q->x, origin.y + res * (float)q->y, origin.z + res * (float)q->z};
        float ds[3] = {q->dx, q->dy, q->dz};
        for (int axis = 0; axis < 3; axis++) {
          if (dc_signbit(q->d0) == dc_signbit(ds[axis])) continue;
          float t = res * dc_isect(q->d0, ds[axis]);
          ip[3 * m] = o.x + (axis == 0 ? t : 0); ip[3 * m + 1] = o.y + (axis == 1 ? t : 0); ip[3 * m + 2] = o.z + (axis == 2 ? t : 0);
          where[m++] = (c0 + i) * 3 + (size_t)axis;
        }
      }
      if (m) {
        float* nn = (float*)malloc(sizeof(float) * 3 * m);
        err = orc_normals_central_diff(s, vp, ip, nn, m, step);
        out->evals += 6 * m;
        for (size_t k = 0; k < m && !err; k++) memcpy(nrm + 3 * where[k], nn + 3 * k, 12);
        free(nn);
      }
      free(where);
    }
    free(ip);
    for (size_t e = 0; e < ncubes && !err; e++) {
      DualCube* q = &cubes[e];
      if (q->nnb == 0) continue;
      V3 co = {origin.x + res * (float)q->x, origin.y + res * (float)q->y, origin.z + res * (float)q->z};
      V3 bias[20], ln[20];
      int nr = 0;
      float ds[3] = {q->dx, q->dy, q->dz};
      for (int axis = 0; axis < 3; axis++) { /* the cube's own active edges first (:67-78) */
        if (dc_signbit(q->d0) == dc_signbit(ds[axis])) continue;
        float t = res * dc_isect(q->d0, ds[axis]);
        V3 v = {co.x + (axis == 0 ? t : 0), co.y + (axis == 1 ? t : 0), co.z + (axis == 2 ? t : 0)};
        bias[nr] = v; ln[nr].x = nrm[9 * e + 3 * axis]; ln[nr].y = nrm[9 * e + 3 * axis + 1]; ln[nr].z = nrm[9 * e + 3 * axis + 2];
        nr++;
      }
      for (int k = 0; k < q->nnb; k++) { /* then the neighbours' edges (:81-96); the cube's own edges appear again here */
        DualCube* t = &cubes[q->nb_cube[k]];
        int axis = q->nb_axis[k];
        V3 no = {origin.x + res * (float)t->x, origin.y + res * (float)t->y, origin.z + res * (float)t->z};
        float dsn[3] = {t->dx, t->dy, t->dz};
        float tt = res * dc_isect(t->d0, dsn[axis]);
        V3 v = {no.x + (axis == 0 ? tt : 0), no.y + (axis == 1 ? tt : 0), no.z + (axis == 2 ? tt : 0)};
        size_t ni = (size_t)q->nb_cube[k] * 9 + 3 * (size_t)axis;
        bias[nr] = v; ln[nr].x = nrm[ni]; ln[nr].y = nrm[ni + 1]; ln[nr].z = nrm[ni + 2];
        nr++;
      }
      float invRes = 1.0f / res;
      float A[20][3], b[20];
      V3 mean = {0, 0, 0};
      for (int i = 0; i < nr; i++) {
        V3 qi = {invRes * (bias[i].x - co.x), invRes * (bias[i].y - co.y), invRes * (bias[i].z - co.z)};
        A[i][0] = ln[i].x; A[i][1] = ln[i].y; A[i][2] = ln[i].z;
        b[i] = ln[i].x * qi.x + ln[i].y * qi.y + ln[i].z * qi.z;
        mean.x = mean.x + bias[i].x; mean.y = mean.y + bias[i].y; mean.z = mean.z + bias[i].z;
      }
      float im = 1.f / (float)nr;
      mean.x = im * mean.x; mean.y = im * mean.y; mean.z = im * mean.z;
      V3 bs = {invRes * (mean.x - co.x), invRes * (mean.y - co.y), invRes * (mean.z - co.z)};
      float sl = chiseled ? (float)(sqrt(1e-5) * 1e-4) : (float)sqrt(1e-5);
      A[nr][0] = sl; A[nr][1] = 0; A[nr][2] = 0; b[nr] = sl * bs.x;
      A[nr + 1][0] = 0; A[nr + 1][1] = sl; A[nr + 1][2] = 0; b[nr + 1] = sl * bs.y;
      A[nr + 2][0] = 0; A[nr + 2][1] = 0; A[nr + 2][2] = sl; b[nr + 2] = sl * bs.z;
      float x[3];
      lsq_mgs64((const float(*)[3])A, b, nr + 3, x);
      for (int k = 0; k < 3; k++) x[k] = ms1_clamp_r(x[k], -0.1f, 1.1f);
      q->fv.x = res * x[0] + co.x; q->fv.y = res * x[1] + co.y; q->fv.z = res * x[2] + co.z;
    }
    free(nrm);
  }
  /* quads :151-213 */
  for (size_t e = 0; e < ncubes && !err; e++) {
    DualCube* q = &cubes[e];
    float ds[3] = {q->dx, q->dy, q->dz};
    for (int axis = 0; axis < 3; axis++) {
      if (dc_signbit(q->d0) == dc_signbit(ds[axis])) continue;
      const int(*NB)[3] = axis == 0 ? NBX : axis == 1 ? NBY : NBZ;
      V3 quad[4];
      int all = 1;
      for (int k = 0; k < 4; k++) {
        int idx = GRID(q->x + NB[k][0], q->y + NB[k][1], q->z + NB[k][2]);
        if (idx < 0) { all = 0; break; }
        quad[k] = cubes[idx].fv;
      }
      if (!all) continue;
      if (ds[axis] - q->d0 < 0) { V3 t0 = quad[0], t1 = quad[1]; quad[0] = quad[3]; quad[1] = quad[2]; quad[2] = t1; quad[3] = t0; }
      mesh_reserve(out, 2);
      float* t = out->tris + 9 * out->n_tris;
      const V3 tri[6] = {quad[0], quad[1], quad[2], quad[2], quad[3], quad[0]};
      for (int k = 0; k < 6; k++) { t[3 * k] = tri[k].x; t[3 * k + 1] = tri[k].y; t[3 * k + 2] = tri[k].z; }
      out->n_tris += 2;
    }
  }
#undef GRID
  out->pruned = (uint64_t)(N - ncubes);
  free(cubes); free(grid); free(posbuf); free(distbuf);
  orc_pool_destroy(vp);
  return err;
}

/* test access to leastSquaresMGS64 (rows x 3 floats, b rows floats) */
/* ---------------- minecraftRender (glrender/dual_contour.go:297-403) ----------------
 * Every level-1 cube of the top cube over the SDF's own Bounds(): origin and the ends of its +x, +y, +z edges evaluated (:323-335,
 * ms3.Add adds the zero components too); an edge whose ends differ in sign bit (ActiveX/Y/Z :259-267) contributes the square face
 * across it -- two triangles, first and third vertex exchanged when the far end is the smaller (FlipX/Y/Z :271-273). Cube order:
 * x fastest (the reference's is its breadth-first decomposition's [external]; the mesh is compared as a set). */
static inline V3 v3add(V3 a, float x, float y, float z) { V3 r = {a.x + x, a.y + y, a.z + z}; return r; }
static void mcr_tri(orc_mesh* out, V3 a, V3 b, V3 c, int flip) {
  mesh_reserve(out, 1);
  float* t = out->tris + 9 * out->n_tris++;
  if (flip) { V3 k = a; a = c; c = k; }
  t[0] = a.x; t[1] = a.y; t[2] = a.z; t[3] = b.x; t[4] = b.y; t[5] = b.z; t[6] = c.x; t[7] = c.y; t[8] = c.z;
}
int orc_render_minecraft(const orc_sdf* s, float res, orc_mesh* out) {
  memset(out, 0, sizeof(*out));
  float bbf[6];
  orc_sdf_bounds(s, bbf);
  Box3 bb = {{bbf[0], bbf[1], bbf[2]}, {bbf[3], bbf[4], bbf[5]}};
  int levels;
  int err = make_icube(bb, res, &levels);
  if (err) return err;
  if (levels > 9) return -6;
  out->levels = levels;
  const V3 origin = bb.min;
  const int n = 1 << (levels - 1);
  const size_t N = (size_t)n * n * n;
  orc_pool* vp = orc_pool_create(4096);
  const size_t B = 1 << 14;
  float* posbuf = (float*)malloc(sizeof(float) * 12 * B);
  float* distbuf = (float*)malloc(sizeof(float) * 4 * B);
  const float sz = cube_size(1, res);
  for (size_t c0 = 0; c0 < N && !err; c0 += B) {
    const size_t nb = N - c0 < B ? N - c0 : B;
    for (size_t i = 0; i < nb; i++) {
      const size_t c = c0 + i;
      Cube cb = {(int)(c % n), (int)((c / n) % n), (int)(c / ((size_t)n * n)), 1};
      const V3 o = cube_origin(cb, origin, sz);
      const V3 q[4] = {o, v3add(o, sz, 0.0f, 0.0f), v3add(o, 0.0f, sz, 0.0f), v3add(o, 0.0f, 0.0f, sz)};
      for (int k = 0; k < 4; k++) { posbuf[12 * i + 3 * k] = q[k].x; posbuf[12 * i + 3 * k + 1] = q[k].y; posbuf[12 * i + 3 * k + 2] = q[k].z; }
    }
    err = orc_eval3(s, vp, posbuf, distbuf, 4 * nb);
    out->evals += 4 * nb;
    for (size_t i = 0; i < nb && !err; i++) {
      const size_t c = c0 + i;
      Cube cb = {(int)(c % n), (int)((c / n) % n), (int)(c / ((size_t)n * n)), 1};
      const V3 o = cube_origin(cb, origin, sz);
      const float d0 = distbuf[4 * i], dx = distbuf[4 * i + 1], dy = distbuf[4 * i + 2], dz = distbuf[4 * i + 3];
      if (dc_signbit(d0) != dc_signbit(dx)) {
        const V3 xo = v3add(o, sz, 0.0f, 0.0f);
        const int flip = dx - d0 < 0;
        mcr_tri(out, xo, v3add(xo, 0.0f, sz, 0.0f), v3add(xo, 0.0f, sz, sz), flip);
        mcr_tri(out, v3add(xo, 0.0f, sz, sz), v3add(xo, 0.0f, 0.0f, sz), xo, flip);
      }
      if (dc_signbit(d0) != dc_signbit(dy)) {
        const V3 yo = v3add(o, 0.0f, sz, 0.0f);
        const int flip = dy - d0 < 0;
        mcr_tri(out, yo, v3add(yo, 0.0f, 0.0f, sz), v3add(yo, sz, 0.0f, sz), flip);
        mcr_tri(out, v3add(yo, sz, 0.0f, sz), v3add(yo, sz, 0.0f, 0.0f), yo, flip);
      }
      if (dc_signbit(d0) != dc_signbit(dz)) {
        const V3 zo = v3add(o, 0.0f, 0.0f, sz);
        const int flip = dz - d0 < 0;
        mcr_tri(out, zo, v3add(zo, sz, 0.0f, 0.0f), v3add(zo, sz, sz, 0.0f), flip);
        mcr_tri(out, v3add(zo, sz, sz, 0.0f), v3add(zo, 0.0f, sz, 0.0f), zo, flip);
      }
    }
  }
  free(posbuf); free(distbuf);
  orc_pool_destroy(vp);
  return err;
}

void orc_lsq_mgs64(const float* A, const float* b, int K, float x[3]) { lsq_mgs64((const float(*)[3])A, b, K, x); }
