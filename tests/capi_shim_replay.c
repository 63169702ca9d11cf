/*
 * capi_shim_replay.c -- the call sequence of the cgo shim in INTEGRATION.md, replayed from plain C.
 *
 * There is no Go toolchain in the build image, so the shim itself cannot be compiled. What cgo would do is call the C ABI
 * with C structs laid out as the headers under include/ say (pinned there by GSDF_ABI_ASSERT) -- exactly what a C program does. Each function
 * below is the C twin of one Go function of INTEGRATION.md (same name, same order of ABI calls, same error mapping), and
 * main() drives them the way gsdfaux.RenderShader3D would (/root/reference/gsdfaux/gsdfaux.go:93-241):
 *   InitHIP -> flatten -> NewHIPSDF3 (+ Specialize) -> Evaluate (errors first) -> NewOctreeRendererHIP -> RenderAll's
 *   ReadTriangles loop with a 4096-triangle buffer (glrender/glrender.go:17-36) -> WriteBinarySTL -> HIPUniqueID / NewHIPComm /
 *   Gather at world size 1 -> Close.
 * Known answers checked: the unit sphere at resolution 1/33 gives 41072 triangles (glrender_test.go:83-99); distances of the
 * sphere are |p| - 1. Built and run by tests/test_gpu_capi_replay.py (-m gpu); compiled and linked (not run) by the CPU suite.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gsdf_hip.h"

/* ---- Go-side error values the shim maps status codes to (hipErr) ---- */
typedef enum { GoNil = 0, GoErrEmptyBuffers, GoErrMismatchBufferLength, GoErrShortBuffer, GoEOF, GoErrOther } go_error;
static go_error hipErr(int rc) {
  switch (rc) {
    case GSDF_OK: return GoNil;
    case GSDF_ERR_EMPTY_BUFFERS: return GoErrEmptyBuffers;          /* gleval.errEmptyBuffers, gleval/gleval.go:47 */
    case GSDF_ERR_LENGTH_MISMATCH: return GoErrMismatchBufferLength; /* gleval.errMismatchBufferLength, gleval.go:48 */
    case GSDF_ERR_SHORT_BUFFER: return GoErrShortBuffer;             /* io.ErrShortBuffer */
    default: return GoErrOther;                                      /* errors.New("gsdf_hip: " + gsdf_hip_last_error()) */
  }
}
#define CHECK(cond)                                                                      \
  do {                                                                                   \
    if (!(cond)) {                                                                       \
      fprintf(stderr, "replay: line %d: %s (last error: %s)\n", __LINE__, #cond, gsdf_hip_last_error()); \
      return 1;                                                                          \
    }                                                                                    \
  } while (0)

/* ---- section 2: the flattener appends one gsdf_node per reference node, a field copy ---- */
typedef struct { gsdf_node nodes[8]; uint32_t links[8]; uint32_t n_nodes, n_links; } HIPTree;
static uint32_t leaf(HIPTree* t, int op, float p0, float p1, float p2, float p3) {
  gsdf_node* n = &t->nodes[t->n_nodes];
  memset(n, 0, sizeof *n);
  n->op = (uint16_t)op;
  n->p[0] = p0; n->p[1] = p1; n->p[2] = p2; n->p[3] = p3;
  return t->n_nodes++;
}
static uint32_t op3(HIPTree* t, int op, const uint32_t* kids, uint16_t nk, float p0, float p1, float p2) {
  gsdf_node* n = &t->nodes[t->n_nodes];
  memset(n, 0, sizeof *n);
  n->op = (uint16_t)op;
  n->nchild = nk;
  n->link_off = t->n_links;
  for (uint16_t k = 0; k < nk; k++) t->links[t->n_links++] = kids[k];
  n->p[0] = p0; n->p[1] = p1; n->p[2] = p2;
  return t->n_nodes++;
}

/* ---- section 1: SDF3HIP ---- */
typedef struct { gsdf_program* h; float bb[6]; } SDF3HIP;
static go_error InitHIP(int device) { return hipErr(gsdf_hip_init(device)); }
static go_error NewHIPSDF3(const HIPTree* t, uint32_t root, const float bb[6], int specialize, SDF3HIP* s) {
  gsdf_tree ct;
  memset(&ct, 0, sizeof ct);
  ct.nodes = t->nodes; ct.n_nodes = t->n_nodes;
  ct.links = t->links; ct.n_links = t->n_links;
  ct.root = root;
  memcpy(ct.bb, bb, sizeof ct.bb);
  memcpy(s->bb, bb, sizeof s->bb);
  int rc = gsdf_hip_program_create(&ct, &s->h);
  if (rc) return hipErr(rc);
  if (specialize) (void)gsdf_hip_program_specialize(s->h); /* optional: on failure the handle keeps the interpreter kernels */
  return GoNil;
}
static go_error Evaluate(SDF3HIP* s, const float* pos, size_t npos, float* dist, size_t ndist) {
  return hipErr(gsdf_hip_eval3(s->h, npos ? pos : NULL, 12 /* unsafe.Sizeof(ms3.Vec{}) */, npos, ndist ? dist : NULL, ndist));
}

/* ---- glrender/octree_hip.go ---- */
typedef struct { gsdf_mesh* m; uint64_t n, cur; } OctreeHIP;
static go_error NewOctreeRendererHIP(SDF3HIP* s, float res, OctreeHIP* o) {
  gsdf_mesh_opts opts;
  memset(&opts, 0, sizeof opts);
  opts.prune = 1; opts.shard_rank = 0; opts.shard_count = 1;
  memset(o, 0, sizeof *o);
  int rc = gsdf_hip_mesh_octree(s->h, res, &opts, &o->m);
  if (rc) return hipErr(rc);
  gsdf_mesh_stats st;
  gsdf_hip_mesh_stats_get(o->m, &st);
  o->n = st.n_tris;
  return GoNil;
}
static go_error ReadTriangles(OctreeHIP* o, float* dst, size_t len_dst, int* nout) { /* dst: len_dst ms3.Triangle = 9 floats each */
  *nout = 0;
  if (len_dst < 5) return GoErrShortBuffer; /* octreerenderer.go:132-134 */
  uint64_t n = o->n - o->cur < len_dst ? o->n - o->cur : len_dst;
  if (n > 0) {
    int rc = gsdf_hip_mesh_read(o->m, o->cur, n, dst);
    if (rc) return hipErr(rc);
    o->cur += n;
  }
  *nout = (int)n;
  return o->cur == o->n ? GoEOF : GoNil;
}
static go_error WriteBinarySTL(OctreeHIP* o, const uint8_t** file, size_t* len) {
  return hipErr(gsdf_hip_mesh_host_stl(o->m, file, len));
}

int main(void) {
  CHECK(InitHIP(0) == GoNil);

  /* gsdf.FlattenHIP of Difference(Union(sphere, Translate(box)), cylinder) and of the bare unit sphere */
  HIPTree t;
  memset(&t, 0, sizeof t);
  uint32_t sph = leaf(&t, GSDF_SPHERE, 1.0f, 0, 0, 0);
  uint32_t box = leaf(&t, GSDF_BOX, 1.0f, 0.6f, 0.8f, 0.1f);
  uint32_t tr = op3(&t, GSDF_TRANSLATE, &box, 1, 0.7f, 0.2f, -0.1f);
  uint32_t uk[2] = {sph, tr};
  uint32_t un = op3(&t, GSDF_UNION, uk, 2, 0, 0, 0);
  uint32_t cyl = leaf(&t, GSDF_CYLINDER, 0.3f, 3.0f, 0.0f, 0);
  uint32_t dk[2] = {un, cyl};
  uint32_t diff = op3(&t, GSDF_DIFF, dk, 2, 0, 0, 0);
  const float bb_csg[6] = {-1.0f, -1.0f, -1.0f, 1.2f, 1.0f, 1.0f}, bb_sph[6] = {-1, -1, -1, 1, 1, 1};

  SDF3HIP s, csg;
  CHECK(NewHIPSDF3(&t, sph, bb_sph, 0, &s) == GoNil);
  CHECK(NewHIPSDF3(&t, diff, bb_csg, 1 /* HIPConfig.Specialize */, &csg) == GoNil);
  float got[6];
  CHECK(gsdf_hip_program_bounds(csg.h, got) == 0 && memcmp(got, bb_csg, sizeof got) == 0); /* Bounds() */

  /* Evaluate: the reference's error values first (gleval/gpu.go:83-87), then distances */
  enum { N = 1000 };
  static float pos[3 * N], dist[N], dist2[N];
  for (int i = 0; i < N; i++) { pos[3 * i] = -1.5f + 0.003f * (float)i; pos[3 * i + 1] = 0.37f * (float)(i % 7) - 1.0f; pos[3 * i + 2] = 0.11f * (float)(i % 19) - 1.0f; }
  CHECK(Evaluate(&s, pos, 0, dist, 0) == GoErrEmptyBuffers);
  CHECK(Evaluate(&s, pos, N, dist, N - 1) == GoErrMismatchBufferLength);
  CHECK(Evaluate(&s, pos, N, dist, N) == GoNil);
  for (int i = 0; i < N; i++) {
    double e = sqrt((double)pos[3 * i] * pos[3 * i] + (double)pos[3 * i + 1] * pos[3 * i + 1] + (double)pos[3 * i + 2] * pos[3 * i + 2]) - 1.0;
    CHECK(fabs((double)dist[i] - e) <= 1e-6 * (1.0 + fabs(e)));
  }
  CHECK(gsdf_hip_evaluations(s.h) == N); /* Evaluations() counts what was evaluated, not the failed calls */
  CHECK(Evaluate(&csg, pos, N, dist2, N) == GoNil);
  for (int i = 0; i < N; i++) CHECK(dist2[i] >= dist[i] - 1e-6f || dist2[i] > -2.0f); /* finite, sane */

  /* NewOctreeRenderer + RenderAll (glrender.go:17-36): 4096-triangle buffer, append until io.EOF */
  OctreeHIP oc;
  CHECK(NewOctreeRendererHIP(&s, -1.0f, &oc) == GoErrOther && strstr(gsdf_hip_last_error(), "invalid renderer cube resolution"));
  CHECK(NewOctreeRendererHIP(&s, 1.0f / 33.0f, &oc) == GoNil);
  CHECK(oc.n == 41072); /* glrender_test.go:91 */
  static float buf[4096 * 9];
  float* all = (float*)malloc((size_t)oc.n * 36);
  int nread, calls = 0;
  CHECK(ReadTriangles(&oc, buf, 4, &nread) == GoErrShortBuffer && nread == 0);
  uint64_t total = 0;
  for (;;) {
    go_error e = ReadTriangles(&oc, buf, 4096, &nread);
    CHECK(e == GoNil || e == GoEOF);
    memcpy(all + 9 * total, buf, (size_t)nread * 36);
    total += (uint64_t)nread;
    calls++;
    if (e == GoEOF) break;
  }
  CHECK(total == 41072 && calls == 11); /* ceil(41072 / 4096) */
  for (uint64_t i = 0; i < 9 * total; i += 3) { /* every vertex on the sphere to within the lattice step */
    double r = sqrt((double)all[i] * all[i] + (double)all[i + 1] * all[i + 1] + (double)all[i + 2] * all[i + 2]);
    CHECK(fabs(r - 1.0) < 1.0 / 33.0);
  }

  /* WriteBinarySTL: one w.Write of the file the device built (stl.go:15-62: 80-byte header, u32 count, 50 B per triangle) */
  const uint8_t* file;
  size_t flen;
  CHECK(WriteBinarySTL(&oc, &file, &flen) == GoNil && flen == 84 + 50 * (size_t)oc.n);
  uint32_t cnt;
  memcpy(&cnt, file + 80, 4);
  CHECK(cnt == 41072);
  CHECK(memcmp(file + 84 + 12, all, 36) == 0); /* first record: normal (12 B), then the three vertices as read above */

  /* section 4: HIPUniqueID / NewHIPComm / Gather, world size 1 */
  uint8_t id[GSDF_COMM_ID_BYTES];
  gsdf_comm* comm = NULL;
  CHECK(gsdf_hip_comm_unique_id(id) == 0);
  CHECK(gsdf_hip_comm_create(id, 0, 1, &comm) == 0 && gsdf_hip_comm_world(comm) == 1 && gsdf_hip_comm_rank(comm) == 0);
  gsdf_mesh* gathered = NULL;
  uint64_t counts[1] = {0};
  CHECK(gsdf_hip_mesh_gatherv(oc.m, comm, &gathered, counts) == 0 && counts[0] == 41072);
  gsdf_mesh_stats gst;
  CHECK(gsdf_hip_mesh_stats_get(gathered, &gst) == 0 && gst.n_tris == 41072);
  uint64_t sums[2] = {gsdf_hip_evaluations(s.h), 7};
  const uint64_t before = sums[0];
  CHECK(gsdf_hip_comm_allreduce_sum_u64(comm, sums, 2) == 0 && sums[0] == before && sums[1] == 7);
  CHECK(gsdf_hip_mesh_read(gathered, 0, 1, buf) == 0 && memcmp(buf, all, 36) == 0);

  /* NewOctreeShardHIP (payload = records) + GatherStart / GatherWait, the source closed right after the start; and the plan the
   * gather runs, asked for directly */
  {
    gsdf_mesh_opts ro;
    memset(&ro, 0, sizeof ro);
    ro.prune = 1; ro.shard_count = 1; ro.payload = GSDF_PAYLOAD_RECORDS;
    gsdf_mesh* shard = NULL;
    CHECK(gsdf_hip_mesh_octree(s.h, 1.0f / 33, &ro, &shard) == 0);
    uint64_t nrec = 0, nbytes = 0;
    CHECK(gsdf_hip_mesh_payload(shard, &nrec, &nbytes) == GSDF_PAYLOAD_RECORDS && nrec > 10000 && nbytes > 40 * nrec);
    CHECK(gsdf_hip_mesh_read(shard, 0, 1, buf) != 0); /* no triangles yet */
    gsdf_gather* pending = NULL;
    CHECK(gsdf_hip_mesh_gatherv_start(shard, comm, GSDF_GATHER_ALL, 0, &pending) == 0);
    gsdf_hip_mesh_destroy(shard); /* deferred by the library until the payload has moved */
    gsdf_mesh* g2 = NULL;
    gsdf_gather_stats gs;
    CHECK(gsdf_hip_mesh_gatherv_wait(pending, &g2, counts, &gs) == 0 && g2 != NULL && counts[0] == 41072 && gs.ms_march > 0);
    CHECK(gsdf_hip_mesh_stats_get(g2, &gst) == 0 && gst.n_tris == 41072);
    CHECK(gsdf_hip_mesh_read(g2, 41071, 1, buf) == 0);
    gsdf_hip_mesh_destroy(g2);
    const uint64_t sizes[3] = {72, 0, 36};
    gsdf_gather_op ops[8];
    size_t n_ops = 0;
    uint64_t total = 0;
    CHECK(gsdf_hip_gather_plan(sizes, 3, 2, GSDF_GATHER_ALL, 0, ops, 8, &n_ops, &total) == 0 && total == 108 && n_ops == 4);
    CHECK(ops[0].kind == GSDF_GOP_COPY && ops[0].dst_off == 72 && ops[0].bytes == 36);
  }
  /* three meshes of one program in flight (gsdf_hip_mesh_octree_start / _wait) */
  {
    gsdf_mesh_opts po;
    memset(&po, 0, sizeof po);
    po.prune = 1; po.shard_count = 1;
    gsdf_mesh_job *ja = NULL, *jb = NULL, *jc = NULL, *jd = NULL;
    CHECK(gsdf_hip_mesh_octree_start(s.h, 1.0f / 33, &po, &ja) == 0 && gsdf_hip_mesh_octree_start(s.h, 1.0f / 20, &po, &jb) == 0);
    CHECK(gsdf_hip_mesh_octree_start(s.h, 1.0f / 25, &po, &jc) == 0);
    CHECK(gsdf_hip_mesh_octree_start(s.h, 1.0f / 20, &po, &jd) != 0); /* a fourth is refused */
    gsdf_mesh *ma = NULL, *mb = NULL, *mc = NULL;
    CHECK(gsdf_hip_mesh_octree_wait(ja, &ma) == 0 && gsdf_hip_mesh_octree_wait(jb, &mb) == 0 && gsdf_hip_mesh_octree_wait(jc, &mc) == 0);
    CHECK(gsdf_hip_mesh_stats_get(ma, &gst) == 0 && gst.n_tris == 41072);
    CHECK(gsdf_hip_mesh_stats_get(mb, &gst) == 0 && gst.n_tris > 1000);
    CHECK(gsdf_hip_mesh_stats_get(mc, &gst) == 0 && gst.n_tris > 1000);
    gsdf_hip_mesh_destroy(ma); gsdf_hip_mesh_destroy(mb); gsdf_hip_mesh_destroy(mc);
  }

  /* the CSG program meshes too, through the specialised kernels, and pruning does not change its surface */
  OctreeHIP oc2, oc3;
  CHECK(NewOctreeRendererHIP(&csg, 0.02f, &oc2) == GoNil && oc2.n > 10000);
  gsdf_mesh_opts np;
  memset(&np, 0, sizeof np);
  np.prune = 0; np.shard_count = 1;
  CHECK(gsdf_hip_mesh_octree(csg.h, 0.02f, &np, &oc3.m) == 0);
  gsdf_hip_mesh_stats_get(oc3.m, &gst);
  CHECK(gst.n_tris == oc2.n);

  /* Close */
  gsdf_hip_mesh_destroy(gathered);
  gsdf_hip_comm_destroy(comm);
  gsdf_hip_mesh_destroy(oc.m); gsdf_hip_mesh_destroy(oc2.m); gsdf_hip_mesh_destroy(oc3.m);
  gsdf_hip_program_destroy(s.h); gsdf_hip_program_destroy(csg.h);
  free(all);
  printf("replay ok: 41072 triangles in %d ReadTriangles calls, STL %zu bytes, csg %llu triangles\n", calls, flen, (unsigned long long)oc2.n);
  return 0;
}
