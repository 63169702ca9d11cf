/*
 * gsdf_hip.h -- C ABI of libgsdfhip.so: the MI355X (gfx950) drop-in for the reference's GPU seam.
 *
 * What each entry point replaces in /root/reference (the cgo binding a maintainer would add is in
 * INTEGRATION.md):
 *   gsdf_hip_init              gleval.Init1x1GLFW                     gleval/gpu.go:21-32
 *   gsdf_hip_program_create    gleval.NewComputeGPUSDF3 / ...SDF2     gleval/gpu.go:35-55,105-130
 *                              (takes the flattened tree instead of GLSL text: glbuild.Programmer
 *                               .WriteComputeSDF3, gsdfaux/gsdfaux.go:122-126)
 *   gsdf_hip_program_bounds    (*SDF3Compute).Bounds                  gleval/gpu.go:75-77
 *   gsdf_hip_evaluations       (*SDF3Compute).Evaluations             gleval/gpu.go:80
 *   gsdf_hip_eval3 / _eval2    (*SDF3Compute).Evaluate / SDF2Compute  gleval/gpu.go:82-103,140-160
 *                              + computeEvaluate                      gleval/gpu_cgo.go:194-258
 *   gsdf_hip_eval3_dev         same, positions/distances already resident in HBM (no reference
 *                              equivalent: the GL path re-uploads every call, gpu_cgo.go:238-257)
 *   gsdf_hip_normals3          gleval.NormalsCentralDiff              gleval/gleval.go:53-108
 *   gsdf_hip_image2            glrender.ImageRendererSDF2.Render (default conversion)  glrender/image.go:46-118
 *   gsdf_hip_mesh_octree       glrender.NewOctreeRenderer + RenderAll glrender/octreerenderer.go:43-178,
 *                              (octree prune + marching cubes on device) glrender/marchcubes.go:14-98
 *   gsdf_hip_mesh_dualcontour  glrender.DualContourRenderer.Reset/RenderAll + DualContourLeastSquares
 *                                                                      glrender/dual_contour.go:26-219, dual_contour_vertexplacement.go:26-223
 *   gsdf_hip_mesh_read         (*Octree).ReadTriangles drain          glrender/octreerenderer.go:131-178
 *   gsdf_hip_mesh_stl          glrender.WriteBinarySTL                glrender/stl.go:15-62
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a negative
 * gsdf_status; gsdf_hip_last_error() gives the message for the calling thread. Buffers are borrowed
 * for the duration of the call only (Go pointer rules: nothing is retained). A handle is not
 * thread-safe (same as the reference's SDF3Compute, gpu.go:94-101); distinct handles are independent.
 */
#ifndef GSDF_HIP_H
#define GSDF_HIP_H
#include <stddef.h>
#include <stdint.h>

#include "gsdf_program.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum gsdf_status {
  GSDF_OK = 0,
  GSDF_ERR_EMPTY_BUFFERS = -1,   /* gleval.errEmptyBuffers (gleval/gleval.go:47) */
  GSDF_ERR_LENGTH_MISMATCH = -2, /* gleval.errMismatchBufferLength (gleval/gleval.go:48) */
  GSDF_ERR_BAD_ARGUMENT = -3,
  GSDF_ERR_BAD_TREE = -4,
  GSDF_ERR_HIP = -5,             /* HIP runtime error, message has hipGetErrorString */
  GSDF_ERR_NO_DEVICE = -6,
  GSDF_ERR_DIMENSION = -7,       /* 2D program given to a 3D entry point or vice versa */
  GSDF_ERR_RESOLUTION = -8,      /* "invalid renderer cube resolution" / "resolution not fine enough" */
  GSDF_ERR_SHORT_BUFFER = -9,    /* io.ErrShortBuffer */
  GSDF_ERR_CAPACITY = -10        /* device triangle buffer capacity exceeded */
} gsdf_status;

typedef struct gsdf_program gsdf_program; /* compiled tree, resident on one GPU */
typedef struct gsdf_mesh gsdf_mesh;       /* triangles produced on device, resident in HBM */

const char* gsdf_hip_last_error(void);

/* Select the HIP device for handles created by this thread afterwards. device < 0: keep current. */
int gsdf_hip_init(int device);

int gsdf_hip_program_create(const gsdf_tree* tree, gsdf_program** out);
void gsdf_hip_program_destroy(gsdf_program* p);
int gsdf_hip_program_bounds(const gsdf_program* p, float bb[6]);
int gsdf_hip_program_is2d(const gsdf_program* p);
/* Introspection for tests/benchmarks: lowered program size (32-bit words) and LDS slots per lane. */
int gsdf_hip_program_info(const gsdf_program* p, uint32_t* code_words, uint32_t* lds_slots);
uint64_t gsdf_hip_evaluations(const gsdf_program* p);
/* Test hook (needs a GPU): exhaustive check of the interpreter's exact division by a wave-uniform divisor d against
 * the IEEE division, over all 2^32 numerators. recip receives RN(1/d) (0: d not eligible, nothing to check). */
int gsdf_hip_selftest_div(float d, uint64_t* mismatches, uint64_t* fast_path_numerators, float* recip);
/* Test hook (needs a GPU): the interpreter's sqrt for hypot's [1,2] argument range against sqrtf, all floats in range; and the
 * per-tree kernels' sqrt without range handling (taken where a wave's arguments are all >= 2^-96) against sqrtf, every such float. */
int gsdf_hip_selftest_sqrt(uint64_t* mismatches);
/* Test hook (needs a GPU): the circular array's sector index from a float32 angle estimate (taken only where it provably
 * decides floor(atan2(y, x) / angle), cpu_evaluators.go:1047-1056) against that expression, over 2^32 points. */
int gsdf_hip_selftest_circ(float ncirc, uint64_t* mismatches, uint64_t* fast_path_points);
/* Test hook (needs a GPU): float32(math.Atan2(y, x)) by the short float64 route of the evaluator's screw / circular-array
 * instructions (forge/threads/threads.go:160, cpu_evaluators.go:1060), which is taken only where it provably rounds like the
 * reference's own operation sequence, against that sequence: 2^log2n pairs; mode 0 hashed pairs of every sign and magnitude,
 * 1 pairs searched towards float32 rounding boundaries, 2 lattice-shaped pairs. mismatches must come back 0. */
int gsdf_hip_selftest_atan2(int mode, int log2n, uint64_t* mismatches, uint64_t* fast_path_points);
/* Test hook (needs a GPU): float32(math.Cos(x)), float32(math.Sin(x)) of a twist's angle (cpu_evaluators.go:1269-1270) by the short
 * float64 route of the per-tree kernels, taken only where it provably rounds like the reference's own operation sequence, against that
 * sequence over EVERY float32 argument. mismatches must come back 0. */
int gsdf_hip_selftest_cossin(uint64_t* mismatches, uint64_t* fast_path_arguments);
/* Run-time specialisation (no reference counterpart; the reference's GPU path compiles GLSL per tree at
 * gleval/gpu.go:35-54, this is the same step for the HIP backend): builds, with hiprtc, eval / prune / leaf kernels in
 * which this program's instructions are laid out straight-line with literal parameters, and makes the handle launch
 * them. Same statements and compiler flags as the interpreter, so results stay bit-identical; costs a few seconds
 * once per handle. Without it (or if hiprtc is unavailable: GSDF_ERR_HIP) the handle runs the interpreter kernels.
 * Environment: GSDF_HIP_CACHE_DIR=<dir> keeps the built code objects on disk (key: generated source, device headers,
 * architecture, options, hiprtc version), so the next process that specialises the same tree reads a file instead. */
int gsdf_hip_program_specialize(gsdf_program* p);
int gsdf_hip_program_is_specialized(const gsdf_program* p, double* compile_seconds);
/* The same build on a thread of its own: returns at once, the handle keeps working through the interpreter kernels and switches to
 * the specialised ones at its first entry-point call after they are ready (bit-identical results either way). For the callers the
 * reference actually has -- one tree, one mesh, one file (examples/npt-flange/flange.go:61-98; gsdfaux.RenderShader3D,
 * gsdfaux/gsdfaux.go:93-171, compiles its shader and then renders once): the first mesh does not wait for a compiler.
 * _poll: 1 = specialised kernels in use, 0 = still building or never started (wait != 0: block until the build has finished),
 * negative status = the build failed (the interpreter kernels stay in use). Destroying the handle waits for a build under way. */
int gsdf_hip_program_specialize_async(gsdf_program* p);
int gsdf_hip_program_specialize_poll(gsdf_program* p, int wait);
/* Names of the kernels this handle launches, as a profiler shows them: "eval=eval_kernel<3,4,4>:specialised
 * leaf=leaf_eval_kernel<4,4>:specialised prune=prune_kernel:specialised compiler=hipcc code=<32 hex digits>" (":interpreter" =
 * the ahead-of-time kernels; compiler = what built the specialised ones: the installed hipcc out of process, or the process's
 * hiprtc; code = key of the code that runs: generated source + device headers + options + compiler identity for specialised
 * kernels, the library's device sources otherwise -- a stored profile describes this handle only if it carries the same key). */
int gsdf_hip_program_kernels(const gsdf_program* p, char* dst, size_t dst_cap);
/* Host-only (run without a GPU): text of the generated evaluator, and a gfx950 hiprtc build of the specialised kernels
 * that stops before loading them. dst may be NULL to query the length. */
int gsdf_hip_specialize_source(const gsdf_tree* tree, char* dst, size_t dst_cap, size_t* len);
int gsdf_hip_specialize_check(const gsdf_tree* tree, size_t* code_object_bytes);
/* Host-only (runs without a GPU): lower a tree to the device instruction stream (gsdf_amd/csrc/dev_ops.h) for
 * inspection. code_out may be NULL to query the size. */
int gsdf_hip_lower(const gsdf_tree* tree, uint32_t* code_out, uint32_t code_cap, uint32_t* code_words, uint32_t* lds_slots);
/* Host-only test hook: the region outside which the lowering claims a positive lower bound of the subtree rooted at
 * `node` (what its skip gates test): *kind 0 = no claim, 1 = box {min xyz, max xyz}, 2 = z-axis cylinder {cx cy r z0 z1
 * rs rin} (rin > 0: an annulus -- the shape also keeps rin away from the axis); tests check the claim against the CPU
 * oracle (tests/test_gate_regions.py). */
int gsdf_hip_lower_region(const gsdf_tree* tree, uint32_t node, int* kind, float params[8]);

/* gleval.BlockCachedSDF3 (gleval/gleval.go:110-218): host-side lossy cache in front of a 3-D program, keyed by the
 * lattice cell int(mul*(p - bb.Min)), mul = 1/res per axis; misses go to the program in one batch. reset = Reset
 * (also clears the statistics), hits = CacheHits, evaluations = Evaluations (cached ones included). */
typedef struct gsdf_blockcache gsdf_blockcache;
int gsdf_hip_blockcache_create(gsdf_program* sdf, float resx, float resy, float resz, gsdf_blockcache** out);
int gsdf_hip_blockcache_reset(gsdf_blockcache* c, gsdf_program* sdf, float resx, float resy, float resz);
int gsdf_hip_blockcache_eval3(gsdf_blockcache* c, const void* pos, size_t pos_stride_bytes, size_t n_pos, float* dist, size_t n_dist);
uint64_t gsdf_hip_blockcache_hits(const gsdf_blockcache* c);
uint64_t gsdf_hip_blockcache_evaluations(const gsdf_blockcache* c);
void gsdf_hip_blockcache_destroy(gsdf_blockcache* c);

/* Host-buffer evaluation (drop-in for SDF3Compute.Evaluate). pos_stride_bytes is the distance between
 * consecutive positions: 12 for []ms3.Vec, 16 for std140 vec3 / ms3.Quat-aligned data; 8 for []ms2.Vec.
 * n_pos/n_dist are the two slice lengths (mismatch and zero are reported like the reference does). */
int gsdf_hip_eval3(gsdf_program* p, const void* pos, size_t pos_stride_bytes, size_t n_pos, float* dist, size_t n_dist);
int gsdf_hip_eval2(gsdf_program* p, const void* pos, size_t pos_stride_bytes, size_t n_pos, float* dist, size_t n_dist);
/* Thread safety of the host-buffer calls: gsdf_hip_eval3 / _eval2 may be called from several host threads on one program at
 * the same time (glrender.FlatRenderer evaluates from numParallel goroutines, flatrenderer.go:120-129): up to 4 calls are
 * in flight on their own streams and staging buffers, further callers wait for a slot. Everything else on a handle is one
 * caller at a time -- and while a background build is pending (gsdf_hip_program_specialize_async started, _poll not yet 1) not at
 * the same time as those concurrent Evaluate calls either: whichever entry point first finds the build finished swaps the
 * handle's kernels in (the Evaluate calls read theirs as one snapshot; a mesh being enqueued on another thread would not).
 * Pipelined form: submit returns at once with a ticket, wait blocks until that call's distances are in `dist` (batches of
 * up to 262144 points, any size in registered memory; at most 4 tickets outstanding per program). */
int gsdf_hip_eval3_submit(gsdf_program* p, const void* pos, size_t pos_stride_bytes, size_t n_pos, float* dist, size_t n_dist, int* ticket);
int gsdf_hip_eval_wait(gsdf_program* p, int ticket);
/* Caller buffers the GPU reaches directly. A host-buffer call whose positions AND distances lie inside memory from
 * gsdf_hip_host_alloc (pinned, device-mapped; C memory, so a Go caller may wrap it in a slice and keep it) or registered
 * with gsdf_hip_host_register (pins and maps memory the caller allocated and keeps alive, e.g. the renderer's long-lived
 * posbuf/distbuf) makes no staging copy: the kernel reads and writes the caller's memory across PCIe.
 * gsdf_hip_host_release frees / unregisters. */
void* gsdf_hip_host_alloc(size_t bytes);
int gsdf_hip_host_register(void* ptr, size_t bytes);
int gsdf_hip_host_release(void* ptr);
/* Device-resident evaluation: d_pos/d_dist are device pointers on the program's GPU; stream is a
 * hipStream_t (NULL = the program's own stream). Asynchronous when stream != NULL. */
int gsdf_hip_eval3_dev(gsdf_program* p, const void* d_pos, size_t pos_stride_bytes, float* d_dist, size_t n, void* stream);
int gsdf_hip_eval2_dev(gsdf_program* p, const void* d_pos, size_t pos_stride_bytes, float* d_dist, size_t n, void* stream);
/* Central-difference normals (not normalised), host buffers, 12-byte xyz in and out. */
int gsdf_hip_normals3(gsdf_program* p, const float* pos, float* normals, size_t n, float step);

/* ImageRendererSDF2.Render of a 2D program over its Bounds(): dist_out (w*h floats, row 0 = top) and/or rgba_out
 * (w*h*4 bytes: black inside, white outside, red for NaN/Inf -- the renderer's default conversion). */
int gsdf_hip_image2(gsdf_program* p, int w, int h, float* dist_out, uint8_t* rgba_out);

/* gsdf_mesh_opts.prune flag: apply the reference's centre test verbatim, |d(centre)| >= size * sqrt3/2
 * (octreerenderer.go:270-273), at the tested levels, as if the field were a true distance field. */
#define GSDF_PRUNE_ASSUME_SDF (1 << 30)
typedef struct gsdf_mesh_opts {
  int prune;          /* which octree levels are centre-tested: 1 = every Level >= 3 (default), 0 = none (visit all leaves), any
                         other value = bit mask (bit L = cubes of Level L, L >= 3). The reference tests the capacity-limited
                         frontier of its DecomposeBFS buffer only (octreerenderer.go:94-105,140): a handful of upper levels.
                         The test drops a cube when the field cannot vanish inside it: the bounds of the field over the cube's
                         bounding ball, obtained by interval evaluation of the tree at the centre, exclude 0. For a true
                         distance field those bounds are d -+ size * sqrt3/2, i.e. the reference's predicate
                         |d| >= size * sqrt3/2 (:270-273); for fields that grow faster than distance (twist, screw, non-rigid
                         transform) or jump (a screw with an asymmetric thread form, across the seams of its sawtooth) they
                         are wider, so that no cube holding surface is dropped at any level: examples/fibonacci-showerhead
                         at resdiv 350 gives the reference's 309,872 triangles (README.md:152,166), where the reference's
                         predicate applied to every Level >= 3 cube gives 309,849. Or'ing GSDF_PRUNE_ASSUME_SDF in selects
                         that predicate (a few percent fewer evaluations; exact only for 1-Lipschitz fields). Sector and cell
                         seams of (circular) arrays are taken as continuous, as those nodes' own Bounds() assume. */
  int shard_rank;     /* multi-GPU: this rank ... */
  int shard_count;    /* ... of this many (1 = whole model). Bricks of 16^3 leaves go to rank gsdf_hip_brick_owner(x, y, z, count). */
  uint64_t max_tris;  /* device triangle buffer capacity; 0 = size automatically */
  void* stream;       /* hipStream_t to run on; NULL = the program's stream */
  int share_corners;  /* 0 (default): every leaf evaluates its own 8 corners like the reference (8 evals/leaf);
                         1: each bitwise-distinct lattice point of a 4x4x4-leaf brick is evaluated once (identical
                         triangles, 1.2-2.2x fewer evaluations; leaf_dense_kernel in the default two-kernel leaf phase -- it
                         gives up the column sharing of the default, so it pays where the field costs more per point than per
                         (x, y) column: threads, knurls, transformed parts);
                         2: the bitwise-distinct z rows of a brick once each (a brick's eight rows of corners are five to eight
                         distinct planes: row 2k-1 = (O + res (i-1)) + res and row 2k = O + res i are the same float on most planes)
                         -- the default's kernels, a quarter fewer evaluations, identical distances, records and triangles.
                         stats.evals then counts the evaluations performed, not the reference's 8 per leaf: for 2, distinct rows
                         x 64 lanes (a second pass rounds the rows it executes up to its width: lane slots that repeat a row
                         are not counted; 1 counts every lane slot of its passes);
                         3: 1 or 2, chosen by how much of the tree's work depends on x and y alone (threads, knurls, transformed
                         parts: 1; mostly axisymmetric parts: 2). */
  int host_output;    /* 1: the triangle buffer is pinned, device-mapped HOST memory and the mesher writes it across PCIe
                         while it runs (gsdf_hip_mesh_host_tris then returns that buffer: mesh + transfer 4.x ms instead
                         of 1.7 + 4.4 ms at npt-flange resdiv 1600). For results that are consumed on the host only:
                         device-side consumers (gsdf_hip_mesh_stl, RCCL gathers) would read back over PCIe. */
  int payload;        /* what the mesh holds when the call returns: GSDF_PAYLOAD_TRIANGLES (0, default), or GSDF_PAYLOAD_RECORDS:
                         the 40-byte records of the leaves the surface cuts (8 corner distances, leaf coordinates, marching-cubes
                         case), packed, and no triangles yet -- the form a rank hands to gsdf_hip_mesh_gatherv_start, which then
                         moves 20 bytes per triangle instead of 36 and runs marching cubes on the receiving ranks, over everybody's
                         records. gsdf_hip_mesh_march turns such a mesh into triangles where it is. Same triangles either way. */
  int reserved;       /* 0 */
} gsdf_mesh_opts;
#define GSDF_PAYLOAD_TRIANGLES 0
#define GSDF_PAYLOAD_RECORDS 1

typedef struct gsdf_mesh_stats {
  uint64_t n_tris;
  uint64_t evals;          /* SDF evaluations performed on device for this mesh */
  uint64_t pruned_leaves;  /* leaf cubes skipped by pruning (Octree.TotalPruned) */
  uint64_t leaf_cubes;     /* leaf cubes visited */
  uint64_t active_leaves;  /* leaf cubes that passed the |d(corner0)| <= 2*sqrt3*res test */
  int levels;              /* octree levels (makeICube) */
  float origin[3];         /* lattice origin (scaled bounds min) */
  float res;
  double ms_total;         /* device time for the whole mesh, HIP events */
  double ms_prune;         /* pruning levels */
  double ms_leaf;          /* leaf phase */
  double ms_march;         /* dominant kernel alone, HIP events: leaf_eval_kernel (the 8 corner evaluations of every leaf);
                              with GSDF_HIP_FUSED_LEAF=1 the fused leaf_kernel (evaluations + marching cubes) */
  uint64_t evals_prune;    /* evaluations done by the pruning levels (cube centres) */
  uint64_t evals_leaf;     /* evaluations done by the leaf phase (leaf corners) */
  double ms_emit;          /* march_records_kernel: marching cubes over the cut-leaf records (0 for the fused kernel) */
  uint64_t cut_leaves;     /* leaves the surface cuts = 40-byte records handed from leaf_eval_kernel to march_records_kernel */
} gsdf_mesh_stats;

GSDF_ABI_ASSERT(sizeof(gsdf_mesh_opts) == 48, "gsdf_mesh_opts is 48 bytes");
GSDF_ABI_ASSERT(offsetof(gsdf_mesh_opts, prune) == 0 && offsetof(gsdf_mesh_opts, shard_rank) == 4 && offsetof(gsdf_mesh_opts, shard_count) == 8, "gsdf_mesh_opts head");
GSDF_ABI_ASSERT(offsetof(gsdf_mesh_opts, max_tris) == 16 && offsetof(gsdf_mesh_opts, stream) == 24, "gsdf_mesh_opts middle");
GSDF_ABI_ASSERT(offsetof(gsdf_mesh_opts, share_corners) == 32 && offsetof(gsdf_mesh_opts, host_output) == 36 && offsetof(gsdf_mesh_opts, payload) == 40, "gsdf_mesh_opts tail");
GSDF_ABI_ASSERT(sizeof(gsdf_mesh_stats) == 128, "gsdf_mesh_stats is 128 bytes");
GSDF_ABI_ASSERT(offsetof(gsdf_mesh_stats, n_tris) == 0 && offsetof(gsdf_mesh_stats, evals) == 8 && offsetof(gsdf_mesh_stats, pruned_leaves) == 16, "gsdf_mesh_stats counters");
GSDF_ABI_ASSERT(offsetof(gsdf_mesh_stats, leaf_cubes) == 24 && offsetof(gsdf_mesh_stats, active_leaves) == 32 && offsetof(gsdf_mesh_stats, levels) == 40, "gsdf_mesh_stats counters 2");
GSDF_ABI_ASSERT(offsetof(gsdf_mesh_stats, origin) == 44 && offsetof(gsdf_mesh_stats, res) == 56 && offsetof(gsdf_mesh_stats, ms_total) == 64, "gsdf_mesh_stats lattice");
GSDF_ABI_ASSERT(offsetof(gsdf_mesh_stats, ms_prune) == 72 && offsetof(gsdf_mesh_stats, ms_leaf) == 80 && offsetof(gsdf_mesh_stats, ms_march) == 88, "gsdf_mesh_stats times");
GSDF_ABI_ASSERT(offsetof(gsdf_mesh_stats, evals_prune) == 96 && offsetof(gsdf_mesh_stats, evals_leaf) == 104 && offsetof(gsdf_mesh_stats, ms_emit) == 112 && offsetof(gsdf_mesh_stats, cut_leaves) == 120, "gsdf_mesh_stats tail");
GSDF_ABI_ASSERT(GSDF_ERR_EMPTY_BUFFERS == -1 && GSDF_ERR_LENGTH_MISMATCH == -2 && GSDF_ERR_SHORT_BUFFER == -9 && GSDF_ERR_CAPACITY == -10, "status codes are part of the ABI");

int gsdf_hip_mesh_octree(gsdf_program* p, float res, const gsdf_mesh_opts* opts, gsdf_mesh** out);
/* The same in two halves, for a caller with several meshes to make (a part at several resolutions, a batch of parts on one
 * program): _start enqueues the whole chain of kernels and returns; _wait blocks until that mesh is complete. Up to three meshes
 * of a program may be in flight, each with a workspace and a stream of its own (opts.stream, if given, takes them all): the ~30 us a
 * blocking call spends between a mesh's last kernel and the next mesh's first (completion wake-up, the caller's bookkeeping,
 * launch latency) pass under a busy GPU, and the latency-bound top and tail of one mesh (centre tests, marching cubes over the
 * records) run beside the others' evaluating kernel: 0.34 ms per mesh with three in flight, 0.37 with two, 0.45 one at a time
 * (npt-flange resdiv 1600).
 * gsdf_hip_mesh_octree is _start followed by _wait. No other mesher call on the program while a job is in flight. */
typedef struct gsdf_mesh_job gsdf_mesh_job;
int gsdf_hip_mesh_octree_start(gsdf_program* p, float res, const gsdf_mesh_opts* opts, gsdf_mesh_job** job);
int gsdf_hip_mesh_octree_wait(gsdf_mesh_job* job, gsdf_mesh** out);
/* Dual contouring (least-squares vertex placement; chiseled = DualContourLeastSquares.Chiseled). The result is a
 * gsdf_mesh like the octree mesher's (stats: leaf_cubes = kept cubes, active_leaves = active edges; evals = the evaluations
 * performed -- fewer than the reference's: lattice blocks an interval evaluation proves empty are not swept, and a kept cube's own
 * origin is evaluated once, by the sweep, not again with its three edge ends). Multi-GPU: rank
 * shard_rank of shard_count emits the quads of its z-slab of the lattice (one-cube halo recomputed, nothing exchanged). */
int gsdf_hip_mesh_dualcontour(gsdf_program* p, float res, int chiseled, int shard_rank, int shard_count, void* stream,
                              gsdf_mesh** out);
/* glrender.FlatRenderer (glrender/flatrenderer.go:36-256; the renderer gsdfaux.RenderShader3D picks for CPU runs,
 * gsdfaux/gsdfaux.go:160-168): the SDF on every corner of the ceil(size/res)+1 lattice of the 1.01-scaled bounds into a
 * device-resident grid, then marching cubes of every cube whose first corner passes |d| <= 2*sqrt3*res. Same triangle
 * set as the reference's ReadTriangles loop (order differs). Stats: evals = lattice corners (FlatRenderer.Evaluations),
 * leaf_cubes = cubes, active_leaves = cubes passing the first-corner test, ms_leaf = lattice pass, ms_march = marching
 * pass, levels = 0. Multi-GPU: rank shard_rank of shard_count takes a z-slab of cubes (the reference's goroutine split,
 * :120-122), nothing exchanged. */
int gsdf_hip_mesh_flat(gsdf_program* p, float res, int shard_rank, int shard_count, void* stream, gsdf_mesh** out);
/* minecraftRender (glrender/dual_contour.go:297-403; unexported there, exercised by glrender_test.go:55-81): every level-1 cube of the
 * top cube over Bounds() has its origin and its +x, +y, +z edge ends evaluated; an edge whose ends differ in sign contributes the
 * square face across it, two triangles wound by which end is inside. Triangles in no particular order (the reference's is the order of
 * its breadth-first decomposition); stats: n_tris, evals = 4 per cube, levels. At most 9 levels (every cube is evaluated). */
int gsdf_hip_mesh_minecraft(gsdf_program* p, float res, void* stream, gsdf_mesh** out);
int gsdf_hip_mesh_stats_get(const gsdf_mesh* m, gsdf_mesh_stats* st);
/* Device time of the mesher's stages, where it records them (dual contouring: dc_origin, dc_edges, dc_normals, dc_place, dc_quads;
 * HIP events between the stages): *n = number of stages, ms / names (optional, cap entries) = milliseconds and kernel names. */
int gsdf_hip_mesh_stage_ms(const gsdf_mesh* m, double* ms, const char** names, int cap, int* n);
/* What the mesh holds: GSDF_PAYLOAD_TRIANGLES or GSDF_PAYLOAD_RECORDS (gsdf_mesh_opts.payload); *n_records / *payload_bytes
 * (optional) = its cut-leaf records and the size of their packed form (0 for a mesh of triangles). */
int gsdf_hip_mesh_payload(const gsdf_mesh* m, uint64_t* n_records, uint64_t* payload_bytes);
/* Marching cubes over a mesh's packed records, in place: afterwards it holds triangles (stats.n_tris was known before) and
 * every accessor below works. No-op on a mesh of triangles. glrender/marchcubes.go:14-98. */
int gsdf_hip_mesh_march(gsdf_mesh* m);
/* Copy triangles [first, first+count) to host memory: 9 floats (36 B) each = ms3.Triangle. */
int gsdf_hip_mesh_read(const gsdf_mesh* m, uint64_t first, uint64_t count, float* dst);
/* Device pointer to the triangle array (for RCCL gathers / further device work). */
const float* gsdf_hip_mesh_dev_tris(const gsdf_mesh* m);
/* Binary STL (84 + 50*n bytes) built on device into dst (host). dst_cap must be >= that size. */
int gsdf_hip_mesh_stl(const gsdf_mesh* m, uint8_t* dst, size_t dst_cap);
/* Zero-copy result views (no reference counterpart: the reference drains a renderer through ReadTriangles into a
 * 4096-triangle buffer and appends, glrender/glrender.go:20-36, and WriteBinarySTL issues one Write per triangle,
 * glrender/stl.go:40-58). The whole result is moved once, by DMA, into pinned host memory owned by the mesh: *tris is
 * n_tris x 9 floats = []ms3.Triangle, *stl the complete binary STL file (84 + 50 n bytes, records built on device).
 * Valid until gsdf_hip_mesh_destroy; repeated calls return the same memory. */
int gsdf_hip_mesh_host_tris(gsdf_mesh* m, const float** tris);
int gsdf_hip_mesh_host_stl(gsdf_mesh* m, const uint8_t** stl, size_t* len);
/* Releases the mesh. Its device buffers go to a per-process pool that the next meshes (and gathers) draw from -- up to 16 idle
 * buffers are kept (environment: GSDF_HIP_POOL_MAX), beyond that the smallest is freed. */
void gsdf_hip_mesh_destroy(gsdf_mesh* m);

/* ---- multi-GPU (one process per GPU). The meshers shard with NO data-path collective (shard_rank / shard_count above); the
 * one exchange is the final variable-length gather of the ranks' results, over xGMI, inside this library: a Go caller needs no
 * Python for it. Replaces nothing in the reference (single device; its analogue of the split is the goroutine split of
 * glrender/flatrenderer.go:120-122); SURVEY.md section 8(e).
 *   rank 0: gsdf_hip_comm_unique_id(id) -> ship the 128 bytes to the other ranks by any means (file, socket, MPI, env)
 *   every rank (after gsdf_hip_init(device)): gsdf_hip_comm_create(id, rank, world, &comm)            [collective]
 *   per mesh: gsdf_hip_mesh_gatherv(local_mesh, comm, &all, counts)                                   [collective]
 * A gather = an all-gather of four counts per rank, then the transfers gsdf_hip_gather_plan lists for those counts as ONE group
 * of point-to-point sends / receives (ncclSend / ncclRecv: xGMI is one link per peer, every rank feeds all its links at once),
 * every rank's payload landing at offset sum(bytes of the ranks before it): no padding, no staging copies. What moves is the
 * meshes' payload (gsdf_mesh_opts.payload): triangles, or packed cut-leaf records -- 20 instead of 36 bytes per triangle on the
 * wire -- which the receiving ranks march into triangles behind the transfer. The result is a gsdf_mesh holding the triangles of
 * ALL ranks in rank order (device resident; every mesh accessor works on it). librccl is loaded at first use; without it these
 * calls fail with GSDF_ERR_HIP and everything else works. GSDF_HIP_COMM=loopback (environment, read by gsdf_hip_comm_unique_id)
 * selects an in-process transport instead -- the ranks are threads of one process on one GPU, a transfer is a device copy
 * ordered by events -- with which the whole path runs at any world size on a one-GPU box (the tests use it). World size <= 64. */
typedef struct gsdf_comm gsdf_comm;
#define GSDF_COMM_ID_BYTES 128
int gsdf_hip_comm_unique_id(uint8_t id[GSDF_COMM_ID_BYTES]);
int gsdf_hip_comm_create(const uint8_t id[GSDF_COMM_ID_BYTES], int rank, int world, gsdf_comm** out);
int gsdf_hip_comm_rank(const gsdf_comm* c);
int gsdf_hip_comm_world(const gsdf_comm* c);
const char* gsdf_hip_comm_transport(const gsdf_comm* c); /* "rccl" or "loopback" */
/* Sum over all ranks, in place, of n host values (Evaluations(), TotalPruned(), triangle totals). Collective. */
int gsdf_hip_comm_allreduce_sum_u64(gsdf_comm* c, uint64_t* vals, size_t n);
/* counts (optional): world entries, triangles contributed by each rank. */
int gsdf_hip_mesh_gatherv(const gsdf_mesh* m, gsdf_comm* c, gsdf_mesh** out, uint64_t* counts);
/* The same with a choice of who receives, and in two halves so that the payload can move while the caller meshes its next
 * part. Every rank of an all-gather INGESTS (world-1)/world of the whole mesh over its xGMI links -- several times what a rank
 * takes to mesh its share -- so the gather, not the meshing, bounds a step that ends in one (DESIGN.md section 7 has the numbers):
 *   GSDF_GATHER_ALL   every rank gets everything;
 *   GSDF_GATHER_ROOT  only `root` does (the other ranks' links carry their own shard only);
 *   GSDF_GATHER_NONE  counts only: every rank keeps its shard where it is.
 * _start: collective; returns when the counts are exchanged and the payload (and, for records, the marching pass behind it) is
 * enqueued on the communicator's own stream. `m` may be destroyed right away: its buffers are kept until the payload has moved.
 * _wait: blocks until the result is complete; *out = the gathered mesh (NULL on ranks that receive nothing), counts[world]
 * (triangles per rank), st (all optional). */
enum { GSDF_GATHER_ALL = 0, GSDF_GATHER_ROOT = 1, GSDF_GATHER_NONE = 2 };
typedef struct gsdf_gather gsdf_gather;
typedef struct gsdf_gather_stats {
  double ms_counts;         /* the counts exchange (all-gather of four u64 + readback), HIP events on the communicator's stream */
  double ms_payload;        /* the payload, first byte enqueued to last byte arrived */
  uint64_t bytes_sent;      /* of this rank's own payload, over all its links */
  uint64_t bytes_received;  /* of the other ranks' payloads */
  double ms_march;          /* records payload: marching cubes over the gathered records (0 for triangles) */
} gsdf_gather_stats;
GSDF_ABI_ASSERT(sizeof(gsdf_gather_stats) == 40, "gsdf_gather_stats is 40 bytes");
int gsdf_hip_mesh_gatherv_start(const gsdf_mesh* m, gsdf_comm* c, int mode, int root, gsdf_gather** pending);
int gsdf_hip_mesh_gatherv_wait(gsdf_gather* pending, gsdf_mesh** out, uint64_t* counts, gsdf_gather_stats* st);
void gsdf_hip_comm_destroy(gsdf_comm* c);

/* Host-only (runs without a GPU), pure: the transfers rank `rank` of `world` performs in a gather of payloads of
 * bytes_per_rank[r] bytes -- what gsdf_hip_mesh_gatherv_start executes as one group. Layout of the gathered buffer: rank-major,
 * payload r at offset sum(bytes_per_rank[< r]). ops (ops_cap entries; NULL to count) receives
 *   GSDF_GOP_COPY  this rank's own payload [src_off, +bytes) -> gathered buffer at dst_off   (a device copy)
 *   GSDF_GOP_SEND  this rank's own payload [src_off, +bytes) -> rank `peer`
 *   GSDF_GOP_RECV  bytes from rank `peer` -> gathered buffer at dst_off
 * in an order in which, executed entry by entry with non-blocking sends, the ranks' lists match up (peers are visited in
 * rotated order: rank+1, rank+2, ...). Ranks with nothing to contribute appear in nobody's list. *total_bytes = size of this
 * rank's gathered buffer (0 if it receives nothing). tests/test_gather_gloo.py runs these lists with gloo on CPU. */
enum { GSDF_GOP_COPY = 0, GSDF_GOP_SEND = 1, GSDF_GOP_RECV = 2 };
typedef struct gsdf_gather_op {
  int32_t kind;
  int32_t peer;
  uint64_t src_off;
  uint64_t dst_off;
  uint64_t bytes;
} gsdf_gather_op;
GSDF_ABI_ASSERT(sizeof(gsdf_gather_op) == 32, "gsdf_gather_op is 32 bytes");
int gsdf_hip_gather_plan(const uint64_t* bytes_per_rank, int world, int rank, int mode, int root, gsdf_gather_op* ops, size_t ops_cap,
                         size_t* n_ops, uint64_t* total_bytes);

/* Host-only helper (runs without a GPU): owner rank of octree brick (x,y,z) under the multi-GPU partition
 * gsdf_hip_mesh_octree applies on device -- a pure function of the coordinates, so ranks never communicate. */
uint32_t gsdf_hip_brick_owner(uint32_t x, uint32_t y, uint32_t z, uint32_t count);
/* Host-only helper: the z-slab [*lo, *hi) of n lattice planes owned by rank `rank` of `count` in gsdf_hip_mesh_flat (cube
 * planes) and gsdf_hip_mesh_dualcontour (cell planes): contiguous, disjoint, covering [0, n) -- the reference's goroutine
 * split of the flat lattice (flatrenderer.go:120-122). */
void gsdf_hip_slab_range(uint32_t n, uint32_t rank, uint32_t count, uint32_t* lo, uint32_t* hi);

#ifdef __cplusplus
}
#endif
#endif
