/*
 * orc_eval.h -- ORACLE public interface (test infrastructure only).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything in oracle/.
 */
#ifndef ORC_EVAL_H
#define ORC_EVAL_H
#include <stddef.h>
#include <stdint.h>

#include "../include/gsdf_program.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_sdf orc_sdf;
typedef struct orc_pool orc_pool;

orc_sdf* orc_sdf_create(const gsdf_tree* t); /* deep copy; NULL on malformed tree */
void orc_sdf_destroy(orc_sdf* s);
void orc_sdf_bounds(const orc_sdf* s, float bb[6]);
int orc_sdf_root_is2d(const orc_sdf* s);

orc_pool* orc_pool_create(size_t min_alloc); /* gleval.VecPool ; SetMinAllocationLen */
void orc_pool_destroy(orc_pool* p);

/* pos: n*3 floats (xyz AoS) / n*2 floats; dist: n floats. 0 on success. */
int orc_eval3(const orc_sdf* s, orc_pool* vp, const float* pos, float* dist, size_t n);
int orc_eval2(const orc_sdf* s, orc_pool* vp, const float* pos, float* dist, size_t n);
/* lo[i] <= field <= hi[i] over the ball of radius h around pos[i] (interval evaluation; see orc_eval.c). Not a reference
 * function: what the octree's centre tests need for fields that are not 1-Lipschitz. */
int orc_eval3_bounds(const orc_sdf* s, orc_pool* vp, const float* pos, float* lo, float* hi, size_t n, float h);

/* ---- renderers (orc_render.c) ---- */
typedef struct orc_mesh {
  float* tris;     /* 9 floats per triangle, malloc'ed; free with orc_mesh_free */
  uint64_t n_tris;
  uint64_t cap;
  uint64_t evals;  /* SDF evaluations performed */
  uint64_t pruned; /* leaf cubes pruned (octree) */
  int levels;      /* octree levels */
  int nx, ny, nz;  /* flat grid cubes per axis */
  double t_eval_s; /* seconds in SDF evaluation (flat: evalGrid) */
  double t_march_s;
  /* octree only; not reference quantities -- what the device's share_corners options must report (include/gsdf_hip.h): over the
   * surviving Level-3 cubes (bricks of 4x4x4 leaves), the corner evaluations left when every bitwise-distinct z row of a brick
   * is evaluated once (rows: 64 columns x 5..8 rows), every bitwise-distinct lattice point once (points), and the latter in lane
   * slots of a 64-wide wave: passes of 256 with a tail of 64 / 128 / 256 (points_tails), passes of 256 only (points_256) */
  uint64_t evals_rows, evals_points, evals_points_tails, evals_points_256;
} orc_mesh;
void orc_mesh_free(orc_mesh* m);

/* glrender.FlatRenderer (flatrenderer.go): batch = evalBufferSize, nthreads = numParallel. */
int orc_render_flat(const orc_sdf* s, float res, int batch, int nthreads, orc_mesh* out);
/* glrender.Octree (octreerenderer.go) with every Level>=3 cube centre-tested. prune = 0: no tests; 1: every level;
 * else bit L = test Level L. A cube is dropped iff the field's bounds over it (orc_eval3_bounds at the centre, radius
 * size * sqrt3/2) exclude 0 -- for a true distance field exactly the reference's |d| >= size * sqrt3/2 (:270-273), and
 * still surface-preserving for fields that grow faster than distance (twists, screws, non-rigid transforms).
 * ORC_PRUNE_ASSUME_SDF (bit 30): the reference's predicate verbatim, whatever the field. */
#define ORC_PRUNE_ASSUME_SDF (1 << 30)
int orc_render_octree(const orc_sdf* s, float res, int batch, int prune, orc_mesh* out);
/* glrender.DualContourRenderer + DualContourLeastSquares (dual_contour.go, dual_contour_vertexplacement.go) */
int orc_render_dualcontour(const orc_sdf* s, float res, int chiseled, orc_mesh* out);
/* glrender/dual_contour.go:297-403 minecraftRender */
int orc_render_minecraft(const orc_sdf* s, float res, orc_mesh* out);
/* glrender.WriteBinarySTL (stl.go:15-62): writes 84+50*n bytes into dst (caller sized). */
size_t orc_stl_size(uint64_t n_tris);
int orc_write_stl(const float* tris, uint64_t n_tris, uint8_t* dst);
/* marchCubes on explicit cubes: pos 8*3*ncubes, dist 8*ncubes -> tris (cap 5*ncubes*9 floats). */
uint64_t orc_march_cubes(const float* pos, const float* dist, uint64_t ncubes, float res, float* tris);
/* gleval.NormalsCentralDiff (gleval/gleval.go:53-108) */
int orc_normals_central_diff(const orc_sdf* s, orc_pool* vp, const float* pos, float* normals, size_t n, float step);
/* table access for tests */
const uint16_t* orc_mc_edge_table(void);
const int8_t* orc_mc_tri_table(void); /* 256 x 16, -1 terminated */

#ifdef __cplusplus
}
#endif
#endif
