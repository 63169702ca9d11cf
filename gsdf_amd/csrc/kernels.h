// kernels.h -- every gfx950 kernel of the MI355X SDF backend (device code only), by phase: kernels_common.h, kernels_eval.h,
// kernels_octree.h, kernels_flat.h, kernels_dc.h, kernels_stl.h.
//
// Compiled twice: ahead of time by hipcc as part of abi_eval.hip and abi_mesh.hip (evaluator = the wave-uniform interpreter of
// interp.h), and at run time -- by the installed hipcc out of process, else hiprtc -- for one lowered program (GSDF_SPECIALIZED: interp.h then takes sdf_eval from
// the generated gsdf_spec_gen.h, the same instruction bodies laid out straight-line with every parameter a literal).
// Nothing here may depend on host headers.
//
// Kernels (all wave64, 256-thread workgroups, grid-stride with wave-uniform trip counts so that the
// interpreter's program counter stays scalar):
//   eval_kernel<DIM>     dist[i] = SDF(pos[i])                      gleval SDF3/SDF2.Evaluate
//   prune_kernel         octree level: centre sample, keep unless the field's bounds over the cube exclude 0
//                        (|d| >= size*sqrt3/2 for a distance field), block-wide compaction of survivors
//                                                                   glrender/octreerenderer.go:240-284
//   leaf_kernel          8 leaf corners (corner 0 first, reject |d0| > 2*sqrt3*res) + marching cubes
//                        with the LDS triangle table; triangles are built one per lane from an LDS
//                        owner list (mc_emit_balanced), staged in LDS and flushed coalesced
//                                                                   glrender/marchcubes.go:14-98
//   flat_grid_kernel     SDF on every corner of the flat lattice    glrender/flatrenderer.go:103-182
//                        (+ two bits per corner: d < 0, |d| <= 2 sqrt3 res)
//   flat_cut_scan_kernel, flat_march_list_kernel
//                        marching cubes of every lattice cube: cut cubes found from the bit planes, listed, marched
//                                                                   glrender/flatrenderer.go:186-256
//   flat_march_kernel    the same from the float grid alone (GSDF_HIP_FLAT_STREAM=1; rounds 1-2)
//   dc_*_kernel          dual contouring stages                     glrender/dual_contour*.go
//   stl_kernel           50-byte STL records staged through LDS     glrender/stl.go:15-62
//   normals_kernel       central differences                        gleval/gleval.go:53-108
#pragma once
#include "kernels_common.h"
#include "kernels_eval.h"
#include "kernels_octree.h"
#include "kernels_flat.h"
#include "kernels_dc.h"
#include "kernels_stl.h"
