// interp.h -- wave-uniform SDF interpreter for gfx950 (see dev_ops.h for the instruction set).
//
// sdf_eval() executes the lowered program for ONE point per lane. Control flow is uniform across the
// wave (the program is straight-line), so the program counter, opcodes and all node parameters are
// fetched with scalar loads from the constant address space and live in SGPRs; VALU instructions
// take them as scalar operands. Per-lane temporaries (saved positions / partial distances) go to the
// lane's private LDS column: lds[slot * nthreads + tid] -> one bank per lane, conflict-free.
//
// Formula order per opcode follows /root/reference/cpu_evaluators.go (line cites in compile.cpp).
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

#include "dev_math.h"
#include "dev_ops.h"

namespace gsdf_dev {

typedef const uint32_t __attribute__((address_space(4))) * code_ptr;  // constant AS -> s_load
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const v4f __attribute__((address_space(4))) * f4ptr;

struct P3 { float x, y, z; };

// Lane number within the wave (mbcnt pair; `threadIdx.x & 63` is equivalent here but, measured, costs the leaf kernels
// registers: knurled-cylinder's leaf_kernel<4,3> 168 VGPRs + 1 spill instead of 168 + 0).
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Slot s of point k of this lane: one float per lane per (slot, k) -> conflict-free columns.
#define LDSF(slot) lds[((slot) * K + kp) * nthreads]
// Each instruction is applied to the K points this lane carries before the next dispatch: the decode,
// the scalar parameter loads and the branch are paid once per K points, and the K independent
// dependency chains hide VALU/transcendental latency at low occupancy.
#define KLOOP _Pragma("unroll") for (int kp = 0; kp < K; ++kp)

// poly2D (cpu_evaluators.go:793-818), vertex-major: each edge record (8 dwords {v1x v1y ex ey |e|^2 v2y 1/|e|^2 -},
// 32-byte aligned: two s_load_dwordx4) is fetched once and applied to the K points. FAST divides by the
// wave-uniform |e|^2 with the host's correctly rounded reciprocal (dm::div_by_uniform) and tracks the
// smallest/largest |numerator|; it returns false (wave-uniform) if some lane left the proven-exact range.
// keepd / keeps (wave-uniform bit per edge, from poly_cull): edges whose distance part / winding part can matter for some
// point of this wave; the others are provably irrelevant and skipped.
// a wave-uniform value held in a vector register (see poly_edges)
template <bool ON>
__device__ __forceinline__ float in_vgpr(float s) {
  if (!ON) return s;
  float v;
  asm("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
  return v;
}

template <int K, bool FAST>
__device__ __forceinline__ bool poly_edges(code_ptr code, uint32_t q, uint32_t nv, float v0x, float v0y, const P3 (&pv)[K],
                                           float (&d)[K], bool (&neg)[K], const uint64_t keepd = ~0ull,
                                           const uint64_t keeps = ~0ull) {
  using namespace dm;
  // The numerator range guard is shared by the K points of the lane (only a wave-wide verdict is needed):
  // consecutive updates fold into v_min3_f32 / v_max3_f32.
  float nmin = 1.0f, nmax = 1.0f;
  // Winding parity as wave-wide lane masks in SGPRs: the three edge predicates are v_cmp results already, so
  // "all three equal" and the parity update are 4 scalar ops per point with no per-lane select.
  uint64_t negm[K];
  KLOOP {
    float wx0 = pv[kp].x - v0x, wy0 = pv[kp].y - v0y;
    d[kp] = wx0 * wx0 + wy0 * wy0;
    negm[kp] = 0;
  }
  for (uint32_t iv = 0; iv < nv; iv++, q += 8) {
    const bool kd = iv >= 64u || ((keepd >> iv) & 1ull) != 0ull, ks = iv >= 64u || ((keeps >> iv) & 1ull) != 0ull;  // wave-uniform
    if (!kd && !ks) continue;
    const f4ptr er = (f4ptr)(code + q);
    const v4f e0 = er[0], e1 = er[1];
    // The edge's constants arrive in scalar registers, and on gfx950 a float add / multiply / fma with a scalar-register operand
    // issues at 4.2 cycles instead of 2.4 (tools/ubench/class_rate.hip: "sgpr src0"): with two or more points per lane the thirteen
    // such operations per point are worth six moves to vector registers per edge. Same values, same operations.
    const float v1x = in_vgpr<(K >= 2)>(e0.x), v1y = in_vgpr<(K >= 2)>(e0.y), ex = in_vgpr<(K >= 2)>(e0.z), ey = in_vgpr<(K >= 2)>(e0.w),
                n2e = in_vgpr<(K >= 2)>(e1.x), v2y = e1.y, rn2e = in_vgpr<(K >= 2)>(e1.z);
    float wx[K], wy[K];
    KLOOP { wx[kp] = pv[kp].x - v1x; wy[kp] = pv[kp].y - v1y; }
    // The two parts under ONE wave-uniform branch each, over the K points (round 6: with the tests inside the point loop the compiler
    // branched per point and carried the two uniform flags through vector registers -- four v_cndmask / v_cmp per edge, and a move
    // behind every v_min), and the K numerators guarded together: nested minima / maxima fold into v_min3_f32 / v_max3_f32.
    if (kd) {
      float num[K];
      KLOOP num[kp] = wx[kp] * ex + wy[kp] * ey;
      if (FAST) {
        if (K == 4) {
          nmin = minf(minf(minf(nmin, absf(num[0])), absf(num[K > 1 ? 1 : 0])), minf(absf(num[K > 2 ? 2 : 0]), absf(num[K > 3 ? 3 : 0])));
          nmax = maxf(maxf(maxf(nmax, absf(num[0])), absf(num[K > 1 ? 1 : 0])), maxf(absf(num[K > 2 ? 2 : 0]), absf(num[K > 3 ? 3 : 0])));
        } else {
          KLOOP { nmin = minf(nmin, absf(num[kp])); nmax = maxf(nmax, absf(num[kp])); }
        }
      }
      KLOOP {
        const float quo = FAST ? div_by_uniform(num[kp], n2e, rn2e) : num[kp] / n2e;
        // clamp(v,0,1) as med3: differs from the reference's if-chain only in the sign of a zero t, which cannot
        // reach d (t only scales e before the square)
        const float t = __builtin_amdgcn_fmed3f(quo, 0.f, 1.f);
        const float bx = wx[kp] - t * ex, by = wy[kp] - t * ey;
        d[kp] = minf(d[kp], bx * bx + by * by);
      }
    }
    if (ks) {
      KLOOP {
        const float py = pv[kp].y;
        const uint64_t b1 = __builtin_amdgcn_ballot_w64(py >= v1y), b2 = __builtin_amdgcn_ballot_w64(py < v2y),
                       b3 = __builtin_amdgcn_ballot_w64(ex * wy[kp] > ey * wx[kp]);
        negm[kp] ^= ~((b1 ^ b2) | (b2 ^ b3));  // flip where all three are true or all three are false
      }
    }
  }
  {  // the mask is the per-lane predicate (shift/and rather than the inverse-ballot builtin: the run-time compiler
     // of a process that loaded an older ROCm first, e.g. PyTorch's bundled one, does not have it)
    const uint32_t lane = lane_id();
    KLOOP neg[kp] = ((negm[kp] >> lane) & 1ull) != 0ull;
  }
  if (!FAST) return true;
  return __all(nmin >= 8.0779357e-28f /* 2^-90 */ && nmax <= 1.2379400e+27f /* 2^90 */);
}

// Edge culling for poly2D where a wave is spatially compact: the leaf kernel's bricks (one wave = the corners of 4x4x4
// leaves). Lattice sweeps (64 columns in a row per wave) span too much of the polygon for it to pay: flat lattice pass
// 3.62 -> 3.90 ms with it.
// The polygon's distance is sqrt(min over edges of the squared distance) and its sign the parity of edge crossings:
// an edge that cannot hold the minimum for ANY point of the wave, or that no point of the wave can cross, does not
// change a single bit of the result. With B = the bounding box of the wave's K*64 points (centre c, half diagonal rb)
// and dc(e) = distance from c to edge e, every point p of the wave has dist(p, e) in [dc(e) - rb, dc(e) + rb]; so with
// U = min over e of dc(e) + rb, an edge with dc(e) - rb > U is never the nearest one. Both sides are padded by a margin (1e-5 of
// the largest |p - v|, |e|, |c| magnitude entering the edge's own sums -- the float error of the compared quantities is < 1e-6
// of that) that keeps the decision on the safe side: a doubtful edge is evaluated. The winding predicates of an edge are
// (p.y >= v1.y, p.y < v2.y, cross > 0) and flip the sign only when all three agree: if the whole box lies at or above
// both endpoints the first two are (true, false), if it lies below both (false, true), for every point -- no flip,
// decided with exact comparisons. Lane e works out edge e (nv <= 64); NaN/Inf anywhere makes every test fail towards
// "evaluate". For the npt-flange thread profile (12 edges) a brick's wave keeps 2-4 edges.
// Wave64 min / max with DPP row shifts and row broadcasts: v_min / v_max_f32 WITH the DPP operand, one instruction per step (no
// LDS crossbar, no address registers, no lgkmcnt wait). Inclusive scan within each row of 16 lanes (row_shr 1, 2, 4, 8; lanes
// without a source are disabled and keep their own value), then lane 15 of rows 0 and 2 into rows 1 and 3 (row_bcast:15), then
// lane 31 into rows 2 and 3 (row_bcast:31): lane 63 holds the result, read back as a scalar. All 64 lanes must be active.
// Written as assembly because the compiler does not fold a DPP move into a float min / max (its combiner knows identities for
// the integer operations only): v_mov + v_mov_dpp + v_min per step -- three instructions and a wait state where one does, and
// the polygon culling below is six such reductions per brick and pass (round 4: 250 of the ~700 instructions a threaded brick
// of npt-flange costs per pass). A DPP operand needs two wait states behind the instruction that wrote its register: the four
// reductions of a box are interleaved (three independent instructions between a step and the next of the same chain), a single
// reduction pays s_nop 1 per step.
#define GSDF_DPP4(op_a, op_b, op_c, op_d, ctrl)          \
  op_a "_dpp %0, %0, %0 " ctrl "\n\t" op_b "_dpp %1, %1, %1 " ctrl "\n\t" op_c "_dpp %2, %2, %2 " ctrl "\n\t" op_d "_dpp %3, %3, %3 " ctrl "\n\t"
// a, c: minima; b, d: maxima -- over the wave, returned in every lane's copy as wave-uniform values
__device__ __forceinline__ void wave_box(float& a, float& b, float& c, float& d) {
  asm volatile("s_nop 1\n\t"
               GSDF_DPP4("v_min_f32", "v_max_f32", "v_min_f32", "v_max_f32", "row_shr:1 row_mask:0xf bank_mask:0xf")
               GSDF_DPP4("v_min_f32", "v_max_f32", "v_min_f32", "v_max_f32", "row_shr:2 row_mask:0xf bank_mask:0xf")
               GSDF_DPP4("v_min_f32", "v_max_f32", "v_min_f32", "v_max_f32", "row_shr:4 row_mask:0xf bank_mask:0xf")
               GSDF_DPP4("v_min_f32", "v_max_f32", "v_min_f32", "v_max_f32", "row_shr:8 row_mask:0xf bank_mask:0xf")
               GSDF_DPP4("v_min_f32", "v_max_f32", "v_min_f32", "v_max_f32", "row_bcast:15 row_mask:0xa bank_mask:0xf")
               GSDF_DPP4("v_min_f32", "v_max_f32", "v_min_f32", "v_max_f32", "row_bcast:31 row_mask:0xc bank_mask:0xf")
               "s_nop 1"
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  a = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(a), 63));
  b = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(b), 63));
  c = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(c), 63));
  d = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(d), 63));
}
#undef GSDF_DPP4
template <bool MAX>
__device__ __forceinline__ float wave_minmax(float v) {
  if (MAX) {
#define GSDF_DPP_CHAIN(op)                                                                                                   \
  asm volatile("s_nop 1\n\t" op "_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                                  \
               "s_nop 1\n\t" op "_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"                                  \
               "s_nop 1\n\t" op "_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"                                  \
               "s_nop 1\n\t" op "_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"                                  \
               "s_nop 1\n\t" op "_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                               \
               "s_nop 1\n\t" op "_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"                               \
               "s_nop 1"                                                                                                     \
               : "+v"(v))
    GSDF_DPP_CHAIN("v_max_f32");
  } else {
    GSDF_DPP_CHAIN("v_min_f32");
  }
#undef GSDF_DPP_CHAIN
  return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 63));
}

// recip (wave-uniform: bit 31 of the polygon's header, every |e|^2 in [2^-30, 2^30] and its reciprocal in the record): the test's own
// arithmetic may then be approximate -- the quotient as a product with the host's reciprocal (within 2^-23 of t in [0, 1]: 1.2e-7 |e| in
// b), the two square roots as the bare v_sqrt_f32 (1 ulp; a subnormal argument costs at most 1.1e-19 absolute, against m >= 1e-5 *
// 2^-15) -- each far inside the margin m below, which is what keeps a doubtful edge. 12 + 2 x 16 instructions become 1 + 2 x 1.
template <int K>
__device__ __forceinline__ void poly_cull(code_ptr code, uint32_t q0, uint32_t nv, const P3 (&pv)[K], uint64_t& keepd,
                                          uint64_t& keeps, const bool recip = false) {
  using namespace dm;
  float x0 = pv[0].x, x1 = pv[0].x, y0 = pv[0].y, y1 = pv[0].y;
  KLOOP {
    x0 = minf(x0, pv[kp].x); x1 = maxf(x1, pv[kp].x);
    y0 = minf(y0, pv[kp].y); y1 = maxf(y1, pv[kp].y);
  }
  wave_box(x0, x1, y0, y1);
  const uint32_t lane = lane_id();
  const bool valid = lane < nv;
  const v4f* rec = (const v4f*)((const float*)(uintptr_t)code + q0 + 8u * (valid ? lane : 0u));  // this lane's edge record (32-byte aligned)
  const v4f r0 = rec[0], r1 = rec[1];
  const float v1x = r0.x, v1y = r0.y, ex = r0.z, ey = r0.w, n2e = r1.x, v2y = r1.y;
  const float cx = 0.5f * (x0 + x1), cy = 0.5f * (y0 + y1), hx = 0.5f * (x1 - x0), hy = 0.5f * (y1 - y0);
#ifdef GSDF_NO_CULL_APPROX
  const bool approx = false;
#else
  const bool approx = recip;
#endif
  const float rb2 = hx * hx + hy * hy;
  const float rb = approx ? __builtin_amdgcn_sqrtf(rb2) : sqrtf_(rb2);
  const float wx = cx - v1x, wy = cy - v1y;
  const float num = wx * ex + wy * ey;
  const float t = __builtin_amdgcn_fmed3f(approx ? num * r1.z : num / n2e, 0.f, 1.f);
  const float bx = wx - t * ex, by = wy - t * ey;
  const float dc2 = bx * bx + by * by;
  const float dc = approx ? __builtin_amdgcn_sqrtf(dc2) : sqrtf_(dc2);
  // Each edge pads its OWN interval by its OWN error bound: m = 1e-5 of the largest magnitude entering its sums, |w|, |e|, rb, |c|
  // in the 1-norm (no square roots; |c| too: the rounded centre may sit half an ulp of its own magnitude off the box's true
  // centre) -- the float error of dc -+ rb is below 1e-6 of that. Edge e is never the nearest one if its padded lower bound lies
  // above the smallest padded upper bound. (Round 3 padded every edge by the largest m of the wave: one more reduction.)
  const float m = 1.0e-5f * (absf(wx) + absf(wy) + absf(ex) + absf(ey) + rb + absf(cx) + absf(cy));
  const float U = wave_minmax<false>(valid ? (dc + rb) + m : __builtin_inff());
  keepd = __builtin_amdgcn_ballot_w64(valid && !((dc - rb) - m > U));
  keeps = __builtin_amdgcn_ballot_w64(valid && !((y0 >= v1y && y0 >= v2y) || (y1 < v1y && y1 < v2y)));
}

// f(P.x, P.y) of the K points of a lane for the instructions flagged D_FLAG_SHXY (hypot, atan2).
//   not shared : every point on its own;
//   paired     : points 2j / 2j+1 entered with equal x,y -> even points compute, odd points copy;
//   brick      : (K = 4, one wave = the 4x4x4 leaves of one level-3 cube, lane = x + 4y + 16z) the four lanes of an
//                x,y column hold the same two x,y pairs, so lanes with even z evaluate pair 0, lanes with odd z pair 1
//                -- ONE evaluation per lane instead of two -- and every lane fetches both results from the z = 0 / z = 1
//                lanes of its column (ds_bpermute). Same inputs, same operations, same bits.
//   column     : all K points of the lane entered with equal x,y (lattice sweeps: one lattice column on K planes)
//                -> point 0 computes, the others copy.
template <int K, typename F>
__device__ __forceinline__ void xy_shared(const P3 (&pv)[K], float (&out)[K], bool shared, bool brick, F f, bool column = false) {
  if (shared && column) {
    const float v = f(pv[0].x, pv[0].y);
    KLOOP out[kp] = v;
    return;
  }
  if (K == 4 && shared && brick) {
    const uint32_t lane = lane_id();
    const bool odd = ((lane >> 4) & 1u) != 0u;
    const float sx = odd ? pv[K > 2 ? 2 : 0].x : pv[0].x, sy = odd ? pv[K > 2 ? 2 : 0].y : pv[0].y;
    const float v = f(sx, sy);
    const float v0 = __shfl(v, (int)(lane & 15u), 64), v1 = __shfl(v, (int)((lane & 15u) + 16u), 64);
    out[0] = v0; out[K > 1 ? 1 : 0] = v0;
    out[K > 2 ? 2 : 0] = v1; out[K > 3 ? 3 : 0] = v1;
    return;
  }
  KLOOP if (!(kp & 1)) out[kp] = f(pv[kp].x, pv[kp].y);
  if (shared) { KLOOP if (kp & 1) out[kp] = out[kp ? kp - 1 : 0]; }
  else { KLOOP if (kp & 1) out[kp] = f(pv[kp].x, pv[kp].y); }
}
// What a lane keeps between the two passes of a column brick (kernels_octree.h: the 8 corners of a lane's (x, y) column are evaluated
// four z rows at a time): the angle atan2(y, x) of an instruction whose points share x, y -- the same float in both passes, since the
// column is the same and the instruction stream in front of it is -- keyed by the instruction's program counter (one entry: what a
// program with one screw needs; another instruction overwrites it). The two passes sit behind their own wave-level gates, so the
// compiler does not merge the two evaluations itself (round 6: the float64 sequence stood twice in npt-flange's kernel).
struct XYCache {
  float th = 0.0f;
  uint32_t pc = 0xffffffffu;  // wave-uniform
};

// math.Atan2 of the points' (y, x) with xy_shared's sharing: dm::atan2_fast for every point, one wave vote, the reference's own
// sequence (dm::atan2_ref) for the whole wave where a point's result is not decided by the fast route (one wave in ~100).
template <int K>
__device__ __forceinline__ void atan2_shared(const P3 (&pv)[K], float (&out)[K], bool shared, bool brick, bool column) {
  // (The short route is for the kernels built per tree. The interpreter holds every instruction's body in one function: the route's
  // float64 temporaries beside the reference's pushed its one-point-per-lane builds into scratch, tests/test_kernel_resources.py.)
#if defined(GSDF_SPECIALIZED) && !defined(GSDF_NO_ATAN2_FAST)
  // (one angle per lane for all of its points -- a column brick under an axis-aligned frame, npt-flange's thread -- is cheap either
  // way, and the short route's float64 temporaries cost that build its fifth workgroup per CU: the reference's sequence there)
  if (!(shared && column)) {  // (wave-uniform: instruction flags)
    bool ok = true;
    xy_shared<K>(pv, out, shared, brick, [&ok](float x, float y) { bool o; const float v = dm::atan2_fast(y, x, o); ok = ok && o; return v; }, column);
    if (__all(ok)) return;
  }
#endif
  xy_shared<K>(pv, out, shared, brick, [](float x, float y) { return dm::atan2_ref(y, x); }, column);
}

// math.Cos / math.Sin of one angle per lane: dm::cossin_fast, one wave vote, the reference's sequence (dm::cossinf_) for the wave where
// some lane's result is not decided by the short route. (Per-tree builds only, like the short atan2: see atan2_shared.)
__device__ __forceinline__ void cossin_voted(float x, float& c, float& s) {
#if defined(GSDF_SPECIALIZED) && !defined(GSDF_NO_COSSIN_FAST)
  bool ok;
  dm::cossin_fast(x, c, s, ok);
  if (__all(ok)) return;
#endif
  dm::cossinf_(x, c, s);
}

// hypot(P.x,P.y) of every point into hxy[] unless the cache is valid
#define ENSURE_HXY() \
  if (!use_hxy) xy_shared<K>(pv, hxy, sh_xy, brick, [](float x, float y) { return dm::hypotf_(x, y); }, sh_col)

// D_GATE*: lower bound L of a child's field outside its region (compile.cpp: lower_region), per point.
//   box        : Chebyshev distance max over axes of (mn - p, p - mx) -- a lower bound of the Euclidean one.
//   z-cylinder : max(z0 - z, z - z1, rs * (rad - r)), rad <= hypot(x - cx, y - cy): the cached hypot when the host says it
//                is valid, else the octagon estimate max(|x|, |y|, (|x| + |y|) / sqrt 2) >= 0.92 hypot, with the constant
//                rounded down so that float rounding cannot lift it above the true radius.
//                an annulus (rin > 0) also bounds points near the axis: rs * (rin - rad'), rad' >= hypot (max + 0.4142 min).
// Inside the region L <= 0: no claim, the gate stays shut.
template <int K, int DIM>
__device__ __forceinline__ void region_lb_box(const P3 (&pv)[K], float mnx, float mny, float mnz, float mxx, float mxy, float mxz,
                                              float (&L)[K]) {
  using namespace dm;
  KLOOP {
    float l = maxf(maxf(mnx - pv[kp].x, pv[kp].x - mxx), maxf(mny - pv[kp].y, pv[kp].y - mxy));
    if (DIM == 3) l = maxf(l, maxf(mnz - pv[kp].z, pv[kp].z - mxz));
    L[kp] = l;
  }
}
template <int K>
__device__ __forceinline__ void region_lb_zcyl(const P3 (&pv)[K], const float (&hxy)[K], bool use_hxy, float cx, float cy, float r,
                                               float z0, float z1, float rs, float rin, float (&L)[K]) {
  using namespace dm;
  KLOOP {
    float rlo, rhi;  // rlo <= rho <= rhi
    if (use_hxy) {
      rlo = hxy[kp] * 0.999999f; rhi = hxy[kp] * 1.000001f;
    } else {
      const float ax = absf(pv[kp].x - cx), ay = absf(pv[kp].y - cy);
      const float mx = maxf(ax, ay), mn = minf(ax, ay);
      rlo = maxf(mx, 0.70710605f * (ax + ay));  // octagon inside the circle
      rhi = (mx + 0.41421402f * mn) * 1.000001f;  // max + (sqrt2 - 1) min >= hypot (octagon around it), constants rounded up
    }
    float l = maxf(maxf(z0 - pv[kp].z, pv[kp].z - z1), rs * (rlo - r));
    if (rin > 0.0f) l = maxf(l, rs * (rin - rhi));  // wave-uniform test
    L[kp] = l;
  }
}
// box turned about z: Chebyshev distance in the box's own frame (its x axis = (c, s) in the current frame)
template <int K, int DIM>
__device__ __forceinline__ void region_lb_obox(const P3 (&pv)[K], float cx, float cy, float c, float s, float hx, float hy, float z0,
                                               float z1, float (&L)[K]) {
  using namespace dm;
  KLOOP {
    const float dx = pv[kp].x - cx, dy = pv[kp].y - cy;
    float l = maxf(absf(c * dx + s * dy) - hx, absf(c * dy - s * dx) - hy);
    if (DIM == 3) l = maxf(l, maxf(z0 - pv[kp].z, pv[kp].z - z1));
    L[kp] = l;
  }
}
// The gate's verdict: true (wave-uniform) if EVERY point of the wave passes: L > 0 and L > sg * a + kk by the margin (the
// child cannot change its own combine), or -- with the context of the enclosing difference, c = its other operand -- the
// upper bound U = max(a, -L) + k4 of the inner result is negative and c + U <= -ok by the margin (the enclosing combine
// discards the inner result).
// LIP (interval mode: the lane's two points are the two ends of `a` and `c` over one ball): the lane passes if BOTH ends pass the
// first test or BOTH ends pass the second -- each test is monotone in (a, c), so its holding at the two extreme corners is its
// holding for every pair of values in between; one end by one test and the other end by the other test says nothing about
// the values in between. *lane_good (optional) = this lane's own verdict, whatever the rest of the wave says (D_LIP_DOM's
// gate form: the brick mask is per cube).
template <int K, int DIM, bool LIP = false>
__device__ __forceinline__ bool gate_far(const P3 (&pv)[K], const float (&a)[K], const float (&L)[K], float sg, float kk,
                                         const float (&c)[K], bool has_outer, float ok, float k4, bool* lane_good = nullptr) {
  using namespace dm;
  bool far = true, far2 = true;
  KLOOP {
    float S = absf(pv[kp].x) + absf(pv[kp].y);
    if (DIM == 3) S += absf(pv[kp].z);
    const float pad = 2e-6f * S;
    const bool g1 = (L[kp] > 0.0f) && (L[kp] > sg * a[kp] + kk + (1e-3f * (L[kp] + absf(a[kp])) + pad));
    bool g2 = false;
    if (has_outer) {
      const float U = maxf(a[kp], -L[kp]) + k4;
      g2 = (U < 0.0f) && (c[kp] + U <= -ok - (1e-3f * (absf(c[kp]) + absf(U)) + pad));
    }
    if (LIP) { far = far && g1; far2 = far2 && g2; }
    else far = far && (g1 || g2);
  }
  if (LIP) far = far || (has_outer && far2);
  if (lane_good != nullptr) *lane_good = far;
  return __all(far) != 0;
}

// hypot of the K points of a lane for the "outside distance" terms hypot(max(a,0), max(b,0)) of boxes, cylinders, prisms
// and extrusions. Near a face of such a shape one argument is 0 for every point of a (spatially coherent) wave, and
// math32.Hypot(p, 0) is p exactly (q/p = 0, sqrt(1 + 0) = 1, p * 1 = p): the wave then skips the division and the
// square root. Same bits either way; the test is one compare per point and a wave vote.
template <int K>
__device__ __forceinline__ void hypot_k(const float (&a)[K], const float (&b)[K], float (&out)[K]) {
  using namespace dm;
  float hi[K], lo[K];
  bool z = true;
  KLOOP {
    const float p = absf(a[kp]), q = absf(b[kp]);
    hi[kp] = maxf(p, q);
    lo[kp] = minf(p, q);
    z = z && (lo[kp] == 0.0f);
  }
  if (__all(z)) {
    KLOOP out[kp] = hi[kp];
  } else {
    KLOOP {
      const float r = lo[kp] / maxf(hi[kp], 1.401298464324817e-45f);
      out[kp] = hi[kp] * sqrt_1to2(1.0f + r * r);
    }
  }
}

// Blend weight of the smooth combines, h = clamp(0.5 + sign * num / k, 0, 1) for the K points of a lane. Away from the
// blend zone -- |num| >= 0.5005 k for every point of the wave -- the quotient is beyond +-0.5 whatever its rounding, so
// the clamp yields exactly 0 or 1 and the division is skipped (wave vote); the callers' formulas run unchanged on h.
template <int K, int SIGN>
__device__ __forceinline__ void smooth_h(const float (&num)[K], float k, float rk, float (&h)[K]) {
  using namespace dm;
  const float thr = absf(0.5f * k) * 1.001f;
  bool out = k != 0.0f;
  KLOOP out = out && (absf(num[kp]) >= thr);
  if (__all(out)) {
    KLOOP h[kp] = (((num[kp] > 0.0f) == (k > 0.0f)) == (SIGN > 0)) ? 1.0f : 0.0f;
  } else {
    float q[K];
    div_uniform_k<K>(num, k, rk, q);
    KLOOP h[kp] = clampf(SIGN > 0 ? 0.5f + q[kp] : 0.5f - q[kp], 0.f, 1.f);
  }
}

// SHARE = 0: K unrelated points. SHARE = 2 (COLUMN; lattice sweeps, the mesher's column bricks): the K points of a lane enter
// with bitwise equal x,y (one lattice column on K planes): instructions flagged D_FLAG_SHXY compute their f(P.x,P.y) once per
// lane. With `brick` (leaf_eval_kernel's column bricks) the caller also guarantees that point kp has the SAME z in every lane
// of the wave: instructions flagged D_FLAG_SHZ compute their g(P.z) once per wave and point (one lane each).
// SHARE = 1 (PAIRED; the mesher's leaf kernels): the caller passes the corners of one leaf cube in the order
// {0,4,1,5 | 3,7,2,6}, i.e. points 2j and 2j+1 enter with bitwise equal x,y and (K = 4) points j and j+2 with equal z.
// Instructions the host compiler flagged D_FLAG_SHXY / D_FLAG_SHZ then compute their f(P.x,P.y) / g(P.z) once per
// pair and copy it: same inputs, same operation sequence, same bits as evaluating every corner separately.
// LIP (interval mode, dev_ops.h: D_LIP_*; prune_kernel only): K = 2, both points are the same cube centre, point 0 ends up
// with a lower and point 1 with an upper bound of the field over the ball of radius lip_h around it. lip_base = first LDS
// slot of the interval stack (the program's own slots come first).
#define LIP_LO 0
#define LIP_HI (K - 1)
// value -+ radius: an exact distance (or any other 1-Lipschitz term of the current frame) over the ball
#define LIP_WIDEN(lo, hi) { lo = lo - lipR; hi = hi + lipR; }
// D_LIP_DOM (dev_ops.h): which operand of the combine that follows dominates over the whole ball, per lane = per cube. One text for
// the interpreter (below, in the dispatch's default branch) and the specialised build (specialize.cpp emits the macro).
#define GSDF_LIP_DOM_BODY \
        if (LIP) {                                                                                                                     \
          if (lip_fired != nullptr) {                                                                                                  \
            const uint32_t kind = PU(0), ida = PU(1), idb = PU(2), pinfo = PU(4);                                                      \
            const float kk = PF(3);                                                                                                    \
            /* first evaluated operand: lds[slot]; second: R. swap_ab: the formula's second operand b was evaluated first. */          \
            const float f_lo = lds[((slot) * K + LIP_LO) * nthreads], f_hi = lds[((slot) * K + LIP_HI) * nthreads];                    \
            const float alo = swap_ab ? Rv[LIP_LO] : f_lo, ahi = swap_ab ? Rv[LIP_HI] : f_hi;                                          \
            const float blo = swap_ab ? f_lo : Rv[LIP_LO], bhi = swap_ab ? f_hi : Rv[LIP_HI];                                          \
            /* |x| + |y| + |z| of the frame's entry position (the centre): D_LIP_PUSH left it on the interval stack, second column */  \
            const float S = lds[((lip_base + (pinfo & 0xffffu)) * K + LIP_HI) * nthreads];                                             \
            const float pad = kk + 2e-6f * (S + 4.0f * lipR);                                                                          \
            auto GSDF_DOM_M = [&](float u, float v) { return pad + 1e-3f * (absf(u) + absf(v)); };                                     \
            bool sa = false, sb = false;                                                                                               \
            switch (kind) {                                                                                                            \
              case 0u: sb = blo > ahi + GSDF_DOM_M(blo, ahi); sa = alo > bhi + GSDF_DOM_M(alo, bhi); break;                            \
              case 1u: sb = bhi < alo - GSDF_DOM_M(bhi, alo); sa = ahi < blo - GSDF_DOM_M(ahi, blo); break;                            \
              case 2u: sb = blo > -alo + GSDF_DOM_M(blo, alo); sa = ahi < -bhi - GSDF_DOM_M(ahi, bhi); break;                          \
              case 3u: sb = blo - ahi > GSDF_DOM_M(blo, ahi); sa = alo - bhi > GSDF_DOM_M(alo, bhi); break;                            \
              case 4u: sb = blo + alo > GSDF_DOM_M(blo, alo); sa = bhi + ahi < -GSDF_DOM_M(bhi, ahi); break;                           \
              case 5u: sb = alo - bhi > GSDF_DOM_M(alo, bhi); sa = blo - ahi > GSDF_DOM_M(blo, ahi); break;                            \
              default: break;                                                                                                          \
            }                                                                                                                          \
            if (sa && ida < 32u) *lip_fired |= 1u << ida;                                                                              \
            if (sb && idb < 32u) *lip_fired |= 1u << idb;                                                                              \
          }                                                                                                                            \
        }                                                                                                                              \

// The test of a D_GATE* instruction (dev_ops.h), shared by the interpreter's four cases and the specialised build's generated
// text: REGION fills L[] from the gate's region parameters; I0 = index of the `sg` parameter (then kk, oslot, ok, k4, skip word).
// Leaves far_ (wave-uniform: the child is skipped) and L (the substitute) in scope.
#define GSDF_GATE_TEST(REGION, DIM, I0)                                                                                      \
  float L[K];                                                                                                               \
  bool far_ = false;                                                                                                        \
  {                                                                                                                         \
    const uint32_t gid1 = GSDF_GATE_ID1(PU((I0) + 5));                                                                      \
    /* a numbered child under a valid brick mask: the centre test decided for the whole brick (D_SKIP), no per-point test */ \
    const bool untested = !LIP && gid1 != 0u && (bmask & GSDF_BRICK_MASK_VALID) != 0u;                                      \
    if (!untested) {                                                                                                        \
      float a[K], cc[K];                                                                                                    \
      const uint32_t oslot = PU((I0) + 2);                                                                                  \
      const bool has_outer = oslot != 0xffffu; /* wave-uniform */                                                           \
      KLOOP { a[kp] = LDSF(slot); cc[kp] = has_outer ? LDSF(oslot) : 0.0f; }                                                \
      REGION;                                                                                                               \
      if (LIP) KLOOP L[kp] = L[kp] - lipR; /* interval mode: the bound over the whole ball (L is 1-Lipschitz), against both ends of a */ \
      bool lane_good = false;                                                                                               \
      far_ = gate_far<K, DIM, LIP>(pv, a, L, PF(I0), PF((I0) + 1), cc, has_outer, PF((I0) + 3), PF((I0) + 4), &lane_good);  \
      if (LIP && gid1 != 0u && lip_fired != nullptr && lane_good) *lip_fired |= 1u << (gid1 - 1u);                          \
    }                                                                                                                       \
  }
// the skipped child's value: its lower bound (interval mode: the valid interval [bound, "no upper bound"])
#define GSDF_GATE_TAKEN                                                       \
  {                                                                           \
    if (LIP) { Rv[LIP_LO] = L[LIP_LO]; Rv[LIP_HI] = GSDF_SKIP_BIG; }          \
    else { KLOOP Rv[kp] = L[kp]; }                                            \
  }
#ifndef GSDF_SPECIALIZED
template <int K, int SHARE = 0, bool LIP = false>
__device__ __forceinline__ void sdf_eval(code_ptr code, P3 (&pv)[K], float (&Rv)[K],
                                         float* __restrict__ lds /* already offset by tid */, const uint32_t nthreads,
                                         const bool brick = false /* wave-uniform: the wave is spatially compact (polygon edge culling pays, poly_cull); with SHARE = 2 also: point kp has the same z in every lane; see xy_shared */,
                                         const float lip_h = 0.0f, const uint32_t lip_base = 0u,
                                         const uint32_t bmask = 0u /* wave-uniform: the brick mask of the cube this wave evaluates (dev_ops.h: D_SKIP); 0 where a wave is no brick */,
                                         uint32_t* lip_fired = nullptr /* interval mode: this lane's brick mask is OR-ed in here */,
                                         XYCache* xyc = nullptr /* column bricks: what the lane's other pass already knows (see XYCache) */) {
  using namespace dm;
  static_assert(!LIP || (K == 2 && SHARE == 0), "interval mode: two points per lane, lower and upper bound");
  [[maybe_unused]] float lipR = lip_h;
  KLOOP Rv[kp] = 0.0f;
  float hxy[K];  // hypot(P.x, P.y) cache shared by sibling primitives (validity is tracked by the host compiler)
  KLOOP hxy[kp] = 0.0f;
  uint32_t pc = 0;
#define PF(k) __uint_as_float(code[pc + 1 + (k)])
#define PU(k) (code[pc + 1 + (k)])
  for (;;) {
    // The program is straight-line and identical for every lane: pin the program counter to an SGPR so
    // that opcode/parameter fetches are scalar loads (s_load_dword*) and parameters are scalar operands.
    pc = __builtin_amdgcn_readfirstlane(pc);
    const uint32_t w = code[pc];
    const uint32_t op = w & D_OP_MASK;
    const uint32_t slot = w >> 16;
    const bool use_hxy = (w & D_FLAG_HXY) != 0u;   // wave-uniform
    const bool swap_ab = (w & D_FLAG_SWAP) != 0u;  // wave-uniform
    const bool sh_xy = SHARE != 0 && K >= 2 && (w & D_FLAG_SHXY) != 0u;
    const bool sh_z = SHARE == 1 && K >= 4 && (w & D_FLAG_SHZ) != 0u;
    const bool sh_col = SHARE == 2;  // only read together with sh_xy
    const bool sh_zu = SHARE == 2 && brick && (w & D_FLAG_SHZ) != 0u;  // column brick (leaf_eval_kernel): P.z of point kp is wave-uniform
    switch (op) {
      case D_END:
        return;
      // ------------------------------------------------ 3D primitives
      case D_SPHERE: {
        KLOOP {
          [[maybe_unused]] P3& p = pv[kp];
          [[maybe_unused]] float& R = Rv[kp];
          R = norm3(p.x, p.y, p.z) - PF(0);
        }
        pc += 2;
        break;
      }
      case D_BOX: {
        {
          const float r = PF(3);
          float mx[K], my[K], mz[K], in[K], hyz[K], n3[K];
          KLOOP {
            const P3& p = pv[kp];
            const float qx = (absf(p.x) - PF(0)) + r, qy = (absf(p.y) - PF(1)) + r, qz = (absf(p.z) - PF(2)) + r;
            mx[kp] = maxf(qx, 0.f); my[kp] = maxf(qy, 0.f); mz[kp] = maxf(qz, 0.f);
            in[kp] = minf(maxf(qx, maxf(qy, qz)), 0.0f);
          }
          hypot_k<K>(my, mz, hyz);   // ms3.Norm = hypot(x, hypot(y, z))
          hypot_k<K>(mx, hyz, n3);
          KLOOP Rv[kp] = n3[kp] + in[kp] - r;
        }
        pc += 5;
        break;
      }
      case D_BOXFRAME: {
        {
          // three "outside distance" norms of clamped components: ms3.Norm = hypot(x, hypot(y, z)) through hypot_k
          const float e = PF(0);
          float ax[K], ay[K], az[K], bx[K], by[K], bz[K], s1[K], s2[K], s3[K], t[K], n1[K], n2[K], n3[K];
          KLOOP {
            const P3& p = pv[kp];
            const float px = absf(p.x) - PF(1), py = absf(p.y) - PF(2), pz = absf(p.z) - PF(3);
            const float qx = absf(px + e) + (-e), qy = absf(py + e) + (-e), qz = absf(pz + e) + (-e);
            s1[kp] = minf(0.f, maxf(px, maxf(qy, qz)));
            s2[kp] = minf(0.f, maxf(qx, maxf(py, qz)));
            s3[kp] = minf(0.f, maxf(qx, maxf(qy, pz)));
            ax[kp] = maxf(px, 0.f); ay[kp] = maxf(py, 0.f); az[kp] = maxf(pz, 0.f);
            bx[kp] = maxf(qx, 0.f); by[kp] = maxf(qy, 0.f); bz[kp] = maxf(qz, 0.f);
          }
          hypot_k<K>(by, bz, t); hypot_k<K>(ax, t, n1);   // norm3(max(px,0), max(qy,0), max(qz,0))
          hypot_k<K>(ay, bz, t); hypot_k<K>(bx, t, n2);   // norm3(max(qx,0), max(py,0), max(qz,0))
          hypot_k<K>(by, az, t); hypot_k<K>(bx, t, n3);   // norm3(max(qx,0), max(qy,0), max(pz,0))
          KLOOP Rv[kp] = minf(n1[kp] + s1[kp], minf(n2[kp] + s2[kp], n3[kp] + s3[kp]));
        }
        pc += 5;
        break;
      }
      case D_TORUS: {
        ENSURE_HXY();
        KLOOP {
          [[maybe_unused]] P3& p = pv[kp];
          [[maybe_unused]] float& R = Rv[kp];
          float qx = hxy[kp] - PF(0);
          R = norm2(qx, p.z) - PF(1);
        }
        pc += 3;
        break;
      }
      case D_CYL0: {
        ENSURE_HXY();
        {
          float ax[K], ay[K], in[K], h[K];
          KLOOP {
            const float dx = hxy[kp] - PF(0), dy = absf(pv[kp].z) - PF(1);
            in[kp] = minf(0.f, maxf(dx, dy));
            ax[kp] = maxf(0.f, dx); ay[kp] = maxf(0.f, dy);
          }
          hypot_k<K>(ax, ay, h);
          KLOOP Rv[kp] = in[kp] + h[kp];
        }
        pc += 3;
        break;
      }
      case D_CYLR: {
        ENSURE_HXY();
        {
          const float round = PF(2);
          float ax[K], ay[K], in[K], h[K];
          KLOOP {
            const float dx = hxy[kp] - PF(0) + round, dy = absf(pv[kp].z) - PF(1);
            in[kp] = minf(maxf(dx, dy), 0.f);
            ax[kp] = maxf(dx, 0.f); ay[kp] = maxf(dy, 0.f);
          }
          hypot_k<K>(ax, ay, h);
          KLOOP Rv[kp] = in[kp] + h[kp] - round;
        }
        pc += 4;
        break;
      }
      case D_HEX: {
        float hax[K], hay[K], hh[K];
        KLOOP {
          [[maybe_unused]] P3& p = pv[kp];
          [[maybe_unused]] float& R = Rv[kp];
          const float k1 = -0.8660254037844386f, k2 = 0.5f, twok1 = -1.7320508075688772f;
          const float h1 = PF(0), h2 = PF(1), clm = PF(2);
          float px = absf(p.x), py = absf(p.y), pz = absf(p.z);
          float pm = minf(k1 * px + k2 * py, 0.f);
          px -= twok1 * pm;
          py -= 1.0f * pm;
          float d1 = hypotf_(px - clampf(px, -clm, clm), py - h1) * signf(py - h1);
          float d2 = pz - h2;
          hax[kp] = maxf(d1, 0.f); hay[kp] = maxf(d2, 0.f);
          R = minf(maxf(d1, d2), 0.f);
        }
        hypot_k<K>(hax, hay, hh);
        KLOOP Rv[kp] = Rv[kp] + hh[kp];
        pc += 4;
        break;
      }
      // ------------------------------------------------ 2D primitives
      case D_LINE2D: {
        KLOOP {
          [[maybe_unused]] P3& p = pv[kp];
          [[maybe_unused]] float& R = Rv[kp];
          const float bax = PF(2), bay = PF(3);
          float pax = p.x - PF(0), pay = p.y - PF(1);
          float h = clampf((pax * bax + pay * bay) / PF(4), 0.f, 1.f);
          R = norm2(pax - h * bax, pay - h * bay) - PF(5);
        }
        pc += 7;
        break;
      }
      case D_ARC2D: {
        KLOOP {
          [[maybe_unused]] P3& p = pv[kp];
          [[maybe_unused]] float& R = Rv[kp];
          const float r = PF(0), t = PF(1), s = PF(2), c = PF(3);
          float px = absf(p.x), py = p.y;
          float a = norm2(px - PF(4), py - PF(5)) - t;
          float b = absf(norm2(px, py) - r) - t;
          R = (c * px > s * py) ? a : b;
        }
        pc += 7;
        break;
      }
      case D_QUADBEZIER2D: {
        KLOOP {
          [[maybe_unused]] P3& p = pv[kp];
          [[maybe_unused]] float& R = Rv[kp];
          const float Ax = PF(0), Ay = PF(1), ax = PF(2), ay = PF(3), a2 = PF(4), bx = PF(5), by = PF(6), cx = PF(7),
                      cy = PF(8), kk = PF(9), kx = PF(10), kx2 = PF(11), thick = PF(12);
          float dx = Ax - p.x, dy = Ay - p.y;
          float ky = kk * (2.f * a2 + (dx * bx + dy * by)) / 3.f;
          float kz = kk * (dx * ax + dy * ay);
          float g = ky - kx2;
          float q = kx * (2.f * kx2 - 3.f * ky) + kz;
          float g3 = g * g * g;
          float q2 = q * q;
          float h = q2 + 4.f * g3;
          float res;
          if (h >= 0.f) {
            h = sqrtf_(h);
            float xx = 0.5f * (h + -q), xy = 0.5f * (-h + -q);
            if (absf(g) < 0.001f) {
              float k = (1.0f - g3 / q2) * g3 / q;
              xx = k;
              xy = -k - q;
            }
            float uvx = signf(xx) * pow13f_(absf(xx)), uvy = signf(xy) * pow13f_(absf(xy));
            float t = uvx + uvy;
            t -= (t * (t * t + 3.0f * g) + q) / (3.0f * t * t + 3.0f * g);
            t = clampf(t - kx, 0.f, 1.f);
            float wx = dx + t * (cx + t * bx), wy = dy + t * (cy + t * by);
            res = wx * wx + wy * wy;
          } else {
            float z = sqrtf_(-g);
            float v = q / (2.f * g * z);
            v = sqrtf_(0.5f + 0.5f * v);
            float m = v * (v * (v * (v * -0.008972f + 0.039071f) - 0.107074f) + 0.576975f) + 0.5f;
            float nn = sqrtf_(1.f - m * m);
            nn *= 1.7320508075688772f;
            float tx = clampf((m + m) * z - kx, 0.f, 1.f);
            float ty = clampf((-nn - m) * z - kx, 0.f, 1.f);
            float qxx = dx + tx * (cx + tx * bx), qxy = dy + tx * (cy + tx * by);
            float qyx = dx + ty * (cx + ty * bx), qyy = dy + ty * (cy + ty * by);
            float ddx = qxx * qxx + qxy * qxy, ddy = qyx * qyx + qyy * qyy;
            res = ddx < ddy ? ddx : ddy;
          }
          R = sqrtf_(res) - thick;
        }
        pc += 14;
        break;
      }
      case D_CIRCLE2D: {
        ENSURE_HXY();
        KLOOP {
          [[maybe_unused]] P3& p = pv[kp];
          [[maybe_unused]] float& R = Rv[kp];
          R = hxy[kp] - PF(0);
        }
        pc += 2;
        break;
      }
      case D_EQTRI2D: {
        KLOOP {
          [[maybe_unused]] P3& p = pv[kp];
          [[maybe_unused]] float& R = Rv[kp];
          const float k = 1.7320508075688772f;
          const float r = PF(0);
          float px = absf(p.x) - r, py = p.y + PF(1);
          if (px + k * py > 0.f) {
            float tx = px - k * py, ty = -k * px - py;
            px = 0.5f * tx;
            py = 0.5f * ty;
          }
          px -= clampf(px, -2.f * r, 0.f);
          R = -norm2(px, py) * signf(py);
        }
        pc += 3;
        break;
      }
      case D_RECT2D: {
        float rax[K], ray[K], rh[K];
        KLOOP {
          [[maybe_unused]] P3& p = pv[kp];
          [[maybe_unused]] float& R = Rv[kp];
          float dx = absf(p.x) - PF(0), dy = absf(p.y) - PF(1);
          rax[kp] = maxf(dx, 0.f); ray[kp] = maxf(dy, 0.f);
          R = minf(0.f, maxf(dx, dy));
        }
        hypot_k<K>(rax, ray, rh);
        KLOOP Rv[kp] = rh[kp] + Rv[kp];
        pc += 3;
        break;
      }
      case D_DIAMOND2D: {
        KLOOP {
          [[maybe_unused]] P3& p = pv[kp];
          [[maybe_unused]] float& R = Rv[kp];
          const float bx = PF(0), by = PF(1);
          float px = absf(p.x), py = absf(p.y);
          float tx = bx - 2.f * px, ty = by - 2.f * py;
          float h = clampf((tx * bx - ty * by) / PF(2), -1.f, 1.f);
          float d = norm2(px - PF(3) * (1.f - h), py - PF(4) * (1.f + h));
          R = d * signf(px * by + py * bx - PF(5));
        }
        pc += 7;
        break;
      }
      case D_X2D: {
        KLOOP {
          [[maybe_unused]] P3& p = pv[kp];
          [[maybe_unused]] float& R = Rv[kp];
          float px = absf(p.x), py = absf(p.y);
          float sub = 0.5f * minf(px + py, PF(0));
          R = norm2(px - sub, py - sub) - PF(1);
        }
        pc += 3;
        break;
      }
      case D_HEX2D: {
        KLOOP {
          [[maybe_unused]] P3& p = pv[kp];
          [[maybe_unused]] float& R = Rv[kp];
          const float kx = -0.8660254037844386f, ky = 0.5f;
          const float r = PF(0), kzr = PF(1);
          float px = absf(p.x), py = absf(p.y);
          float f = 2.f * minf(kx * px + ky * py, 0.f);
          px = px - f * kx;
          py = py - f * ky;
          px = px - clampf(px, -kzr, kzr);
          py = py - r;
          R = signf(py) * norm2(px, py);
        }
        pc += 3;
        break;
      }
      case D_OCT2D: {
        KLOOP {
          [[maybe_unused]] P3& p = pv[kp];
          [[maybe_unused]] float& R = Rv[kp];
          const float kx = -0.9238795325f, ky = 0.3826834323f;
          const float r = PF(0), kzr = PF(1);
          float px = absf(p.x), py = absf(p.y);
          float f1 = 2.f * minf(kx * px + ky * py, 0.f);
          px = px - f1 * kx;
          py = py - f1 * ky;
          float f2 = 2.f * minf((-kx) * px + ky * py, 0.f);
          px = px - f2 * (-kx);
          py = py - f2 * ky;
          px = px - clampf(px, -kzr, kzr);
          py = py - r;
          R = signf(py) * norm2(px, py);
        }
        pc += 3;
        break;
      }
      case D_ELLIPSE2D: {
        KLOOP {
          [[maybe_unused]] P3& p = pv[kp];
          [[maybe_unused]] float& R = Rv[kp];
          float a = PF(0), b = PF(1);
          float px = absf(p.x), py = absf(p.y);
          if (px > py) { float t = px; px = py; py = t; t = a; a = b; b = t; }
          float l = b * b - a * a;
          float m = a * px / l;
          float m2 = m * m;
          float nn = b * py / l;
          float n2 = nn * nn;
          float c = (m2 + n2 - 1.f) / 3.f;
          float c3 = c * c * c;
          float q = c3 + 2.f * m2 * n2;
          float d = c3 + m2 * n2;
          float g = m + m * n2;
          float co;
          if (d < 0.f) {
            float h = acosf_(q / c3) / 3.f;
            float sh, ch;
            sincosf_(h, sh, ch);
            float t = 1.7320508075688772f * sh;
            float rx = sqrtf_(-c * (ch + t + 2.f) + m2);
            float ry = sqrtf_(-c * (ch - t + 2.f) + m2);
            co = (ry + signf(l) * rx + absf(g) / (rx * ry) - m) / 2.f;
          } else {
            float h = 2.f * m * nn * sqrtf_(d);
            float sv = signf(q + h) * cbrtf_(absf(q + h));
            float u = signf(q - h) * cbrtf_(absf(q - h));
            float rx = -sv - u - 4.f * c + 2.f * m2;
            float ry = 1.7320508075688772f * (sv - u);
            float rm = hypotf_(rx, ry);
            co = (ry / sqrtf_(rm - rx) + 2.f * g / rm - m) / 2.f;
          }
          float rx2 = a * co, ry2 = b * sqrtf_(1.f - co * co);
          R = norm2(rx2 - px, ry2 - py) * signf(py - ry2);
        }
        pc += 3;
        break;
      }
      case D_POLY2D: {
        // See poly_edges below. Pass with the exact reciprocal division first; only if some lane's numerator left
        // the range where that form is proven exact, redo the polygon with the IEEE expansion (identical bits).
        const uint32_t hdr = PU(0);
        const uint32_t nv = hdr & 0x7fffffffu;
        const float v0x = PF(1), v0y = PF(2);
        const uint32_t q0 = (pc + 4u + 7u) & ~7u;
        if (LIP) {
          // interval mode: the lane's two points are the same centre (bit for bit: every position instruction treats them alike) --
          // the polygon once, the value into both columns (the primitives' widening by the radius follows below)
          const P3 p1[1] = {pv[0]};
          float d1[1];
          bool neg1[1];
          bool done1 = false;
          if (hdr >> 31) done1 = poly_edges<1, true>(code, q0, nv, v0x, v0y, p1, d1, neg1);
          if (!done1) poly_edges<1, false>(code, q0, nv, v0x, v0y, p1, d1, neg1);
          const float sd = sqrtf_(d1[0]);
          KLOOP Rv[kp] = neg1[0] ? -sd : sd;
        } else {
          float d[K];
          bool neg[K];
          bool done = false;
          uint64_t keepd = ~0ull, keeps = ~0ull;
          if (brick && nv >= 6u && nv <= 64u) poly_cull<K>(code, q0, nv, pv, keepd, keeps, (hdr >> 31) != 0u);  // a spatially compact wave: a 4x4x4-leaf brick, a patch of lattice cells, a run of kept cubes
          if (hdr >> 31) done = poly_edges<K, true>(code, q0, nv, v0x, v0y, pv, d, neg, keepd, keeps);
          if (!done) poly_edges<K, false>(code, q0, nv, v0x, v0y, pv, d, neg, keepd, keeps);
          float sd[K];
          sqrt_k<K>(d, sd);
          KLOOP Rv[kp] = neg[kp] ? -sd[kp] : sd[kp];  // s * sqrt(d), s = +-1
        }
        pc = q0 + 8u * nv;
        break;
      }
      case D_LINES2D: {
        const uint32_t ns = PU(0);
        const float w2 = PF(1);
        float d[K];
        KLOOP d[kp] = 1e23f;
        uint32_t q = pc + 3;
        for (uint32_t sg = 0; sg < ns; sg++, q += 5) {
          const float ax = __uint_as_float(code[q]), ay = __uint_as_float(code[q + 1]), bax = __uint_as_float(code[q + 2]),
                      bay = __uint_as_float(code[q + 3]), dotba = __uint_as_float(code[q + 4]);
          KLOOP {
            float pax = pv[kp].x - ax, pay = pv[kp].y - ay;
            float h = clampf((pax * bax + pay * bay) / dotba, 0.f, 1.f);
            float rx = pax - h * bax, ry = pay - h * bay;
            d[kp] = minf(d[kp], rx * rx + ry * ry);
          }
        }
        float sd[K];
        sqrt_k<K>(d, sd);
        KLOOP Rv[kp] = sd[kp] - w2;
        pc = q;
        break;
      }
      // ------------------------------------------------ position pre-ops
      case D_TRANSLATE: {
        KLOOP {
          [[maybe_unused]] P3& p = pv[kp];
          [[maybe_unused]] float& R = Rv[kp];
          p.x = p.x - PF(0); p.y = p.y - PF(1); p.z = p.z - PF(2);
        }
        pc += 4;
        break;
      }
      case D_SCALE_PRE: {
        KLOOP {
          [[maybe_unused]] P3& p = pv[kp];
          [[maybe_unused]] float& R = Rv[kp];
          const float f = PF(0);
          p.x = f * p.x; p.y = f * p.y; p.z = f * p.z;
        }
        if (LIP) lipR = lipR * absf(PF(0));  // (a negative factor mirrors: the radius scales by its magnitude)
        pc += 2;
        break;
      }
      case D_SYMMETRY: {
        KLOOP {
          [[maybe_unused]] P3& p = pv[kp];
          [[maybe_unused]] float& R = Rv[kp];
          const uint32_t bits = PU(0);
          if (bits & 1u) p.x = absf(p.x);
          if (bits & 2u) p.y = absf(p.y);
          if (bits & 4u) p.z = absf(p.z);
        }
        pc += 2;
        break;
      }
      case D_TRANSFORM: {
        // (A short form for matrices with exact zeros / ones -- the sum of the non-zero terms where every partial sum a dropped
        // +-0 would have joined is non-zero, wave vote, else the full form -- was measured on knurled-cylinder's z rotations:
        // 7 operations instead of 21 on paper, +2.5-4 % kernel time in practice (compares, vote and the second code path cost
        // more than twelve multiply-adds). Not kept.)
        KLOOP {
          [[maybe_unused]] P3& p = pv[kp];
          [[maybe_unused]] float& R = Rv[kp];
          float x = PF(0) * p.x + PF(1) * p.y + PF(2) * p.z + PF(3);
          float y = PF(4) * p.x + PF(5) * p.y + PF(6) * p.z + PF(7);
          float z = PF(8) * p.x + PF(9) * p.y + PF(10) * p.z + PF(11);
          p.x = x; p.y = y; p.z = z;
        }
        pc += 13;
        break;
      }
      case D_TWIST: {
        float tc[K], ts[K];  // cos/sin(k * P.z): a function of z only
        if (LIP) lipR = minf(lipR * lip_twist(hypotf_(pv[0].x, pv[0].y), lipR, absf(PF(0))), GSDF_LIP_BIG);
        {
          const float k = PF(0);
          if (sh_zu) {
            // column brick: P.z of point kp is the same in every lane, so lane l evaluates point l % K and everyone reads
            // lanes 0..K-1 -- one evaluation per lane instead of K (same function of the same input: same bits)
            const uint32_t sel = lane_id() & (uint32_t)(K - 1);
            float zs = pv[0].z;
            KLOOP if (sel == (uint32_t)kp) zs = pv[kp].z;
            float c1, s1;
            cossin_voted(k * zs, c1, s1);
            KLOOP {
              tc[kp] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c1), kp));
              ts[kp] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s1), kp));
            }
          } else {
            KLOOP if (kp < (K + 1) / 2) cossin_voted(k * pv[kp].z, tc[kp], ts[kp]);
            if (sh_z) { KLOOP if (kp >= (K + 1) / 2) { tc[kp] = tc[kp - K / 2]; ts[kp] = ts[kp - K / 2]; } }
            else { KLOOP if (kp >= (K + 1) / 2) cossin_voted(k * pv[kp].z, tc[kp], ts[kp]); }
          }
        }
        KLOOP {
          [[maybe_unused]] P3& p = pv[kp];
          [[maybe_unused]] float& R = Rv[kp];
          const float c = tc[kp], s = ts[kp];
          float x = c * p.x - s * p.y, y = s * p.x + c * p.y;
          p.x = x; p.y = y;
        }
        pc += 2;
        break;
      }
      case D_ROT2D: {
        KLOOP {
          [[maybe_unused]] P3& p = pv[kp];
          [[maybe_unused]] float& R = Rv[kp];
          float x = PF(0) * p.x + PF(1) * p.y, y = PF(2) * p.x + PF(3) * p.y;
          p.x = x; p.y = y;
        }
        pc += 5;
        break;
      }
      case D_EXTRUDE_PRE: {
        KLOOP {
          [[maybe_unused]] P3& p = pv[kp];
          [[maybe_unused]] float& R = Rv[kp];
          LDSF(slot) = absf(p.z) - PF(0);
        }
        if (LIP) LIP_WIDEN(lds[((slot) * K + LIP_LO) * nthreads], lds[((slot) * K + LIP_HI) * nthreads]);  // |z| - h/2 over the ball
        pc += 2;
        break;
      }
      case D_REVOLVE_PRE: {
        KLOOP {
          [[maybe_unused]] P3& p = pv[kp];
          [[maybe_unused]] float& R = Rv[kp];
          float x = hypotf_(p.x, p.z) - PF(0);
          p.x = x;  // p.y stays
        }
        pc += 2;
        break;
      }
      case D_SCREW_PRE: {
        float th[K];  // atan2(P.y, P.x): a function of x,y only
#ifdef GSDF_NO_XYCACHE  // (A/B knob of the specialised build: GSDF_HIP_SPEC_FLAGS=-DGSDF_NO_XYCACHE)
        xyc = nullptr;
#endif
        if (!LIP && xyc != nullptr && sh_xy && sh_col && xyc->pc == pc) {
          KLOOP th[kp] = xyc->th;  // the column's other pass has it
        } else {
          atan2_shared<K>(pv, th, sh_xy, brick, sh_col);
          if (!LIP && xyc != nullptr && sh_xy && sh_col) { xyc->th = th[0]; xyc->pc = pc; }
        }
        ENSURE_HXY();
        {
          // z' = z + lead*theta/2pi ; sawTooth(z', pitch): both divisors are wave-uniform -> exact reciprocal form
          const float pitch = PF(0), lead = PF(1), L = PF(2), tanTaper = PF(3), halfp = PF(4), rpitch = PF(5);
          const float twopi = 6.2831853071795862f, rtwopi = 0.15915493667125702f;  // RN(1/float32(2pi)) = 0x3e22f983
          float n1[K], q1[K], xs[K], t[K];
          KLOOP n1[kp] = lead * th[kp];
          div_uniform_k<K>(n1, twopi, rtwopi, q1);
          KLOOP xs[kp] = (pv[kp].z + q1[kp]) + halfp;
          div_uniform_k<K>(xs, pitch, rpitch, t);
          KLOOP {
            [[maybe_unused]] P3& p = pv[kp];
            float y0 = hxy[kp];
            y0 += p.z * tanTaper;
            float x0 = pitch * (t[kp] - floorf_(t[kp])) - halfp;
            LDSF(slot) = absf(p.z) - L;
            p.x = x0; p.y = y0;
          }
          if (LIP) {
            LIP_WIDEN(lds[((slot) * K + LIP_LO) * nthreads], lds[((slot) * K + LIP_HI) * nthreads]);  // |z| - L in the screw's own frame
            lipR = minf(lipR * lip_screw(hxy[0], lipR, absf(lead), tanTaper), GSDF_LIP_BIG);
          }
        }
        pc += 7;
        break;
      }
      case D_ELONGATE_PRE: {
        KLOOP {
          [[maybe_unused]] P3& p = pv[kp];
          [[maybe_unused]] float& R = Rv[kp];
          float qx = absf(p.x) - PF(0), qy = absf(p.y) - PF(1), qz = absf(p.z) - PF(2);
          LDSF(slot) = minf(maxf(qx, maxf(qy, qz)), 0.f);
          p.x = maxf(qx, 0.f); p.y = maxf(qy, 0.f); p.z = maxf(qz, 0.f);
        }
        if (LIP) LIP_WIDEN(lds[((slot) * K + LIP_LO) * nthreads], lds[((slot) * K + LIP_HI) * nthreads]);  // min(max3(q), 0) is 1-Lipschitz here
        pc += 4;
        break;
      }
      case D_ELONGATE2D_PRE: {
        KLOOP {
          [[maybe_unused]] P3& p = pv[kp];
          [[maybe_unused]] float& R = Rv[kp];
          float qx = absf(p.x) - PF(0), qy = absf(p.y) - PF(1);
          LDSF(slot) = minf(maxf(qx, qy), 0.f);
          p.x = maxf(qx, 0.f); p.y = maxf(qy, 0.f);
        }
        if (LIP) LIP_WIDEN(lds[((slot) * K + LIP_LO) * nthreads], lds[((slot) * K + LIP_HI) * nthreads]);
        pc += 3;
        break;
      }
      case D_ARRAY_PRE: {
        KLOOP {
          [[maybe_unused]] P3& p = pv[kp];
          [[maybe_unused]] float& R = Rv[kp];
          float ox = LDSF(slot), oy = LDSF(slot + 1), oz = LDSF(slot + 2);
          const float sx = PF(3), sy = PF(4), sz = PF(5);
          float idx = roundf_(ox / sx), idy = roundf_(oy / sy), idz = roundf_(oz / sz);
          float o_x = signf(ox - sx * idx), o_y = signf(oy - sy * idy), o_z = signf(oz - sz * idz);
          float rx = clampf(idx + PF(0) * o_x, 0.f, PF(6));
          float ry = clampf(idy + PF(1) * o_y, 0.f, PF(7));
          float rz = clampf(idz + PF(2) * o_z, 0.f, PF(8));
          p.x = ox - sx * rx; p.y = oy - sy * ry; p.z = oz - sz * rz;
        }
        pc += 10;
        break;
      }
      case D_ARRAY2D_PRE: {
        KLOOP {
          [[maybe_unused]] P3& p = pv[kp];
          [[maybe_unused]] float& R = Rv[kp];
          float ox = LDSF(slot), oy = LDSF(slot + 1);
          const float sx = PF(2), sy = PF(3);
          float idx = roundf_(ox / sx), idy = roundf_(oy / sy);
          float o_x = signf(ox - sx * idx), o_y = signf(oy - sy * idy);
          float rx = clampf(idx + PF(0) * o_x, 0.f, PF(4));
          float ry = clampf(idy + PF(1) * o_y, 0.f, PF(5));
          p.x = ox - sx * rx; p.y = oy - sy * ry;
        }
        pc += 7;
        break;
      }
      case D_CIRC_PRE: {
        // the sector index floor(atan2(P.y, P.x) / angle): from a float32 estimate of the angle where that decides it for
        // every point of the wave (dm::circ_sector_fast), else by the reference's expression (where the points share x,y
        // that one is computed once per pair / column anyway)
        float idv[K];
        bool fast = !sh_xy;
        if (fast) {
          const float inv_angle = __builtin_amdgcn_rcpf(PF(0)), m = 6e-6f * inv_angle;
          KLOOP fast = dm::circ_sector_fast(pv[kp].x, pv[kp].y, inv_angle, m, idv[kp]) && fast;
          fast = __builtin_amdgcn_ballot_w64(!fast) == 0ull;  // wave-uniform
        }
        if (!fast) {
          float th[K];  // atan2(P.y, P.x): a function of x,y only (a rare path -- a point on a sector boundary: the reference's sequence)
          xy_shared<K>(pv, th, sh_xy, brick, [](float x, float y) { return dm::atan2_ref(y, x); }, sh_col);
          KLOOP idv[kp] = floorf_(th[kp] / PF(0));
        }
        KLOOP {
          [[maybe_unused]] P3& p = pv[kp];
          [[maybe_unused]] float& R = Rv[kp];
          const float angle = PF(0), ncirc = PF(1), ninsm1 = PF(2);
          float id = idv[kp];
          if (id < 0.f) id += ncirc;
          float i0, i1;
          if (id >= ninsm1) { i0 = ninsm1; i1 = 0.f; } else { i0 = id; i1 = id + 1.f; }
          float s0, c0, s1, c1;
          const uint32_t tab = PU(3);
          if (tab) {  // i0, i1 are integers in [0, ncirc): {sin, cos}(angle * i) from the host's table (per-lane loads)
            const uint32_t a0 = tab + 2u * (uint32_t)(int)i0, a1 = tab + 2u * (uint32_t)(int)i1;
            s0 = __uint_as_float(code[a0]); c0 = __uint_as_float(code[a0 + 1]);
            s1 = __uint_as_float(code[a1]); c1 = __uint_as_float(code[a1 + 1]);
          } else {
            sincosf_(angle * i0, s0, c0);
            sincosf_(angle * i1, s1, c1);
          }
          LDSF(slot) = c0 * p.x + s0 * p.y;
          LDSF(slot + 1) = (-s0) * p.x + c0 * p.y;
          float x1 = c1 * p.x + s1 * p.y, y1 = (-s1) * p.x + c1 * p.y;
          p.x = x1; p.y = y1;
        }
        pc += 5;
        break;
      }
      case D_LOADP2_SUB: {
        KLOOP {
          [[maybe_unused]] P3& p = pv[kp];
          [[maybe_unused]] float& R = Rv[kp];
          p.x = LDSF(slot) - PF(0);
          p.y = LDSF(slot + 1) - PF(1);
        }
        pc += 3;
        break;
      }
      // ------------------------------------------------ distance post-ops
      case D_MULR: {
        if (LIP) {  // the scale nodes' d * f on an interval (f of either sign); the ball's image is back in the outer frame
          lipR = lipR * absf(PF(0));
          const float x0 = Rv[LIP_LO] * PF(0), x1 = Rv[LIP_HI] * PF(0);
          Rv[LIP_LO] = minf(x0, x1); Rv[LIP_HI] = maxf(x0, x1);
        } else {
          KLOOP { P3& p = pv[kp]; float& R = Rv[kp]; (void)p; (void)R; R = R * PF(0); }
        }
        pc += 2;
        break;
      }
      case D_SHELL_POST: {
        const float th = PF(0);
        if (LIP) {
          lipR = lipR * absf(th);
          float alo, ahi;
          lip_abs(Rv[LIP_LO], Rv[LIP_HI], alo, ahi);
          const float x0 = th * (alo - th), x1 = th * (ahi - th);
          Rv[LIP_LO] = minf(x0, x1); Rv[LIP_HI] = maxf(x0, x1);
        } else {
          KLOOP { float& R = Rv[kp]; R = th * (absf(R) - th); }
        }
        pc += 2;
        break;
      }
      case D_ADDR: { KLOOP { P3& p = pv[kp]; float& R = Rv[kp]; (void)p; (void)R; R = R + PF(0); } pc += 2; break; }
      case D_ANNULUS: {
        if (LIP) {
          float alo, ahi;
          lip_abs(Rv[LIP_LO], Rv[LIP_HI], alo, ahi);
          Rv[LIP_LO] = alo - PF(0); Rv[LIP_HI] = ahi - PF(0);
        } else {
          KLOOP { P3& p = pv[kp]; float& R = Rv[kp]; (void)p; (void)R; R = absf(R) - PF(0); }
        }
        pc += 2;
        break;
      }
      case D_EXTRUDE_POST: {
        float eax[K], eay[K], eh[K];
        KLOOP {
          [[maybe_unused]] P3& p = pv[kp];
          [[maybe_unused]] float& R = Rv[kp];
          float wy = LDSF(slot);
          eax[kp] = maxf(R, 0.f); eay[kp] = maxf(wy, 0.f);
          R = minf(0.f, maxf(R, wy));
        }
        hypot_k<K>(eax, eay, eh);
        KLOOP Rv[kp] = Rv[kp] + eh[kp];
        pc += 1;
        break;
      }
      case D_MAXR_SLOT: { KLOOP { P3& p = pv[kp]; float& R = Rv[kp]; (void)p; (void)R; R = maxf(R, LDSF(slot)); } pc += 1; break; }
      case D_ADDR_SLOT: { KLOOP { P3& p = pv[kp]; float& R = Rv[kp]; (void)p; (void)R; R = R + LDSF(slot); } pc += 1; break; }
      // ------------------------------------------------ scratch
      case D_SAVEP3: { KLOOP { P3& p = pv[kp]; float& R = Rv[kp]; (void)p; (void)R; LDSF(slot) = p.x; LDSF(slot + 1) = p.y; LDSF(slot + 2) = p.z; } pc += 1; break; }
      case D_LOADP3: { KLOOP { P3& p = pv[kp]; float& R = Rv[kp]; (void)p; (void)R; p.x = LDSF(slot); p.y = LDSF(slot + 1); p.z = LDSF(slot + 2); } pc += 1; break; }
      case D_SAVEP2: { KLOOP { P3& p = pv[kp]; float& R = Rv[kp]; (void)p; (void)R; LDSF(slot) = p.x; LDSF(slot + 1) = p.y; } pc += 1; break; }
      case D_LOADP2: { KLOOP { P3& p = pv[kp]; float& R = Rv[kp]; (void)p; (void)R; p.x = LDSF(slot); p.y = LDSF(slot + 1); } pc += 1; break; }
      case D_SAVER: { KLOOP { P3& p = pv[kp]; float& R = Rv[kp]; (void)p; (void)R; LDSF(slot) = R; } pc += 1; break; }
      case D_SETSLOT: { KLOOP { P3& p = pv[kp]; float& R = Rv[kp]; (void)p; (void)R; LDSF(slot) = PF(0); } pc += 2; break; }
      case D_SETR: { KLOOP { P3& p = pv[kp]; float& R = Rv[kp]; (void)p; (void)R; R = PF(0); } pc += 2; break; }
      // ------------------------------------------------ combine (a = saved first operand, b = R)
      case D_COMBINE_MIN: { KLOOP { P3& p = pv[kp]; float& R = Rv[kp]; (void)p; (void)R; R = minf(LDSF(slot), R); } pc += 1; break; }
      case D_COMBINE_MAX: { KLOOP { P3& p = pv[kp]; float& R = Rv[kp]; (void)p; (void)R; R = maxf(LDSF(slot), R); } pc += 1; break; }
      case D_COMBINE_DIFF: {
        float av[K], bv[K];
        KLOOP { float a = LDSF(slot), b = Rv[kp]; if (swap_ab) { float t = a; a = b; b = t; } av[kp] = a; bv[kp] = b; }
        if (LIP) { const float t = bv[LIP_LO]; bv[LIP_LO] = bv[LIP_HI]; bv[LIP_HI] = t; }  // decreasing in b: its bounds change places
        KLOOP Rv[kp] = maxf(av[kp], -bv[kp]);
        pc += 1;
        break;
      }
      case D_COMBINE_XOR: {
        if (LIP) {
          const float alo = lds[((slot) * K + LIP_LO) * nthreads], ahi = lds[((slot) * K + LIP_HI) * nthreads], blo = Rv[LIP_LO], bhi = Rv[LIP_HI];
          Rv[LIP_LO] = maxf(minf(alo, blo), -maxf(ahi, bhi));
          Rv[LIP_HI] = maxf(minf(ahi, bhi), -maxf(alo, blo));
        } else {
          KLOOP { float& R = Rv[kp]; float a = LDSF(slot), b = R; R = maxf(minf(a, b), -maxf(a, b)); }
        }
        pc += 1;
        break;
      }
      case D_COMBINE_SUNION: {
        {
          const float k = PF(0), rk = PF(1);  // (0.5*(b -+ a)) / k by a wave-uniform k: exact reciprocal form
          float av[K], bv[K], num[K], hv[K];
          KLOOP {
            float a = LDSF(slot), b = Rv[kp];
            if (swap_ab) { float t = a; a = b; b = t; }
            av[kp] = a; bv[kp] = b;
            num[kp] = 0.5f * (b - a);
          }
          smooth_h<K, 1>(num, k, rk, hv);
          KLOOP {
            float& R = Rv[kp];
            const float a = av[kp], b = bv[kp];
            const float h = hv[kp];
            R = mixf(b, a, h) - k * h * (1.f - h);
          }
        }
        pc += 3;
        break;
      }
      case D_COMBINE_SDIFF: {
        {
          const float k = PF(0), rk = PF(1);  // (0.5*(b -+ a)) / k by a wave-uniform k: exact reciprocal form
          float av[K], bv[K], num[K], hv[K];
          KLOOP {
            float a = LDSF(slot), b = Rv[kp];
            if (swap_ab) { float t = a; a = b; b = t; }
            av[kp] = a; bv[kp] = b;
          }
          if (LIP) { const float t = bv[LIP_LO]; bv[LIP_LO] = bv[LIP_HI]; bv[LIP_HI] = t; }  // decreasing in b
          KLOOP num[kp] = 0.5f * (bv[kp] + av[kp]);
          smooth_h<K, -1>(num, k, rk, hv);
          KLOOP {
            float& R = Rv[kp];
            const float a = av[kp], b = bv[kp];
            const float h = hv[kp];
            R = mixf(a, -b, h) + k * h * (1.f - h);
          }
        }
        pc += 3;
        break;
      }
      case D_COMBINE_SINTER: {
        {
          const float k = PF(0), rk = PF(1);  // (0.5*(b -+ a)) / k by a wave-uniform k: exact reciprocal form
          float av[K], bv[K], num[K], hv[K];
          KLOOP {
            float a = LDSF(slot), b = Rv[kp];
            if (swap_ab) { float t = a; a = b; b = t; }
            av[kp] = a; bv[kp] = b;
            num[kp] = 0.5f * (b - a);
          }
          smooth_h<K, -1>(num, k, rk, hv);
          KLOOP {
            float& R = Rv[kp];
            const float a = av[kp], b = bv[kp];
            const float h = hv[kp];
            R = mixf(b, a, h) + k * h * (1.f - h);
          }
        }
        pc += 3;
        break;
      }
      case D_UBOUND2D: {
        // running minimum of a wide union <- upper bound: (1 + 1e-3) * min over boxes of the distance to the far corner
        const uint32_t nb = PU(0);
        float m2[K];
        KLOOP m2[kp] = 3.0e38f;
        uint32_t q = pc + 2;
        for (uint32_t ib = 0; ib < nb; ib++, q += 4) {
          const float x0 = __uint_as_float(code[q]), y0 = __uint_as_float(code[q + 1]), x1 = __uint_as_float(code[q + 2]),
                      y1 = __uint_as_float(code[q + 3]);
          KLOOP {
            const float dx = maxf(absf(pv[kp].x - x0), absf(pv[kp].x - x1)), dy = maxf(absf(pv[kp].y - y0), absf(pv[kp].y - y1));
            m2[kp] = minf(m2[kp], dx * dx + dy * dy);
          }
        }
        if (LIP) {
          // interval mode: the bound over the whole ball (the distance to a box's farthest corner is 1-Lipschitz in the point, so is
          // the minimum over the boxes), above every value the per-point form can take there -- 1.002 against its 1.001 and the
          // rounding of its sums, which is below 1e-6 of the coordinates' magnitudes. BOTH columns: the running minimum of the
          // union never exceeds it anywhere in the ball, and the children the gates now skip from the first one on (each proven
          // above it over the whole ball, D_GATE*) cannot lower the interval's lower end below what the evaluated ones give.
          const float u = 1.002f * (sqrtf_(m2[0]) + lipR) + 2e-6f * (absf(pv[0].x) + absf(pv[0].y)) + 1e-30f;
          KLOOP LDSF(slot) = u;
        } else {
          KLOOP LDSF(slot) = 1.001f * sqrtf_(m2[kp]) + 1e-30f;
        }
        pc = q;
        break;
      }
      case D_UBOUND3D: {
        const uint32_t nb = PU(0);
        float m2[K];
        KLOOP m2[kp] = 3.0e38f;
        uint32_t q = pc + 2;
        for (uint32_t ib = 0; ib < nb; ib++, q += 6) {
          const float x0 = __uint_as_float(code[q]), y0 = __uint_as_float(code[q + 1]), z0 = __uint_as_float(code[q + 2]),
                      x1 = __uint_as_float(code[q + 3]), y1 = __uint_as_float(code[q + 4]), z1 = __uint_as_float(code[q + 5]);
          KLOOP {
            const float dx = maxf(absf(pv[kp].x - x0), absf(pv[kp].x - x1)), dy = maxf(absf(pv[kp].y - y0), absf(pv[kp].y - y1)),
                        dz = maxf(absf(pv[kp].z - z0), absf(pv[kp].z - z1));
            m2[kp] = minf(m2[kp], dx * dx + dy * dy + dz * dz);
          }
        }
        if (LIP) {  // (see D_UBOUND2D)
          const float u = 1.002f * (sqrtf_(m2[0]) + lipR) + 2e-6f * (absf(pv[0].x) + absf(pv[0].y) + absf(pv[0].z)) + 1e-30f;
          KLOOP LDSF(slot) = u;
        } else {
          KLOOP LDSF(slot) = 1.001f * sqrtf_(m2[kp]) + 1e-30f;
        }
        pc = q;
        break;
      }
      case D_GATE2D: {
        GSDF_GATE_TEST((region_lb_box<K, 2>(pv, PF(0), PF(1), 0.f, PF(2), PF(3), 0.f, L)), 2, 4)
        if (far_) { GSDF_GATE_TAKEN pc += GSDF_GATE_SKIP(PU(9)); } else { pc += 11; }
        break;
      }
      case D_GATE3D: {
        GSDF_GATE_TEST((region_lb_box<K, 3>(pv, PF(0), PF(1), PF(2), PF(3), PF(4), PF(5), L)), 3, 6)
        if (far_) { GSDF_GATE_TAKEN pc += GSDF_GATE_SKIP(PU(11)); } else { pc += 13; }
        break;
      }
      case D_GATEZC: {
        GSDF_GATE_TEST((region_lb_zcyl<K>(pv, hxy, use_hxy, PF(0), PF(1), PF(2), PF(3), PF(4), PF(5), PF(6), L)), 3, 7)
        if (far_) { GSDF_GATE_TAKEN pc += GSDF_GATE_SKIP(PU(12)); } else { pc += 14; }
        break;
      }
      case D_GATEOB: {
        GSDF_GATE_TEST((region_lb_obox<K, 3>(pv, PF(0), PF(1), PF(2), PF(3), PF(4), PF(5), PF(6), PF(7), L)), 3, 8)
        if (far_) { GSDF_GATE_TAKEN pc += GSDF_GATE_SKIP(PU(13)); } else { pc += 15; }
        break;
      }
      case D_CIRC_ORDER: {
        // which of the circular array's two copies lies nearer the child's region, for most lanes of the wave: that one first
        // (min is commutative bit for bit), so that the gate in front of the second has the best chance (dev_ops.h). Both
        // copies are equally far from the axis, so the nearer one to the region's centre is the one with the larger
        // projection on it; one point per lane votes (a heuristic: either order gives the same bits).
        if (!LIP) {
          const float d1 = pv[0].x * PF(0) + pv[0].y * PF(1);
          const float d0 = lds[((slot) * K) * nthreads] * PF(0) + lds[((slot + 1) * K) * nthreads] * PF(1);
          const uint64_t v0 = __builtin_amdgcn_ballot_w64(d0 > d1), v1 = __builtin_amdgcn_ballot_w64(d1 > d0);
          if (__builtin_popcountll(v0) > __builtin_popcountll(v1)) {  // wave-uniform
            KLOOP {
              const float tx = LDSF(slot), ty = LDSF(slot + 1);
              LDSF(slot) = pv[kp].x; LDSF(slot + 1) = pv[kp].y;
              pv[kp].x = tx; pv[kp].y = ty;
            }
          }
        }
        pc += 7;
        break;
      }
      // ------------------------------------------------ interval mode bookkeeping (no-ops elsewhere)
      case D_LIP_PUSH: {
        if (LIP) {  // second column: the magnitude of the position here (D_LIP_DOM pads its comparisons by it)
          lds[((lip_base + slot) * K) * nthreads] = lipR;
          lds[((lip_base + slot) * K + LIP_HI) * nthreads] = absf(pv[0].x) + absf(pv[0].y) + absf(pv[0].z);
        }
        pc += 1;
        break;
      }
      case D_LIP_POP: { if (LIP) lipR = lds[((lip_base + slot) * K) * nthreads]; pc += 1; break; }
      case D_LIP_MUL: { if (LIP) lipR = minf(lipR * PF(0), GSDF_LIP_BIG); pc += 2; break; }
      case D_LIP_WRAP: { if (LIP) { if (absf(pv[0].x) + lipR >= PF(0)) lipR = lipR + PF(1); } pc += 3; break; }
      // ------------------------------------------------ brick masks (dev_ops.h: D_SKIP, D_LIP_DOM). Not cases of their own: two
      // more entries in the dispatch tree cost the ahead-of-time kernels that sit exactly at their register budget four spilled
      // VGPRs (flat_grid_kernel<4, 4>, dc_origin_kernel<2, 4>: tests/test_kernel_resources.py), and a kernel with scratch is not used.
      default:
        if (op == D_SKIP) {
        const uint32_t sw = PU(2);
        if (!LIP && ((bmask >> (PU(0) & 31u)) & 1u) != 0u) {  // wave-uniform: the centre test proved this operand irrelevant for the whole brick
          if (use_hxy) xy_shared<K>(pv, hxy, sh_xy, brick, [](float x, float y) { return dm::hypotf_(x, y); }, sh_col);  // (flag on D_SKIP: code behind the subtree reuses the hypot the subtree would have left)
          KLOOP Rv[kp] = PF(1);
          pc += GSDF_GATE_SKIP(sw);
        } else {
          pc += 4;
        }
          break;
        }
        if (op == D_LIP_DOM) {
          GSDF_LIP_DOM_BODY
          pc += 6;
          break;
        }
        KLOOP Rv[kp] = __builtin_nanf("");
        return;
    }
    if (LIP && op >= D_SPHERE && op <= D_LINES2D) LIP_WIDEN(Rv[LIP_LO], Rv[LIP_HI]);  // a primitive: exact distance -+ radius
  }
#undef PF
#undef PU
}
#else
// Run-time specialisation (hiprtc): sdf_eval<K, SHARE> for ONE lowered program, generated by specialize.cpp from the
// case bodies above -- the same statements in program order, with the instruction word, its flags, the slot number and
// every parameter as literals, so there is no fetch/decode, no dispatch branch and no dead flag test left.
}  // namespace gsdf_dev
#include "gsdf_spec_gen.h"
namespace gsdf_dev {
#endif

#undef ENSURE_HXY
#undef LDSF
#undef KLOOP
#undef LIP_WIDEN
#undef LIP_LO
#undef LIP_HI
#undef GSDF_GATE_TEST
#undef GSDF_GATE_TAKEN
#undef GSDF_LIP_DOM_BODY

}  // namespace gsdf_dev
