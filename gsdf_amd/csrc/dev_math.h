// dev_math.h -- gfx950 device math for the SDF interpreter.
//
// float32 semantics of the reference's CPU evaluators (Go on amd64: IEEE single ops, never fused):
// every expression below is a sequence of correctly rounded +,-,*,/,sqrt; compile with
// -ffp-contract=off and without fast-math so hipcc keeps v_mul/v_add pairs (no v_fma contraction)
// and its correctly rounded fp32 divide/sqrt expansions. Transcendentals follow the algorithms the
// reference's math package uses (chewxy/math32 v1.11.1: float32(math.X(float64)) wrappers over Go's
// Cephes-derived float64 routines for Atan2/Sin/Cos/Acos/Cbrt; float32 ports for Hypot/Sincos), so
// distances are bit-identical to the CPU path for finite inputs. Not reproduced: Inf/NaN special
// cases of hypot/atan2 (unreachable with finite positions) and math.Min/Max NaN propagation.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#else  // hiprtc: no host headers; the fixed-width types are all the device code needs from them
typedef signed char int8_t;
typedef unsigned char uint8_t;
typedef unsigned short uint16_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long long int64_t;
typedef unsigned long long uint64_t;
typedef unsigned long uintptr_t;
#endif

namespace dm {

#define DM_INL __device__ __forceinline__

DM_INL float absf(float x) { return __builtin_fabsf(x); }
DM_INL float minf(float a, float b) { return __builtin_fminf(a, b); }  // v_min_f32: -0 < +0 like math.Min
DM_INL float maxf(float a, float b) { return __builtin_fmaxf(a, b); }
DM_INL float sqrtf_(float x) { return __builtin_sqrtf(x); }
DM_INL float floorf_(float x) { return __builtin_floorf(x); }
DM_INL float roundf_(float x) { return __builtin_roundf(x); }  // half away from zero == Go math.Round
DM_INL float copysignf_(float x, float s) { return __builtin_copysignf(x, s); }
DM_INL float signf(float a) { return a == 0.0f ? 0.0f : copysignf_(1.0f, a); }
DM_INL float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
DM_INL float mixf(float x, float y, float a) { return x * (1.0f - a) + y * a; }

// Correctly rounded n/d for a WAVE-UNIFORM divisor d whose correctly rounded reciprocal r = RN(1/d) was
// computed on the host (compile.cpp: recip_for): q0 = RN(n*r), two Markstein corrections
// q <- RN(q + RN(n - d*q)*r) with the residual exact by FMA. After the first correction q is a faithful
// rounding of n/d, so the second yields RN(n/d) (Markstein 1990; Muller et al., Handbook of FP Arithmetic,
// ch. 4: y = RN(1/b), q faithful, r = a - b*q exact  =>  RN(q + r*y) = RN(a/b)), provided nothing
// under/overflows: callers guard |n| in [2^-90, 2^90] (divisors are restricted to [2^-30, 2^30] by the
// host) and fall back to the IEEE expansion for the whole wave otherwise (also for n == 0, whose zero
// sign the FMA chain does not preserve). 5 full-rate VALU ops instead of ~10 + a quarter-rate v_rcp.
// Exhaustively checked against v_div for every float32 numerator: tests/test_gpu_eval.py (div selftest).
DM_INL float div_by_uniform(float n, float d, float r) {
  float q = n * r;
  float e = __builtin_fmaf(-d, q, n);
  q = __builtin_fmaf(e, r, q);
  e = __builtin_fmaf(-d, q, n);
  q = __builtin_fmaf(e, r, q);
  return q;
}
// K numerators of one lane by a wave-uniform divisor: the exact reciprocal form when the host supplied RN(1/d)
// (r != 0) and every numerator of the wave is in the proven range, else the IEEE expansion for the whole wave
// (wave-uniform branch; identical bits either way).
template <int K>
DM_INL void div_uniform_k(const float (&n)[K], float d, float r, float (&q)[K]) {
  bool fast = r != 0.0f;
  if (fast) {
    float nmin = 1.0f, nmax = 1.0f;
#pragma unroll
    for (int k = 0; k < K; k++) {
      q[k] = div_by_uniform(n[k], d, r);
      nmin = minf(nmin, absf(n[k]));
      nmax = maxf(nmax, absf(n[k]));
    }
    fast = __all(nmin >= 8.0779357e-28f /* 2^-90 */ && nmax <= 1.2379400e+27f /* 2^90 */) != 0;
  }
  if (!fast) {
#pragma unroll
    for (int k = 0; k < K; k++) q[k] = n[k] / d;
  }
}
DM_INL bool div_fast_ok(float n) {
  const float a = absf(n);
  return a >= 8.0779357e-28f /* 2^-90 */ && a <= 1.2379400e+27f /* 2^90 */;
}

// Correctly rounded sqrt for s in [1, 2] (hypot's 1 + q*q with q in [0,1]): hipcc's generic expansion of sqrtf
// spends 7 of its 16 instructions on denormal pre-scaling and zero/inf class handling that cannot apply here. What
// remains is its own fix-up: v_sqrt_f32 is within 1 ulp; compare the residuals of the neighbours r-1ulp / r+1ulp
// (exact by FMA) and step if needed. Checked against sqrtf for every float in [1, 2] on the GPU
// (test_sqrt_unit_range_exhaustive).
DM_INL float sqrt_1to2(float s) {
  // r in [1, sqrt 2]: ulp(r) = 2^-23 throughout. With e = s - r*r (FMA; exact wherever the decision is close, see
  // below) the correctly rounded root is r+1ulp iff s > (r + 2^-24)^2 and r-1ulp iff s < (r - 2^-24)^2. e, r*2^-23
  // and s are multiples of 2^-46, so the 2^-48 term of the squares cannot decide: the tests reduce to
  // e > r*2^-23 and e <= -r*2^-23. |e| near r*2^-23 < 2^-22 has at most 24 significant bits (exact); larger |e|
  // is rounded but stays on its side of the threshold. One residual instead of two neighbour products, and the
  // +-1 ulp steps are carry-in adds on the bit pattern.
  const float r = __builtin_amdgcn_sqrtf(s);
  const float t = r * 1.1920928955078125e-07f;  // r * 2^-23
  const float e = __builtin_fmaf(-r, r, s);
  uint32_t u = __float_as_uint(r);
  u += (e > t) ? 1u : 0u;
  u -= (e <= -t) ? 1u : 0u;
  return __uint_as_float(u);
}

// Correctly rounded sqrt of the K values of a lane where every value of the WAVE is at least 2^-96 (or NaN / +Inf): the compiler's
// expansion of sqrtf without its range handling. That expansion is v_sqrt_f32 (1 ulp), the two neighbours r -+ 1 ulp, their residuals
// s - r' r by FMA and two selects -- wrapped in a pre-scaling by 2^32 for arguments below 2^-96 (a multiply, a compare, a select; a
// multiply and a select behind) and a class test that hands zeros and infinities through (a compare, a select): seven of its sixteen
// instructions, all idle for a squared distance that is neither zero nor vanishing. Here: the wave's smallest argument decides
// (two v_min3 for four values, a compare, a vote) between the nine-instruction core for everyone and the compiler's expansion for
// everyone. The core is the expansion's own instruction sequence, so the bits are the same by construction wherever it runs: an
// argument >= 2^-96 is not pre-scaled (the threshold is the expansion's own, 0x0f800000), and +Inf / NaN come out of the core as they
// come out of v_sqrt_f32 (the neighbours' residuals are NaN, both selects keep r). Zero takes the compiler's path with its wave.
// Per-tree kernels only, like the other short routes (interp.h: atan2_shared).
DM_INL float sqrt_core(float s) {
  const float r = __builtin_amdgcn_sqrtf(s);
  const uint32_t u = __float_as_uint(r);
  const float rm = __uint_as_float(u - 1u), rp = __uint_as_float(u + 1u);
  const float em = __builtin_fmaf(-rm, r, s), ep = __builtin_fmaf(-rp, r, s);
  float o = (em <= 0.0f) ? rm : r;
  o = (ep > 0.0f) ? rp : o;
  return o;
}
template <int K>
DM_INL void sqrt_k(const float (&s)[K], float (&r)[K]) {
#if defined(GSDF_SPECIALIZED) && !defined(GSDF_NO_SQRT_CORE)
  float m = s[0];
  if (K == 4) m = minf(minf(minf(s[0], s[K > 1 ? 1 : 0]), s[K > 2 ? 2 : 0]), s[K > 3 ? 3 : 0]);
  else {
#pragma unroll
    for (int k = 1; k < K; k++) m = minf(m, s[k]);
  }
  // (v_min_f32 drops a NaN operand: a NaN argument beside ordinary ones takes the core, which returns what v_sqrt_f32 returns for it,
  // as the expansion does; all NaN: the compare fails, the compiler's path)
  if (__all(m >= 1.2621774483536189e-29f /* 2^-96 */)) {
#pragma unroll
    for (int k = 0; k < K; k++) r[k] = sqrt_core(s[k]);
    return;
  }
#endif
#pragma unroll
  for (int k = 0; k < K; k++) r[k] = sqrtf_(s[k]);
}

// math32.Hypot (float32 port of go/src/math/hypot.go)
DM_INL float hypotf_(float p, float q) {
  p = absf(p);
  q = absf(q);
  // "if p < q swap" as max/min (identical for non-NaN magnitudes); "if p == 0 return 0" by dividing by at least the
  // smallest subnormal: hi > 0 is unchanged by the max, hi == 0 gives 0/tiny = 0 and 0 * sqrt(1) = 0.
  const float hi = maxf(p, q), lo = minf(p, q);
  const float r = lo / maxf(hi, 1.401298464324817e-45f);
  return hi * sqrt_1to2(1.0f + r * r);  // r in [0,1]
}
// ---- interval mode of the octree's centre tests (dev_ops.h: D_LIP_*). Stretch factors of the two position maps of the
// reference that are not 1-Lipschitz; rho = distance of the cube centre from the node's z axis, rl = radius of the ball
// that holds the cube's image there. The same float32 sequences as the oracle's (orc_eval.c: lip_twist / lip_screw).
//   twist  (x,y) -> R(k z)(x,y) (cpu_evaluators.go:1257-1274): in the (radial, tangential, axial) frame the Jacobian is a
//          shear by s = |k| rho in the (tangential, axial) plane, spectral norm (s + sqrt(s^2 + 4)) / 2, largest at the
//          largest rho;
//   screw  (threads.go:141-181) x' = saw(z + lead theta / 2pi), y' = rho + z tanT: rows (0, a, 1) and (1, 0, t) in that frame,
//          a = |lead| / (2 pi rho); J J^T = [[1 + a^2, t], [t, 1 + t^2]], largest eigenvalue ((2 + a^2 + t^2) + sqrt((a^2 -
//          t^2)^2 + 4 t^2)) / 2, largest at the smallest rho; no bound on the axis (LIP_BIG, finite so that nothing
//          downstream sees Inf - Inf).
#define GSDF_LIP_BIG 1e18f
DM_INL float lip_twist(float rho, float rl, float ak) {
  const float s = ak * (rho + rl);
  return 0.5f * (s + sqrtf_(s * s + 4.0f));
}
DM_INL float lip_screw(float rho, float rl, float alead, float t) {
  const float rmin = rho - rl;
  if (!(rmin > 0.0f)) return GSDF_LIP_BIG;
  const float a = alead / (6.2831855f * rmin);
  const float a2 = a * a, t2 = t * t, dd = a2 - t2;
  return sqrtf_(0.5f * ((2.0f + a2 + t2) + sqrtf_(dd * dd + 4.0f * t2)));
}
// |x| of the interval [lo, hi]
DM_INL void lip_abs(float lo, float hi, float& alo, float& ahi) {
  alo = maxf(maxf(lo, -hi), 0.0f);
  ahi = maxf(-lo, hi);
}

DM_INL float norm3(float x, float y, float z) { return hypotf_(x, hypotf_(y, z)); }  // ms3.Norm
DM_INL float norm2(float x, float y) { return hypotf_(x, y); }                       // ms2.Norm

// math32.Sincos (float32 port of go/src/math/sincos.go)
DM_INL void sincosf_(float x, float& s_out, float& c_out) {
  const float PI4A = 7.85398125648498535156e-1f, PI4B = 3.77489470793079817668e-8f, PI4C = 2.69515142907905952645e-15f;
  const float M4PI = 1.2732395447351628f;  // float32(4/Pi)
  const float S0 = 1.58962301576546568060e-10f, S1 = -2.50507477628578072866e-8f, S2 = 2.75573136213857245213e-6f,
              S3 = -1.98412698295895385996e-4f, S4 = 8.33333333332211858878e-3f, S5 = -1.66666666666666307295e-1f;
  const float C0 = -1.13585365213876817300e-11f, C1 = 2.08757008419747316778e-9f, C2 = -2.75573141792967388112e-7f,
              C3 = 2.48015872888517045348e-5f, C4 = -1.38888888888730564116e-3f, C5 = 4.16666666666665929218e-2f;
  bool sinSign = x < 0.0f, cosSign = false;
  float ax = absf(x);
  uint32_t j = (uint32_t)(ax * M4PI);
  float y = (float)j;
  if (j & 1u) { j++; y += 1.0f; }
  j &= 7u;
  float z = ((ax - y * PI4A) - y * PI4B) - y * PI4C;
  if (j > 3u) { j -= 4u; sinSign = !sinSign; cosSign = !cosSign; }
  if (j > 1u) cosSign = !cosSign;
  float zz = z * z;
  float c = 1.0f - 0.5f * zz + zz * zz * ((((((C0 * zz) + C1) * zz + C2) * zz + C3) * zz + C4) * zz + C5);
  float s = z + z * zz * ((((((S0 * zz) + S1) * zz + S2) * zz + S3) * zz + S4) * zz + S5);
  bool sw = (j == 1u || j == 2u);
  float s2 = sw ? c : s, c2 = sw ? s : c;
  c2 = cosSign ? -c2 : c2;
  s2 = sinSign ? -s2 : s2;
  // x == 0: Go returns (x, 1)
  s_out = x == 0.0f ? x : s2;
  c_out = x == 0.0f ? 1.0f : c2;
}

// ---- float64 Go routines behind float32(math.X(float64(x))) ----
#define DM_PI 3.14159265358979323846264338327950288

DM_INL double xatan(double x) {
  const double P0 = -8.750608600031904122785e-01, P1 = -1.615753718733365076637e+01, P2 = -7.500855792314704667340e+01,
               P3 = -1.228866684490136173410e+02, P4 = -6.485021904942025371773e+01, Q0 = +2.485846490142306297962e+01,
               Q1 = +1.650270098316988542046e+02, Q2 = +4.328810604912902668951e+02, Q3 = +4.853903996359136964868e+02,
               Q4 = +1.945506571482613964425e+02;
  double z = x * x;
  z = z * ((((P0 * z + P1) * z + P2) * z + P3) * z + P4) / (((((z + Q0) * z + Q1) * z + Q2) * z + Q3) * z + Q4);
  z = x * z + x;
  return z;
}
// satan (go/src/math/atan.go), x >= 0. The three argument ranges are folded into ONE xatan evaluation by
// selecting the reduced argument first (x/1, 1/x or (x-1)/(x+1)): lanes of a wave fall into all three
// ranges, so the branchy form executes xatan three times. Same operations per lane, same results:
//   x <= 0.66        : xatan(x)                          = (0 + z) + 0
//   x > tan(3pi/8)   : Pi/2 - xatan(1/x) + Morebits      = (Pi/2 - z) + Morebits
//   otherwise        : Pi/4 + xatan((x-1)/(x+1)) + 0.5*Morebits
DM_INL double satan(double x) {  // x >= 0
  const double Morebits = 6.123233995736765886130e-17, Tan3pio8 = 2.41421356237309504880;
  const bool lo = x <= 0.66, hi = x > Tan3pio8;
  const double num = lo ? x : (hi ? 1.0 : x - 1.0);
  const double den = lo ? 1.0 : (hi ? x : x + 1.0);
  const double z = xatan(num / den);  // x/1.0 == x exactly
  const double c0 = lo ? 0.0 : (hi ? DM_PI / 2 : DM_PI / 4);
  const double c1 = lo ? 0.0 : (hi ? Morebits : 0.5 * Morebits);
  const double t = hi ? c0 - z : c0 + z;  // 0 + z == z exactly (z > 0 for x > 0; satan(0) is never called)
  return t + c1;
}
DM_INL double atan64(double x) {
  if (x == 0.0) return x;
  const double r = satan(__builtin_fabs(x));
  return x > 0.0 ? r : -r;
}
// math.Atan2 (go/src/math/atan2.go), finite inputs.
// Sector of a circular array without the angle. circarray (cpu_evaluators.go:1042-1092) uses atan2(y, x) only through
// id = floor(float32(atan2) / angle): an integer. A float32 estimate r of the angle (octant reduction, rcp, degree-6
// polynomial in t^2: |r - atan2| < 1.5e-6 including the rounding of the reference's own float32 result) and the reciprocal of
// the wave-uniform `angle` give q = r * inv_angle within (4e-6 + 1.3e-6) * inv_angle of the reference's quotient (its
// |q| <= pi * inv_angle, so the roundings of the reciprocal, the product and the reference's division are <= 4e-7 pi inv_angle);
// floor is monotone, so floor(q - m) == floor(q + m) with m = 6e-6 * inv_angle fixes id. Otherwise -- on a sector boundary,
// at the origin (NaN), for non-finite input -- the caller evaluates the reference's expression. 25 instructions against the
// ~130 of the float64 atan2 + division + floor; checked against that expression on device (gsdf_hip_selftest_circ).
DM_INL bool circ_sector_fast(float x, float y, float inv_angle, float m, float& id) {
  const float ax = absf(x), ay = absf(y);
  const float hi = maxf(ax, ay), lo = minf(ax, ay);
  const float t = lo * __builtin_amdgcn_rcpf(hi);
  const float u = t * t;
  float r = 0.00782548263669014f;
  r = __builtin_fmaf(r, u, -0.03689862787723541f);
  r = __builtin_fmaf(r, u, 0.08374155312776566f);
  r = __builtin_fmaf(r, u, -0.13480405509471893f);
  r = __builtin_fmaf(r, u, 0.19879871606826782f);
  r = __builtin_fmaf(r, u, -0.3332637548446655f);
  r = __builtin_fmaf(r, u, 0.9999993443489075f);
  r = r * t;
  r = ay > ax ? 1.5707964f - r : r;
  r = x < 0.f ? 3.1415927f - r : r;
  r = (__float_as_uint(y) >> 31) ? -r : r;  // by the sign bit: atan2(-0, x < 0) = -pi
  const float q = r * inv_angle;
  const float l = floorf_(q - m), h = floorf_(q + m);
  id = l;
  return l == h;
}

// math.Atan2 as the reference evaluates it: Go's float64 Cephes routine, rounded once (the arithmetic contract).
DM_INL float atan2_ref(float yf, float xf) {
  double y = (double)yf, x = (double)xf;
  double r;
  if (y == 0.0) {
    r = (x >= 0.0 && !__builtin_signbit(x)) ? __builtin_copysign(0.0, y) : __builtin_copysign(DM_PI, y);
  } else if (x == 0.0) {
    r = __builtin_copysign(DM_PI / 2, y);
  } else {
    double q = atan64(y / x);
    r = x < 0.0 ? (q <= 0.0 ? q + DM_PI : q - DM_PI) : q;
  }
  return (float)r;
}

// The same float32 WITHOUT the reference's operation sequence, where that is provably enough (round 6). atan2_ref costs ~115
// instructions: three IEEE float64 divisions, Cephes' rational function in unfused multiplies and adds, 64-bit selects. What the
// contract needs is only float32(g), g = Go's float64 result, and g is within 2^-50 (relative) of the true angle theta (Cephes
// atan: peak relative error 1.8e-16; the quotient's rounding adds 2^-53). So: compute r within e_r of theta by the cheapest
// float64 route --
//   t = lo / hi in [0, 1]  (|y|, |x| sorted; the reciprocal from v_rcp_f32 -- 1 ulp: |1 - hi r0| <= 2^-23 -- and ONE Newton step
//       in float64: relative error e^2 <= 2^-46)
//   atan t = t P(t^2), P of degree 15 fitted to atan(sqrt u)/sqrt u on [0, 1]: relative error 2^-44.28 with the coefficients
//       rounded to binary64 (tools/gen/atan_poly.py; 16 FMAs, 2^-53 each)
//   theta = k pi/2 +- atan t by octant (one FMA: k in {0, 1, 2} times pi/2 is exact up to the constant's own 2^-54), sign of y
// -- e_r <= 2^-43.8 -- and accept when every float64 within ETA = 2^-40 (relative) of r rounds to the same float32:
// float32(r (1 + ETA)) == float32(r (1 - ETA)) bit for bit. Then float32(g) is that float32 too (|g - r| <= (2^-50 + 2^-43.8) |theta|
// < ETA |r|, a 13-fold margin). Rejected: one point in ~2^15 (r within 2^-16 ulp of a rounding boundary) and the inputs the route
// does not cover (hi outside [2^-100, 2^100]: zeros, infinities, NaN, where v_rcp_f32 would leave the normal range) -- the caller
// votes and evaluates atan2_ref for the wave. ~46 instructions. Checked on device against atan2_ref: gsdf_hip_selftest_atan2
// (2^32 hashed pairs of every magnitude and sign, 2^30 pairs searched towards rounding boundaries, lattice-shaped pairs).
DM_INL float atan2_fast(float yf, float xf, bool& ok) {
  const float ax = absf(xf), ay = absf(yf);
  const float hi = maxf(ax, ay), lo = minf(ax, ay);
  const bool swap = ay > ax;
  ok = (__float_as_uint(hi) - 0x0D800000u) < (0x71800000u - 0x0D800000u);  // 2^-100 <= hi < 2^100 (also: not 0, Inf, NaN)
  const double hid = (double)hi, lod = (double)lo;
  double rr = (double)__builtin_amdgcn_rcpf(hi);
  rr = __builtin_fma(rr, __builtin_fma(-hid, rr, 1.0), rr);
  const double t = lod * rr, u = t * t;
  double p = -0x1.cb482a6cb5224p-14;
  p = __builtin_fma(p, u, 0x1.05217b419a6d7p-10);
  p = __builtin_fma(p, u, -0x1.17ce579499acep-8);
  p = __builtin_fma(p, u, 0x1.7bef9c85ccda5p-7);
  p = __builtin_fma(p, u, -0x1.756cd8238c9c5p-6);
  p = __builtin_fma(p, u, 0x1.22637d152cb30p-5);
  p = __builtin_fma(p, u, -0x1.81111fde7ae21p-5);
  p = __builtin_fma(p, u, 0x1.d185bc23a2f84p-5);
  p = __builtin_fma(p, u, -0x1.0ee3841b8af57p-4);
  p = __builtin_fma(p, u, 0x1.3aa7733d3ec56p-4);
  p = __builtin_fma(p, u, -0x1.744e578217a19p-4);
  p = __builtin_fma(p, u, 0x1.c71b1bf660394p-4);
  p = __builtin_fma(p, u, -0x1.24923fb1fa2b1p-3);
  p = __builtin_fma(p, u, 0x1.99999952cd8edp-3);
  p = __builtin_fma(p, u, -0x1.55555554ebae1p-2);
  p = __builtin_fma(p, u, 0x1.ffffffffffe5ap-1);
  const double a = t * p;  // atan(lo / hi) in [0, pi/4]
  // octants: x >= 0: a | pi/2 - a (swapped);  x < 0: pi - a | pi/2 + a (swapped);  x's sign by its bit (atan2(y, -0) = +-pi)
  const uint32_t xneg = __float_as_uint(xf) >> 31, sw = swap ? 1u : 0u;
  const uint32_t k = swap ? 1u : (xneg << 1);
  const double sa = __longlong_as_double(__double_as_longlong(a) ^ ((long long)(sw ^ xneg) << 63));
  double r = __builtin_fma((double)k, DM_PI / 2, sa);  // >= +0
  r = __longlong_as_double(__double_as_longlong(r) | ((long long)(__float_as_uint(yf) >> 31) << 63));  // copysign(r, y)
  const float f1 = (float)__builtin_fma(r, 0x1p-40, r), f2 = (float)__builtin_fma(r, -0x1p-40, r);
  ok = ok && (__float_as_uint(f1) == __float_as_uint(f2));
  return f1;
}
// (single points: selftests, code outside the interpreter's voted call sites)
DM_INL float atan2f_(float yf, float xf) {
  bool ok;
  const float f = atan2_fast(yf, xf, ok);
  return ok ? f : atan2_ref(yf, xf);
}

DM_INL double trig_poly_sin(double z, double zz) {
  return z + z * zz * ((((((1.58962301576546568060e-10 * zz) + -2.50507477628578072866e-8) * zz + 2.75573136213857245213e-6) * zz +
                          -1.98412698295895385996e-4) * zz + 8.33333333332211858878e-3) * zz + -1.66666666666666307295e-1);
}
DM_INL double trig_poly_cos(double zz) {
  return 1.0 - 0.5 * zz + zz * zz * ((((((-1.13585365213876817300e-11 * zz) + 2.08757008419747316778e-9) * zz + -2.75573141792967388112e-7) * zz +
                                        2.48015872888517045348e-5) * zz + -1.38888888888730564116e-3) * zz + 4.16666666666665929218e-2);
}
// math.Cos / math.Sin (go/src/math/sin.go), |x| < 2^29.
DM_INL float cosf_(float xf) {
  double x = __builtin_fabs((double)xf);
  bool sign = false;
  uint64_t j = (uint64_t)(x * (4.0 / DM_PI));
  double y = (double)j;
  if (j & 1) { j++; y += 1.0; }
  j &= 7;
  double z = ((x - y * 7.85398125648498535156e-1) - y * 3.77489470793079817668e-8) - y * 2.69515142907905952645e-15;
  if (j > 3) { j -= 4; sign = !sign; }
  if (j > 1) sign = !sign;
  double zz = z * z;
  double r = (j == 1 || j == 2) ? trig_poly_sin(z, zz) : trig_poly_cos(zz);
  return (float)(sign ? -r : r);
}
DM_INL float sinf_(float xf) {
  double x = (double)xf;
  if (x == 0.0) return xf;
  bool sign = false;
  if (x < 0.0) { x = -x; sign = true; }
  uint64_t j = (uint64_t)(x * (4.0 / DM_PI));
  double y = (double)j;
  if (j & 1) { j++; y += 1.0; }
  j &= 7;
  double z = ((x - y * 7.85398125648498535156e-1) - y * 3.77489470793079817668e-8) - y * 2.69515142907905952645e-15;
  if (j > 3) { sign = !sign; j -= 4; }
  double zz = z * z;
  double r = (j == 1 || j == 2) ? trig_poly_cos(zz) : trig_poly_sin(z, zz);
  return (float)(sign ? -r : r);
}
// math.Cos(x) and math.Sin(x) of the same argument (twist, cpu_evaluators.go:1269-1270): both routines perform the
// identical Cody-Waite reduction and pick between the same two polynomials, so they are evaluated once.
DM_INL void cossinf_(float xf, float& c_out, float& s_out) {
  const double xs = (double)xf;
  double x = __builtin_fabs(xs);
  uint64_t j = (uint64_t)(x * (4.0 / DM_PI));
  double y = (double)j;
  if (j & 1) { j++; y += 1.0; }
  j &= 7;
  const double z = ((x - y * 7.85398125648498535156e-1) - y * 3.77489470793079817668e-8) - y * 2.69515142907905952645e-15;
  bool csign = false, ssign = xs < 0.0;
  if (j > 3) { j -= 4; csign = !csign; ssign = !ssign; }
  if (j > 1) csign = !csign;
  const double zz = z * z;
  const double ps = trig_poly_sin(z, zz), pc = trig_poly_cos(zz);
  const bool sw = (j == 1 || j == 2);
  const double c = sw ? ps : pc, sn = sw ? pc : ps;
  c_out = (float)(csign ? -c : c);
  s_out = xs == 0.0 ? xf : (float)(ssign ? -sn : sn);  // Sin(+-0) = +-0
}
// math.Cos(x) and math.Sin(x) of a twist's angle WITHOUT the reference's operation sequence, where that provably rounds the same (round
// 6; as atan2_fast). cossinf_ costs ~190 instructions: a float64 -> uint64 conversion, 64-bit integer octant logic, Cody-Waite and
// both polynomials in unfused multiplies and adds, 64-bit selects. Here: |x| < 2^20 (else: the caller's fall-back), the octant as a
// 32-bit integer, z = x - y pi/4 by three FMAs -- for y < 2^21 the reference's first two steps are exact (PI4A, PI4B carry 30 and 21
// significant bits: exact products, exact differences) and its third rounds the same real this FMA rounds: the two z differ by at
// most one ulp -- and the reference's own two polynomials (sin.go: _sin, _cos) in FMA Horner form: a few 2^-53 (relative) from the
// unfused evaluation, no cancellation (|z| <= pi/4: the cosine form stays above 0.7, the sine form is z (1 + small)). So r is within
// 2^-49 (relative) of Go's float64 g, and float32(g) is decided wherever every float64 within 2^-44 of r rounds to one float32:
// float32(r (1 + 2^-44)) == float32(r (1 - 2^-44)), bit for bit, for both results. One value in ~2^19 is rejected. ~62 instructions.
// Checked against cossinf_ on device for EVERY float32 argument (gsdf_hip_selftest_cossin: 2^32 bit patterns).
DM_INL void cossin_fast(float xf, float& c_out, float& s_out, bool& ok) {
  const double xs = (double)xf;
  const double x = __builtin_fabs(xs);
  ok = absf(xf) < 1048576.0f;  // 2^20 (NaN: false)
  int j = (int)(x * (4.0 / DM_PI));  // truncates; < 1.34e6
  j += j & 1;                        // "map zeros to origin": odd octants belong to the next even one
  const double y = (double)j;
  double z = __builtin_fma(-y, 7.85398125648498535156e-1, x);
  z = __builtin_fma(-y, 3.77489470793079817668e-8, z);
  z = __builtin_fma(-y, 2.69515142907905952645e-15, z);
  const double zz = z * z;
  double ps = 1.58962301576546568060e-10;
  ps = __builtin_fma(ps, zz, -2.50507477628578072866e-8);
  ps = __builtin_fma(ps, zz, 2.75573136213857245213e-6);
  ps = __builtin_fma(ps, zz, -1.98412698295895385996e-4);
  ps = __builtin_fma(ps, zz, 8.33333333332211858878e-3);
  ps = __builtin_fma(ps, zz, -1.66666666666666307295e-1);
  ps = __builtin_fma(z * zz, ps, z);
  double pc = -1.13585365213876817300e-11;
  pc = __builtin_fma(pc, zz, 2.08757008419747316778e-9);
  pc = __builtin_fma(pc, zz, -2.75573141792967388112e-7);
  pc = __builtin_fma(pc, zz, 2.48015872888517045348e-5);
  pc = __builtin_fma(pc, zz, -1.38888888888730564116e-3);
  pc = __builtin_fma(pc, zz, 4.16666666666665929218e-2);
  pc = __builtin_fma(zz * zz, pc, __builtin_fma(-0.5, zz, 1.0));
  const uint32_t j8 = (uint32_t)j & 7u, hi = j8 >> 2, j4 = j8 & 3u;  // octant mod 8; "j > 3: j -= 4, both signs flip"
  const bool sw = j4 == 1u || j4 == 2u;                              // the sine series gives the cosine and the other way round
  const uint32_t csign = hi ^ (j4 >> 1), ssign = hi ^ (xf < 0.0f ? 1u : 0u);
  const double c0 = sw ? ps : pc, s0 = sw ? pc : ps;
  const double c = __longlong_as_double(__double_as_longlong(c0) ^ ((long long)csign << 63));
  const double sn = __longlong_as_double(__double_as_longlong(s0) ^ ((long long)ssign << 63));
  const float c1 = (float)__builtin_fma(c, 0x1p-44, c), c2 = (float)__builtin_fma(c, -0x1p-44, c);
  const float s1 = (float)__builtin_fma(sn, 0x1p-44, sn), s2 = (float)__builtin_fma(sn, -0x1p-44, sn);
  ok = ok && __float_as_uint(c1) == __float_as_uint(c2) && __float_as_uint(s1) == __float_as_uint(s2);
  c_out = c1;
  s_out = xf == 0.0f ? xf : s1;  // Sin(+-0) = +-0
}

// math.Acos = Pi/2 - Asin (go/src/math/asin.go)
DM_INL float acosf_(float xf) {
  double x = (double)xf;
  double as;
  if (x == 0.0) {
    as = x;
  } else {
    bool sign = x < 0.0;
    x = __builtin_fabs(x);
    if (x > 1.0) return __builtin_nanf("");
    double temp = __builtin_sqrt(1.0 - x * x);
    if (x > 0.7) temp = DM_PI / 2 - satan(temp / x);
    else temp = satan(x / temp);
    as = sign ? -temp : temp;
  }
  return (float)(DM_PI / 2 - as);
}
// math.Cbrt (go/src/math/cbrt.go), x finite.
DM_INL float cbrtf_(float xf) {
  double x = (double)xf;
  if (x == 0.0) return xf;
  const double C = 5.42857142857142815906e-01, D = -7.05306122448979611050e-01, E = 1.41428571428571436819e+00,
               F = 1.60714285714285720630e+00, G = 3.57142857142857150787e-01;
  bool sign = x < 0.0;
  x = __builtin_fabs(x);
  double t = __longlong_as_double((long long)((unsigned long long)__double_as_longlong(x) / 3ull + (715094163ull << 32)));
  // float32 inputs converted to double are never double-subnormal: SmallestNormal branch unreachable.
  double r = t * t / x;
  double s = C + r * t;
  t *= G + F / (s + E + D / s);
  t = __longlong_as_double((long long)(((unsigned long long)__double_as_longlong(t) & (0xFFFFFFFFCull << 28)) + (1ull << 30)));
  s = t * t;
  r = x / s;
  double w = t + t;
  r = (r - t) / (w + r);
  t = t + t * r;
  return (float)(sign ? -t : t);
}
// math32.Pow(x, 1/3) for x >= 0 (gsdf.go:183): exp(yf*log(x)) -- see oracle note; agrees with the
// CPU restatement to <= 1 ulp (ocml log/exp vs libm), quadbezier2d only.
DM_INL float pow13f_(float x) {
  if (x == 0.0f) return 0.0f;
  const float yf = 0.3333333432674407958984375f;
  float l = (float)__ocml_log_f64((double)x);
  return (float)__ocml_exp_f64((double)(yf * l));
}

}  // namespace dm
