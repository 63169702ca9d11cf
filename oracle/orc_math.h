/*
 * orc_math.h -- ORACLE (test infrastructure, never shipped, never linked into the product).
 *
 * CPU restatement of the scalar math the reference's CPU evaluators call. The arithmetic lives in
 * two third-party Go modules that are NOT vendored in /root/reference:
 *   github.com/chewxy/math32   v1.11.1                              (go.mod:8)
 *   github.com/soypat/geometry v0.0.0-20251107203642-291c5648d529   (go.mod:11) ms1/ms2/ms3 helpers
 * so this file restates their published algorithms from the package documentation:
 *   math32 doc.go: "mostly just a wrapper in form of float32(math.XXX). This applies to ... Acos,
 *   Asin, Atan, Atan2, Cbrt, Cos, Log2, Sin, Tan ... Everything else is a float32 implementation."
 *   -> wrappers are restated as Go's pure-Go float64 routines (go/src/math: atan.go, atan2.go,
 *      sin.go, asin.go, cbrt.go; Cephes-derived) evaluated in double and rounded once to float;
 *   -> Sqrt, Hypot, Floor, Round, Min, Max, Sincos, Pow ... are restated in float32.
 * PARITY UNPINNED for these externals (no Go toolchain / module source in this container): the
 * restatement is anchored on the reference's call sites and count-level known answers only.
 *
 * Everything must be compiled with -ffp-contract=off -fno-fast-math (Go on amd64 never fuses).
 */
#ifndef ORC_MATH_H
#define ORC_MATH_H
#include <math.h>
#include <stdint.h>
#include <string.h>

static inline uint32_t orc_f32bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float orc_f32frombits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint64_t orc_f64bits(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
static inline double orc_f64frombits(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }

#define ORC_PI 3.14159265358979323846264338327950288419716939937510582097494459
#define ORC_PI_F ((float)ORC_PI)

/* ---------------- float32-native (math32) ---------------- */
static inline float go_absf(float x) { return orc_f32frombits(orc_f32bits(x) & 0x7fffffffu); }
static inline float go_copysignf(float x, float s) {
  return orc_f32frombits((orc_f32bits(x) & 0x7fffffffu) | (orc_f32bits(s) & 0x80000000u));
}
static inline float go_sqrtf(float x) { return sqrtf(x); } /* IEEE correctly rounded */
static inline float go_floorf(float x) { return floorf(x); } /* exact */
static inline float go_ceilf(float x) { return ceilf(x); }   /* exact */
/* Go Round: half away from zero == C roundf. */
static inline float go_roundf(float x) { return roundf(x); }

/* math.Min / math.Max semantics (go/src/math/dim.go): NaN propagates, -0 < +0, infinities first. */
static inline float go_minf(float x, float y) {
  if (isinf(x) && x < 0) return x;
  if (isinf(y) && y < 0) return y;
  if (x != x || y != y) return NAN;
  if (x == 0 && x == y) return signbit(x) ? x : y;
  return x < y ? x : y;
}
static inline float go_maxf(float x, float y) {
  if (isinf(x) && x > 0) return x;
  if (isinf(y) && y > 0) return y;
  if (x != x || y != y) return NAN;
  if (x == 0 && x == y) return signbit(x) ? y : x;
  return x > y ? x : y;
}

/* math32.Hypot: float32 port of go/src/math/hypot.go. */
static inline float go_hypotf(float p, float q) {
  if (isinf(p) || isinf(q)) return INFINITY;
  if (p != p || q != q) return NAN;
  p = go_absf(p);
  q = go_absf(q);
  if (p < q) { float t = p; p = q; q = t; }
  if (p == 0) return 0;
  q = q / p;
  return p * go_sqrtf(1 + q * q);
}

/* math32.Sincos: float32 port of go/src/math/sincos.go (Cephes sinf/cosf range reduction). */
static inline void go_sincosf(float x, float* s_out, float* c_out) {
  const float PI4A = 7.85398125648498535156e-1f;
  const float PI4B = 3.77489470793079817668e-8f;
  const float PI4C = 2.69515142907905952645e-15f;
  const float M4PI = (float)(4.0 / ORC_PI);
  static const float S[6] = {1.58962301576546568060e-10f, -2.50507477628578072866e-8f,
                             2.75573136213857245213e-6f,  -1.98412698295895385996e-4f,
                             8.33333333332211858878e-3f,  -1.66666666666666307295e-1f};
  static const float C[6] = {-1.13585365213876817300e-11f, 2.08757008419747316778e-9f,
                             -2.75573141792967388112e-7f,  2.48015872888517045348e-5f,
                             -1.38888888888730564116e-3f,  4.16666666666665929218e-2f};
  if (x == 0) { *s_out = x; *c_out = 1; return; }
  if (x != x || isinf(x)) { *s_out = NAN; *c_out = NAN; return; }
  int sinSign = 0, cosSign = 0;
  if (x < 0) { x = -x; sinSign = 1; }
  uint64_t j = (uint64_t)(x * M4PI);
  float y = (float)j;
  if (j & 1) { j++; y++; }
  j &= 7;
  float z = ((x - y * PI4A) - y * PI4B) - y * PI4C;
  if (j > 3) { j -= 4; sinSign = !sinSign; cosSign = !cosSign; }
  if (j > 1) cosSign = !cosSign;
  float zz = z * z;
  float c = 1.0f - 0.5f * zz + zz * zz * ((((((C[0] * zz) + C[1]) * zz + C[2]) * zz + C[3]) * zz + C[4]) * zz + C[5]);
  float s = z + z * zz * ((((((S[0] * zz) + S[1]) * zz + S[2]) * zz + S[3]) * zz + S[4]) * zz + S[5]);
  if (j == 1 || j == 2) { float t = s; s = c; c = t; }
  if (cosSign) c = -c;
  if (sinSign) s = -s;
  *s_out = s;
  *c_out = c;
}

/* ---------------- float64 Go routines behind the float32(math.X(float64)) wrappers ---------------- */
static inline double go_xatan(double x) {
  const double P0 = -8.750608600031904122785e-01, P1 = -1.615753718733365076637e+01,
               P2 = -7.500855792314704667340e+01, P3 = -1.228866684490136173410e+02,
               P4 = -6.485021904942025371773e+01, Q0 = +2.485846490142306297962e+01,
               Q1 = +1.650270098316988542046e+02, Q2 = +4.328810604912902668951e+02,
               Q3 = +4.853903996359136964868e+02, Q4 = +1.945506571482613964425e+02;
  double z = x * x;
  z = z * ((((P0 * z + P1) * z + P2) * z + P3) * z + P4) / (((((z + Q0) * z + Q1) * z + Q2) * z + Q3) * z + Q4);
  z = x * z + x;
  return z;
}
static inline double go_satan(double x) {
  const double Morebits = 6.123233995736765886130e-17;
  const double Tan3pio8 = 2.41421356237309504880;
  if (x <= 0.66) return go_xatan(x);
  if (x > Tan3pio8) return ORC_PI / 2 - go_xatan(1 / x) + Morebits;
  return ORC_PI / 4 + go_xatan((x - 1) / (x + 1)) + 0.5 * Morebits;
}
static inline double go_atan64(double x) {
  if (x == 0) return x;
  if (x > 0) return go_satan(x);
  return -go_satan(-x);
}
static inline double go_atan2_64(double y, double x) {
  if (y != y || x != x) return NAN;
  if (y == 0) {
    if (x >= 0 && !signbit(x)) return copysign(0.0, y);
    return copysign(ORC_PI, y);
  }
  if (x == 0) return copysign(ORC_PI / 2, y);
  if (isinf(x)) {
    if (x > 0) return isinf(y) ? copysign(ORC_PI / 4, y) : copysign(0.0, y);
    return isinf(y) ? copysign(3 * ORC_PI / 4, y) : copysign(ORC_PI, y);
  }
  if (isinf(y)) return copysign(ORC_PI / 2, y);
  double q = go_atan64(y / x);
  if (x < 0) return q <= 0 ? q + ORC_PI : q - ORC_PI;
  return q;
}
static inline float go_atan2f(float y, float x) { return (float)go_atan2_64((double)y, (double)x); }

static const double GO_SIN64[6] = {1.58962301576546568060e-10, -2.50507477628578072866e-8,
                                   2.75573136213857245213e-6,  -1.98412698295895385996e-4,
                                   8.33333333332211858878e-3,  -1.66666666666666307295e-1};
static const double GO_COS64[6] = {-1.13585365213876817300e-11, 2.08757008419747316778e-9,
                                   -2.75573141792967388112e-7,  2.48015872888517045348e-5,
                                   -1.38888888888730564116e-3,  4.16666666666665929218e-2};
#define GO_PI4A 7.85398125648498535156e-1
#define GO_PI4B 3.77489470793079817668e-8
#define GO_PI4C 2.69515142907905952645e-15

static inline double go_cos64(double x) {
  if (x != x || isinf(x)) return NAN;
  int sign = 0;
  x = fabs(x);
  uint64_t j = (uint64_t)(x * (4 / ORC_PI));
  double y = (double)j;
  if (j & 1) { j++; y++; }
  j &= 7;
  double z = ((x - y * GO_PI4A) - y * GO_PI4B) - y * GO_PI4C;
  if (j > 3) { j -= 4; sign = !sign; }
  if (j > 1) sign = !sign;
  double zz = z * z;
  if (j == 1 || j == 2)
    y = z + z * zz * ((((((GO_SIN64[0] * zz) + GO_SIN64[1]) * zz + GO_SIN64[2]) * zz + GO_SIN64[3]) * zz + GO_SIN64[4]) * zz + GO_SIN64[5]);
  else
    y = 1.0 - 0.5 * zz + zz * zz * ((((((GO_COS64[0] * zz) + GO_COS64[1]) * zz + GO_COS64[2]) * zz + GO_COS64[3]) * zz + GO_COS64[4]) * zz + GO_COS64[5]);
  return sign ? -y : y;
}
static inline double go_sin64(double x) {
  if (x == 0 || x != x) return x;
  if (isinf(x)) return NAN;
  int sign = 0;
  if (x < 0) { x = -x; sign = 1; }
  uint64_t j = (uint64_t)(x * (4 / ORC_PI));
  double y = (double)j;
  if (j & 1) { j++; y++; }
  j &= 7;
  double z = ((x - y * GO_PI4A) - y * GO_PI4B) - y * GO_PI4C;
  if (j > 3) { sign = !sign; j -= 4; }
  double zz = z * z;
  if (j == 1 || j == 2)
    y = 1.0 - 0.5 * zz + zz * zz * ((((((GO_COS64[0] * zz) + GO_COS64[1]) * zz + GO_COS64[2]) * zz + GO_COS64[3]) * zz + GO_COS64[4]) * zz + GO_COS64[5]);
  else
    y = z + z * zz * ((((((GO_SIN64[0] * zz) + GO_SIN64[1]) * zz + GO_SIN64[2]) * zz + GO_SIN64[3]) * zz + GO_SIN64[4]) * zz + GO_SIN64[5]);
  return sign ? -y : y;
}
static inline float go_cosf(float x) { return (float)go_cos64((double)x); }
static inline float go_sinf(float x) { return (float)go_sin64((double)x); }

static inline double go_asin64(double x) {
  if (x == 0) return x;
  int sign = 0;
  if (x < 0) { x = -x; sign = 1; }
  if (x > 1) return NAN;
  double temp = sqrt(1 - x * x);
  if (x > 0.7) temp = ORC_PI / 2 - go_satan(temp / x);
  else temp = go_satan(x / temp);
  return sign ? -temp : temp;
}
static inline float go_acosf(float x) { return (float)(ORC_PI / 2 - go_asin64((double)x)); }

/* go/src/math/cbrt.go (FreeBSD s_cbrt.c derived). */
static inline double go_cbrt64(double x) {
  const uint64_t B1 = 715094163, B2 = 696219795;
  const double C = 5.42857142857142815906e-01, D = -7.05306122448979611050e-01,
               E = 1.41428571428571436819e+00, F = 1.60714285714285720630e+00,
               G = 3.57142857142857150787e-01, SmallestNormal = 2.22507385850720138309e-308;
  if (x == 0 || x != x || isinf(x)) return x;
  int sign = 0;
  if (x < 0) { x = -x; sign = 1; }
  double t = orc_f64frombits(orc_f64bits(x) / 3 + (B1 << 32));
  if (x < SmallestNormal) {
    t = (double)((uint64_t)1 << 54);
    t *= x;
    t = orc_f64frombits(orc_f64bits(t) / 3 + (B2 << 32));
  }
  double r = t * t / x;
  double s = C + r * t;
  t *= G + F / (s + E + D / s);
  t = orc_f64frombits((orc_f64bits(t) & ((uint64_t)0xFFFFFFFFC << 28)) + ((uint64_t)1 << 30));
  s = t * t;
  r = x / s;
  double w = t + t;
  r = (r - t) / (w + r);
  t = t + t * r;
  return sign ? -t : t;
}
static inline float go_cbrtf(float x) { return (float)go_cbrt64((double)x); }

/* math.Tan (go/src/math/tan.go; Cephes tan.c), arguments below the Payne-Hanek threshold (the call sites pass a thread's taper
 * angle: a fraction of a radian). */
static inline double go_tan64(double x) {
  static const double P[3] = {-1.30936939181383777646e4, 1.15351664838587416140e6, -1.79565251976484877988e7};
  static const double Q[5] = {1.0, 1.36812963470692954678e4, -1.32089234440210967447e6, 2.50083801823357915839e7, -5.38695755929454629881e7};
  if (x == 0 || x != x) return x;
  if (isinf(x)) return NAN;
  int sign = 0;
  if (x < 0) { x = -x; sign = 1; }
  uint64_t j = (uint64_t)(x * (4 / ORC_PI)); /* integer part of x / (Pi/4) */
  double y = (double)j;
  if (j & 1) { j++; y++; } /* map zeros and singularities to origin */
  double z = ((x - y * GO_PI4A) - y * GO_PI4B) - y * GO_PI4C;
  double zz = z * z;
  if (zz > 1e-14) y = z + z * (zz * (((P[0] * zz) + P[1]) * zz + P[2]) / ((((zz + Q[1]) * zz + Q[2]) * zz + Q[3]) * zz + Q[4]));
  else y = z;
  if (j & 2) y = -1 / y;
  return sign ? -y : y;
}
/* math.Log (go/src/math/log.go; FreeBSD e_log.c) for finite x > 0, and math.Log2 (log10.go: Frexp, exact for powers of two). */
static inline double go_log64(double x) {
  static const double Ln2Hi = 6.93147180369123816490e-01, Ln2Lo = 1.90821492927058770002e-10, L1 = 6.666666666666735130e-01,
                      L2 = 3.999999999940941908e-01, L3 = 2.857142874366239149e-01, L4 = 2.222219843214978396e-01,
                      L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01, L7 = 1.479819860511658591e-01;
  if (x != x || isinf(x)) return x > 0 || x != x ? x : NAN;
  if (x < 0) return NAN;
  if (x == 0) return -INFINITY;
  int ki;
  double f1 = frexp(x, &ki);
  if (f1 < 1.41421356237309504880168872420969808 / 2) { f1 *= 2; ki--; }
  double f = f1 - 1, k = (double)ki;
  double s = f / (2 + f), s2 = s * s, s4 = s2 * s2;
  double t1 = s2 * (L1 + s4 * (L3 + s4 * (L5 + s4 * L7))), t2 = s4 * (L2 + s4 * (L4 + s4 * L6));
  double R = t1 + t2, hfsq = 0.5 * f * f;
  return k * Ln2Hi - ((hfsq - (s * (hfsq + R) + k * Ln2Lo)) - f);
}
static inline double go_log2_64(double x) {
  int e;
  double frac = frexp(x, &e);
  if (frac == 0.5) return (double)(e - 1); /* exact for powers of two */
  return go_log64(frac) * (1 / 0.693147180559945309417232121458176568) + (double)e;
}
/* math32's float32(math.XXX(float64(x))) wrappers, as everywhere in this file: Go's own float64 routines, rounded once */
static inline float go_tanf(float x) { return (float)go_tan64((double)x); }
static inline float go_atanf(float x) { return (float)go_atan64((double)x); }
static inline float go_log2f(float x) { return (float)go_log2_64((double)x); }

/* math32.Pow(x, y) for the only call site (gsdf.go:183, y = 1/3, x >= 0): float32 port of
 * go/src/math/pow.go: yi,yf = Modf(y) -> yi=0, yf=1/3 ; a1 = Exp(yf*Log(x)) ; then Frexp/Ldexp
 * bookkeeping that is the identity for yi = 0. math32.Exp / math32.Log have amd64 assembly in
 * math32; restated here as float32(exp(float64)) of the float32 product (best effort; quadbezier
 * is outside every config). */
static inline float go_pow13f(float x) {
  if (x == 0) return 0;
  if (x != x) return NAN;
  if (isinf(x)) return INFINITY;
  const float yf = (float)(1.0 / 3.0);
  float l = (float)log((double)x);
  return (float)exp((double)(yf * l));
}

/* ---------------- gsdf.go:141-189 helpers ---------------- */
static inline float orc_signf(float a) { return a == 0 ? 0.0f : go_copysignf(1.0f, a); }
static inline float orc_clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline float orc_mixf(float x, float y, float a) { return x * (1 - a) + y * a; }

#endif
