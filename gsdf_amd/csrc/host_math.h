// host_math.h -- the few float32 routines of github.com/chewxy/math32 v1.11.1 (go.mod:8) that the lowering (compile.cpp)
// needs for loop-invariant constants: Sincos (float32 port of go/src/math/sincos.go: arc2D's half angle,
// cpu_evaluators.go:567; the circular arrays' rotation tables, :1074-1075) and Tan (float32(math.Tan(float64)): the screw's
// taper, threads.go:157). The device's versions are in dev_math.h; the scaffold's tree builder (scaffold/ms.hpp) uses these too.
#pragma once
#include <cmath>
#include <cstdint>
#include <utility>

namespace gsdf {

constexpr double kPi = 3.14159265358979323846264338327950288419716939937510582097494459;
constexpr float kPiF = (float)kPi;

// math32.Sincos: float32 port of go/src/math/sincos.go.
inline void sincosf32(float x, float& s_out, float& c_out) {
  const float PI4A = 7.85398125648498535156e-1f, PI4B = 3.77489470793079817668e-8f, PI4C = 2.69515142907905952645e-15f;
  const float M4PI = (float)(4.0 / kPi);
  static const float S[6] = {1.58962301576546568060e-10f, -2.50507477628578072866e-8f, 2.75573136213857245213e-6f,
                             -1.98412698295895385996e-4f, 8.33333333332211858878e-3f, -1.66666666666666307295e-1f};
  static const float C[6] = {-1.13585365213876817300e-11f, 2.08757008419747316778e-9f, -2.75573141792967388112e-7f,
                             2.48015872888517045348e-5f,   -1.38888888888730564116e-3f, 4.16666666666665929218e-2f};
  if (x == 0) { s_out = x; c_out = 1; return; }
  if (x != x || std::isinf(x)) { s_out = NAN; c_out = NAN; return; }
  bool sinSign = false, cosSign = false;
  if (x < 0) { x = -x; sinSign = true; }
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
  if (j == 1 || j == 2) std::swap(s, c);
  if (cosSign) c = -c;
  if (sinSign) s = -s;
  s_out = s; c_out = c;
}
// float32(math.Tan(float64)) (math32 doc.go): Go's own routine (go/src/math/tan.go; Cephes), arguments below the Payne-Hanek
// threshold (a thread's taper angle), evaluated in double and rounded once.
inline float tanf32(float x) {
  static const double P[3] = {-1.30936939181383777646e4, 1.15351664838587416140e6, -1.79565251976484877988e7};
  static const double Q[5] = {1.0, 1.36812963470692954678e4, -1.32089234440210967447e6, 2.50083801823357915839e7, -5.38695755929454629881e7};
  const double PI4A = 7.85398125648498535156e-1, PI4B = 3.77489470793079817668e-8, PI4C = 2.69515142907905952645e-15;
  double v = (double)x;
  if (v == 0 || v != v) return x;
  if (std::isinf(v)) return NAN;
  bool sign = false;
  if (v < 0) { v = -v; sign = true; }
  uint64_t j = (uint64_t)(v * (4 / kPi));
  double y = (double)j;
  if (j & 1) { j++; y++; }
  const double z = ((v - y * PI4A) - y * PI4B) - y * PI4C, zz = z * z;
  if (zz > 1e-14) y = z + z * (zz * (((P[0] * zz) + P[1]) * zz + P[2]) / ((((zz + Q[1]) * zz + Q[2]) * zz + Q[3]) * zz + Q[4]));
  else y = z;
  if (j & 2) y = -1 / y;
  return (float)(sign ? -y : y);
}
// float32(math.Log2(float64)) for finite x > 0 (makeICube's level count, octreerenderer.go:222-235): go/src/math/log10.go over
// log.go (FreeBSD e_log.c) -- Frexp first, so powers of two are exact.
inline float log2f32(float x) {
  static const double Ln2Hi = 6.93147180369123816490e-01, Ln2Lo = 1.90821492927058770002e-10, L1 = 6.666666666666735130e-01,
                      L2 = 3.999999999940941908e-01, L3 = 2.857142874366239149e-01, L4 = 2.222219843214978396e-01,
                      L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01, L7 = 1.479819860511658591e-01;
  const double v = (double)x;
  if (!(v > 0) || std::isinf(v)) return (float)std::log2(v);  // special values: as IEEE says
  int e;
  const double frac = std::frexp(v, &e);
  if (frac == 0.5) return (float)(e - 1);
  int ki;
  double f1 = std::frexp(frac, &ki);
  if (f1 < 1.41421356237309504880168872420969808 / 2) { f1 *= 2; ki--; }
  const double f = f1 - 1, k = (double)ki;
  const double s = f / (2 + f), s2 = s * s, s4 = s2 * s2;
  const double t1 = s2 * (L1 + s4 * (L3 + s4 * (L5 + s4 * L7))), t2 = s4 * (L2 + s4 * (L4 + s4 * L6));
  const double R = t1 + t2, hfsq = 0.5 * f * f;
  const double lg = k * Ln2Hi - ((hfsq - (s * (hfsq + R) + k * Ln2Lo)) - f);
  return (float)(lg * (1 / 0.693147180559945309417232121458176568) + (double)e);
}

}  // namespace gsdf
