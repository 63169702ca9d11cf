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
// float32(math.Tan(float64)) (math32 doc.go); libm double is within 1 ulp(double) of Go's.
inline float tanf32(float x) { return (float)std::tan((double)x); }

}  // namespace gsdf
