/* orc_math_export.c -- ORACLE: exported wrappers so tests can check the math restatement
 * (orc_math.h) against high-precision references. fn: 0 hypot 1 atan2 2 sin 3 cos 4 acos 5 cbrt
 * 6 sincos.s 7 sincos.c 8 min 9 max 10 pow13 11 round 12 floor */
#include "orc_math.h"
#include <stddef.h>

void orc_math_apply(int fn, const float* x, const float* y, float* out, size_t n) {
  for (size_t i = 0; i < n; i++) {
    float s, c;
    switch (fn) {
      case 0: out[i] = go_hypotf(x[i], y[i]); break;
      case 1: out[i] = go_atan2f(x[i], y[i]); break; /* atan2(y=x[i], x=y[i]) */
      case 2: out[i] = go_sinf(x[i]); break;
      case 3: out[i] = go_cosf(x[i]); break;
      case 4: out[i] = go_acosf(x[i]); break;
      case 5: out[i] = go_cbrtf(x[i]); break;
      case 6: go_sincosf(x[i], &s, &c); out[i] = s; break;
      case 7: go_sincosf(x[i], &s, &c); out[i] = c; break;
      case 8: out[i] = go_minf(x[i], y[i]); break;
      case 9: out[i] = go_maxf(x[i], y[i]); break;
      case 10: out[i] = go_pow13f(x[i]); break;
      case 11: out[i] = go_roundf(x[i]); break;
      case 12: out[i] = go_floorf(x[i]); break;
      default: out[i] = NAN;
    }
  }
}
