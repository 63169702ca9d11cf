// kernels_eval.h -- point evaluation kernels: gleval.SDF3 / SDF2.Evaluate (eval_kernel), NormalsCentralDiff (normals_kernel),
// ImageRendererSDF2 (image2_kernel) and the exhaustive self-tests of the device math (sqrt / circular-array sector / exact division).
#pragma once
#include "kernels_common.h"

// dist[i] = SDF(pos[i]). Each lane carries K points per interpreter pass (tile = K*BLOCK points,
// point kp of lane t = tile + kp*BLOCK + t, so every load/store stays coalesced).
// W = workgroups per CU the register budget is sized for: the host asks for 4 (<= 128 VGPRs) whenever 4 workgroups'
// slot columns fit the 160 KB of LDS -- the 4th wave per SIMD is worth more than the spills (interpreter build of
// npt-flange: flat lattice 63 -> 73 G evals/s).
template <int DIM, int K, int W = (K == 1 ? 4 : 3)>
__global__ void __launch_bounds__(BLOCK, W) eval_kernel(const uint32_t* __restrict__ code_g, const float* __restrict__ pos,
                                                     uint32_t stride_f, float* __restrict__ dist, uint64_t n) {
  code_ptr code = as_code(code_g);
  float* lds = g_smem + threadIdx.x;
  const uint64_t step = (uint64_t)gridDim.x * BLOCK * K;
  for (uint64_t base = (uint64_t)blockIdx.x * BLOCK * K; base < n; base += step) {  // uniform trip count
    P3 p[K];
    float d[K];
#pragma unroll
    for (int kp = 0; kp < K; kp++) {
      const uint64_t i = base + (uint64_t)kp * BLOCK + threadIdx.x;
      p[kp] = P3{0.f, 0.f, 0.f};
      if (i < n) {
        const float* q = pos + i * stride_f;
        p[kp].x = q[0];
        p[kp].y = q[1];
        if (DIM == 3) p[kp].z = q[2];
      }
    }
    gsdf_dev::sdf_eval<K>(code, p, d, lds, BLOCK);
#pragma unroll
    for (int kp = 0; kp < K; kp++) {
      const uint64_t i = base + (uint64_t)kp * BLOCK + threadIdx.x;
      if (i < n) dist[i] = d[kp];
    }
  }
}

// glrender.ImageRendererSDF2.Render (image.go:76-118) with the default black/white/red conversion (:52-61):
// pixel (i,j) samples (xmin + i*dx, ymax - j*dy); dist gets the raw distances, rgba the converted pixels.
template <int K>
__global__ void __launch_bounds__(BLOCK, 3) image2_kernel(const uint32_t* __restrict__ code_g, int w, int h, float xmin, float ymax,
                                                          float dx, float dy, float* __restrict__ dist, uint32_t* __restrict__ rgba) {
  code_ptr code = as_code(code_g);
  float* lds = g_smem + threadIdx.x;
  const uint64_t n = (uint64_t)w * (uint64_t)h;
  const uint64_t step = (uint64_t)gridDim.x * BLOCK * K;
  for (uint64_t base = (uint64_t)blockIdx.x * BLOCK * K; base < n; base += step) {
    P3 p[K];
    float d[K];
#pragma unroll
    for (int kp = 0; kp < K; kp++) {
      uint64_t i = base + (uint64_t)kp * BLOCK + threadIdx.x;
      if (i >= n) i = n - 1;
      const unsigned px = (unsigned)(i % (uint64_t)w), py = (unsigned)(i / (uint64_t)w);
      p[kp] = P3{(float)px * dx + xmin, ymax - (float)py * dy, 0.f};
    }
    gsdf_dev::sdf_eval<K>(code, p, d, lds, BLOCK);
#pragma unroll
    for (int kp = 0; kp < K; kp++) {
      const uint64_t i = base + (uint64_t)kp * BLOCK + threadIdx.x;
      if (i < n) {
        const float v = d[kp];
        if (dist) dist[i] = v;
        const bool bad = nb::nan_or_inf(v);
        if (rgba) rgba[i] = bad ? 0xff0000ffu : (v > 0.f ? 0xffffffffu : 0xff000000u);  // R,G,B,A bytes little-endian
      }
    }
  }
}

// Completion flag of a host-buffer call (abi_eval.hip: eval_submit / eval_wait): runs behind the evaluating kernel on its
// stream and writes `v` to a word of pinned host memory the waiting thread polls. (The evaluating kernel's own stores to the
// caller's mapped buffers are released at its end; PCIe keeps posted writes in order, so the host sees them before the flag.)
__global__ void signal_kernel(unsigned* flag, unsigned v) { __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

// Exhaustive self-test of dm::sqrt_1to2 over every float in [1, 2] ...
__global__ void __launch_bounds__(BLOCK) sqrt_selftest_kernel(unsigned long long* __restrict__ bad) {
  unsigned long long nb = 0;
  for (unsigned i = 0x3f800000u + blockIdx.x * BLOCK + threadIdx.x; i <= 0x40000000u; i += gridDim.x * BLOCK) {
    const float s = __uint_as_float(i);
    if (__float_as_uint(dm::sqrt_1to2(s)) != __float_as_uint(__builtin_sqrtf(s))) nb++;
  }
  // ... and of dm::sqrt_core (dm::sqrt_k's route) over every float bit pattern it may be given: s >= 2^-96, +Inf, NaN (a NaN for a NaN)
  for (unsigned long long j = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x; j < (1ull << 32); j += (unsigned long long)gridDim.x * BLOCK) {
    const float s = __uint_as_float((unsigned)j);
    const unsigned sb = (unsigned)j;
    if ((sb >> 31) != 0u || sb < 0x0f800000u) continue;  // (negatives, zeros, below 2^-96: the wave takes the compiler's expansion)
    const unsigned a = __float_as_uint(dm::sqrt_core(s)), b = __float_as_uint(__builtin_sqrtf(s));
    const bool an = (a & 0x7fffffffu) > 0x7f800000u, bn = (b & 0x7fffffffu) > 0x7f800000u;
    if (an || bn ? an != bn : a != b) nb++;
  }
  if (nb) atomicAdd(bad, nb);
}

// Self-test of dm::circ_sector_fast against the reference's expression floor(float32(atan2(y, x)) / angle): 2^32 points --
// even indices: both coordinates from a hash, all magnitudes and signs (exponents 2^-40 .. 2^40, plus zeros); odd indices:
// points within 1e-3 .. 1e-9 rad of a sector boundary at radii 1e-2 .. 1e2. bad counts the points where the fast path decides
// and differs; fast_count the points where it decides.
__global__ void __launch_bounds__(BLOCK) circ_selftest_kernel(float angle, unsigned long long* __restrict__ bad,
                                                              unsigned long long* __restrict__ fast_count) {
  unsigned long long nb = 0, nf = 0;
  const float inv_angle = __builtin_amdgcn_rcpf(angle), m = 6e-6f * inv_angle;
  auto hash = [](unsigned v) { v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16; return v; };
  for (unsigned long long i = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x; i < (1ull << 32); i += (unsigned long long)gridDim.x * BLOCK) {
    const unsigned h1 = hash((unsigned)i), h2 = hash((unsigned)i ^ 0x9e3779b9u);
    float x, y;
    if ((i & 1ull) == 0ull) {
      // sign | exponent 87..167 | 23 mantissa bits; one in 64 is a zero
      x = __uint_as_float((h1 & 0x80000000u) | ((87u + (h1 >> 8) % 81u) << 23) | (h2 & 0x7fffffu));
      y = __uint_as_float((h2 & 0x80000000u) | ((87u + (h2 >> 8) % 81u) << 23) | (h1 & 0x7fffffu));
      if ((h1 & 63u) == 0u) x = (h2 & 1u) ? 0.f : -0.f;
      if ((h2 & 63u) == 1u) y = (h1 & 1u) ? 0.f : -0.f;
    } else {
      const int k = (int)(h1 % 2001u) - 1000;
      const float eps = __builtin_exp2f(-10.f - 20.f * (float)(h2 & 0xffffu) * (1.f / 65536.f)) * ((h2 & 0x10000u) ? 1.f : -1.f);
      const float th = (float)k * angle + eps, rad = __builtin_exp2f(-6.6f + 13.2f * (float)(h1 >> 16) * (1.f / 65536.f));
      float sn, cs;
      dm::sincosf_(th, sn, cs);
      x = rad * cs; y = rad * sn;
    }
    float id;
    if (dm::circ_sector_fast(x, y, inv_angle, m, id)) {
      nf++;
      const float ref = dm::floorf_(dm::atan2_ref(y, x) / angle);
      if (__float_as_uint(id) != __float_as_uint(ref) && !(id == 0.f && ref == 0.f)) nb++;
    }
  }
  if (nb) atomicAdd(bad, nb);
  if (nf) atomicAdd(fast_count, nf);
}

// Self-test of dm::atan2_fast against dm::atan2_ref (float32 of Go's float64 math.Atan2): wherever the fast route ACCEPTS, its
// float32 must be the reference's, bit for bit. 2^log2n points per call:
//   mode 0  both coordinates from a hash: every sign, exponents 2^-60 .. 2^60 apart by up to 2^+-40, plus zeros of both signs
//   mode 1  searched towards rounding boundaries: a hashed pair, then of the 8 float32 neighbours y + j ulp the one whose float64
//           angle lies nearest to the midpoint of two float32 values (where a wrong accept would show first)
//   mode 2  lattice-shaped pairs, as a renderer produces them: x = ox + i res, y = oy + j res with small integers, |x| == |y|,
//           points on the axes, ratios near the octant boundaries
// bad: accepted and different; fast_count: accepted.
__global__ void __launch_bounds__(BLOCK) atan2_selftest_kernel(int mode, int log2n, unsigned long long* __restrict__ bad,
                                                               unsigned long long* __restrict__ fast_count) {
  unsigned long long nb = 0, nf = 0;
  auto hash = [](unsigned v) { v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16; return v; };
  const unsigned long long n = 1ull << log2n;
  for (unsigned long long i = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * BLOCK) {
    const unsigned h1 = hash((unsigned)i ^ (unsigned)(i >> 32) * 0x85ebca6bu), h2 = hash(h1 ^ 0x9e3779b9u), h3 = hash(h2 + 0x7f4a7c15u);
    float x, y;
    if (mode == 2) {
      const float res = __uint_as_float(0x3c000000u + (h3 & 0x03ffffffu)) ;  // 2^-7 .. 2^0 (a lattice step)
      const float ox = (float)((int)(h1 & 0xffffu) - 32768) * 1.52587890625e-05f * 40.f, oy = (float)((int)(h1 >> 16) - 32768) * 1.52587890625e-05f * 40.f;
      x = ox + res * (float)((int)(h2 & 0x7ffu) - 1024);
      y = oy + res * (float)((int)((h2 >> 11) & 0x7ffu) - 1024);
      const unsigned sel = h2 >> 28;
      if (sel == 0u) y = x;
      if (sel == 1u) y = -x;
      if (sel == 2u) y = (h3 >> 31) ? 0.f : -0.f;
      if (sel == 3u) x = (h3 >> 31) ? 0.f : -0.f;
      if (sel == 4u) y = x * 0.41421357f;   // tan(pi/8): the old routine's range boundary
      if (sel == 5u) y = x * 2.4142137f;
      if (sel == 6u) y = __uint_as_float(__float_as_uint(x) + ((h3 & 7u) - 3u));  // |y| within 4 ulp of |x|
    } else {
      // sign | exponent 67..187 | 23 mantissa bits for x; y's exponent within +-40 of x's; one in 64 is a zero
      const unsigned ex = 67u + (h1 >> 8) % 121u;
      int ey = (int)ex + (int)((h2 >> 8) % 81u) - 40;
      ey = ey < 1 ? 1 : (ey > 254 ? 254 : ey);
      x = __uint_as_float((h1 & 0x80000000u) | (ex << 23) | (h2 & 0x7fffffu));
      y = __uint_as_float((h2 & 0x80000000u) | ((unsigned)ey << 23) | (h3 & 0x7fffffu));
      if ((h1 & 63u) == 0u) x = (h2 & 1u) ? 0.f : -0.f;
      if ((h2 & 63u) == 1u) y = (h1 & 1u) ? 0.f : -0.f;
      if (mode == 1 && y != 0.f && x != 0.f) {
        float best = y;
        double bestd = 1.0;
        for (int j = -4; j < 4; j++) {
          const float yj = __uint_as_float(__float_as_uint(y) + (unsigned)j);
          const double g = dm::atan64((double)yj / (double)x);  // (the octant constant does not move the low bits' distance much; good enough to steer)
          const float f = (float)g;
          const float f2 = __uint_as_float(__float_as_uint(f) + (((double)f < g) ? 1u : 0xffffffffu));  // the float32 on the other side of g
          const double mid = 0.5 * ((double)f + (double)f2);
          const double d = __builtin_fabs(g - mid) / __builtin_fabs((double)f - (double)f2);
          if (d < bestd) { bestd = d; best = yj; }
        }
        y = best;
      }
    }
    bool ok;
    const float a = dm::atan2_fast(y, x, ok);
    if (ok) {
      nf++;
      if (__float_as_uint(a) != __float_as_uint(dm::atan2_ref(y, x))) nb++;
    }
  }
  if (nb) atomicAdd(bad, nb);
  if (nf) atomicAdd(fast_count, nf);
}

// Exhaustive self-test of dm::cossin_fast against dm::cossinf_ (float32 of Go's float64 math.Cos / math.Sin): every float32 bit pattern;
// wherever the short route accepts, both of its float32 results must be the reference's, bit for bit. bad: accepted and different;
// fast_count: accepted.
__global__ void __launch_bounds__(BLOCK) cossin_selftest_kernel(unsigned long long* __restrict__ bad, unsigned long long* __restrict__ fast_count) {
  unsigned long long nb = 0, nf = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x; i < (1ull << 32); i += (unsigned long long)gridDim.x * BLOCK) {
    const float x = __uint_as_float((unsigned)i);
    float c, s;
    bool ok;
    dm::cossin_fast(x, c, s, ok);
    if (!ok) continue;
    nf++;
    float cr, sr;
    dm::cossinf_(x, cr, sr);
    if (__float_as_uint(c) != __float_as_uint(cr) || __float_as_uint(s) != __float_as_uint(sr)) nb++;
  }
  if (nb) atomicAdd(bad, nb);
  if (nf) atomicAdd(fast_count, nf);
}

// Exhaustive self-test of dm::div_by_uniform: every float32 numerator against the IEEE division.
__global__ void __launch_bounds__(BLOCK) div_selftest_kernel(float d, float r, unsigned long long* __restrict__ bad,
                                                             unsigned long long* __restrict__ fast_count) {
  unsigned long long nb = 0, nf = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x; i < (1ull << 32); i += (unsigned long long)gridDim.x * BLOCK) {
    const float n = __uint_as_float((unsigned)i);
    if (!dm::div_fast_ok(n)) continue;  // the interpreter takes the IEEE path for these
    nf++;
    const float a = dm::div_by_uniform(n, d, r), b = n / d;
    if (__float_as_uint(a) != __float_as_uint(b)) nb++;
  }
  if (nb) atomicAdd(bad, nb);
  atomicAdd(fast_count, nf);
}

// gleval.NormalsCentralDiff (gleval/gleval.go:53-108); h = step/2.
__global__ void __launch_bounds__(BLOCK) normals_kernel(const uint32_t* __restrict__ code_g, const float* __restrict__ pos,
                                                        float* __restrict__ nrm, uint64_t n, float h) {
  code_ptr code = as_code(code_g);
  float* lds = g_smem + threadIdx.x;
  const uint64_t step = (uint64_t)gridDim.x * BLOCK;
  for (uint64_t base = (uint64_t)blockIdx.x * BLOCK; base < n; base += step) {
    const uint64_t i = base + threadIdx.x;
    const bool valid = i < n;
    float px = 0, py = 0, pz = 0;
    if (valid) { px = pos[3 * i]; py = pos[3 * i + 1]; pz = pos[3 * i + 2]; }
    float out[3];
#pragma unroll 1
    for (int dim = 0; dim < 3; dim++) {
      P3 a = {px + (dim == 0 ? h : 0.f), py + (dim == 1 ? h : 0.f), pz + (dim == 2 ? h : 0.f)};
      P3 b = {px - (dim == 0 ? h : 0.f), py - (dim == 1 ? h : 0.f), pz - (dim == 2 ? h : 0.f)};
      P3 ab[2] = {a, b};
      float dd[2];
      gsdf_dev::sdf_eval<2>(code, ab, dd, lds, BLOCK);
      const float v = dd[0] - dd[1];
      if (dim == 0) out[0] = v; else if (dim == 1) out[1] = v; else out[2] = v;
    }
    if (valid) { nrm[3 * i] = out[0]; nrm[3 * i + 1] = out[1]; nrm[3 * i + 2] = out[2]; }
  }
}
