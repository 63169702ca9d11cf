// What MI355X's HBM gives a plain read + write stream from a HIP kernel: the yardstick march_records_kernel is priced against
// (bench.py: roofline_march.frac_of_copy). The kernel reads `rbytes` and writes `wbytes` (they may differ: marching cubes reads
// 40-byte records and writes 36-byte triangles, 136 MB in and 245 MB out at npt-flange@1600), 16 bytes per lane and access, four
// accesses in flight per lane, grid-stride. Variants: plain, nontemporal stores, nontemporal loads + stores.
//   hipcc --offload-arch=gfx950 -O3 -DCOPY_RATE_MAIN tools/ubench/copy_rate.hip -o tools/ubench/copy_rate     (table on stdout)
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/ubench/copy_rate.hip -o tools/ubench/libcopyrate.so   (bench.py, ctypes)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int VARIANT>
__global__ __launch_bounds__(256) void copy_kernel(const f4* __restrict__ src, f4* __restrict__ dst, size_t nr, size_t nw, unsigned* sink) {
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    const size_t n = nr > nw ? nr : nw;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x; base < n; base += stride) {
        f4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            size_t i = base + (size_t)u * 256;
            v[u] = f4{1.f, 2.f, 3.f, 4.f};
            if (i < nr) v[u] = VARIANT >= 2 ? __builtin_nontemporal_load(src + i) : src[i];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            size_t i = base + (size_t)u * 256;
            if (i < nw) {
                if (VARIANT >= 1) __builtin_nontemporal_store(v[u], dst + i); else dst[i] = v[u];
            } else acc += v[u];  // (reads past the written range must still be consumed)
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345e30f) *sink = 1;
}

static void launch(int variant, const f4* s, f4* d, size_t nr, size_t nw, unsigned* sink, int blocks, hipStream_t st) {
    if (variant == 0) copy_kernel<0><<<blocks, 256, 0, st>>>(s, d, nr, nw, sink);
    else if (variant == 1) copy_kernel<1><<<blocks, 256, 0, st>>>(s, d, nr, nw, sink);
    else copy_kernel<2><<<blocks, 256, 0, st>>>(s, d, nr, nw, sink);
}

// average milliseconds per launch of `reps` back-to-back launches (HIP events on the launching stream); < 0 on error
extern "C" __attribute__((visibility("default"))) float copy_rate_ms(size_t rbytes, size_t wbytes, int variant, int reps, int blocks_per_cu) {
    size_t nr = rbytes / 16, nw = wbytes / 16;
    f4 *s = nullptr, *d = nullptr;
    unsigned* sink = nullptr;
    if (hipMalloc(&s, (nr + 1) * 16) != hipSuccess || hipMalloc(&d, (nw + 1) * 16) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return -1.f;
    (void)hipMemset(s, 0x11, nr * 16);
    (void)hipMemset(d, 0, nw * 16);
    hipStream_t st;
    (void)hipStreamCreate(&st);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    int blocks = 256 * blocks_per_cu;
    for (int i = 0; i < 3; ++i) launch(variant, s, d, nr, nw, sink, blocks, st);
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) launch(variant, s, d, nr, nw, sink, blocks, st);
    (void)hipEventRecord(e1, st);
    (void)hipStreamSynchronize(st);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipStreamDestroy(st);
    (void)hipFree(s);
    (void)hipFree(d);
    (void)hipFree(sink);
    return ms / reps;
}

#ifdef COPY_RATE_MAIN
int main(int argc, char** argv) {
    struct { const char* name; size_t r, w; } cases[] = {
        {"march_records @ npt-flange 1600 (136 MB in, 245 MB out)", 136u << 20, 245u << 20},
        {"equal copy, 256 MB + 256 MB", 256u << 20, 256u << 20},
        {"equal copy, 1 GB + 1 GB", 1u << 30, 1u << 30},
        {"write only, 245 MB", 0, 245u << 20},
        {"read only, 256 MB", 256u << 20, 0},
    };
    const char* vn[] = {"plain", "nt-store", "nt-load+store"};
    printf("%-58s %-14s %5s %9s %9s\n", "case", "variant", "wg/CU", "ms", "GB/s");
    for (auto& c : cases)
        for (int v = 0; v < 3; ++v)
            for (int bpc : {2, 4, 8, 16}) {
                float ms = copy_rate_ms(c.r, c.w, v, 50, bpc);
                printf("%-58s %-14s %5d %9.4f %9.1f\n", c.name, vn[v], bpc, ms, (c.r + c.w) / (ms * 1e-3) / 1e9);
            }
    return 0;
}
#endif
