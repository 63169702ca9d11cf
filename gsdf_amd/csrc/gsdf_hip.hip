// gsdf_hip.hip -- gfx950 kernels + C ABI (include/gsdf_hip.h) of the MI355X SDF backend.
//
// The kernels live in kernels.h (device code only, also compiled at run time by hiprtc for specialised programs).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/gsdf_hip.h"
#include "compile.h"
#include "specialize.h"

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIP_TRY(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t _e = (expr);                                                                             \
    if (_e != hipSuccess) return fail(GSDF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

extern "C" const char* gsdf_hip_last_error(void) { return g_err.c_str(); }

#include "kernels.h"

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct gsdf_program {
  gsdf_dev::Program prog;
  int device = 0;
  uint32_t* d_code = nullptr;
  hipStream_t stream = nullptr;
  std::atomic<uint64_t> evals{0};  // (atomic: host-buffer calls of several threads and a mesher may count at the same time)
  // staging for the host-buffer API
  void* d_pos = nullptr;
  float* d_dist = nullptr;
  size_t cap_pos_bytes = 0, cap_dist = 0;
  // pinned, device-mapped host staging for small host-buffer calls (the reference's callers hand over <= 32768 points)
  void* h_pos = nullptr;   // (slot 0 of the staging slots below; kept for the struct's older users)
  float* h_dist = nullptr;
  // Staging slots of the host-buffer API: pinned, device-mapped position / distance buffers with a stream each, so that
  // several host threads (glrender.FlatRenderer evaluates from numParallel goroutines, flatrenderer.go:120-129) -- or one
  // caller pipelining gsdf_hip_eval3_submit / gsdf_hip_eval_wait -- have calls in flight at the same time.
  struct Slot { void* h_pos = nullptr; float* h_dist = nullptr; hipStream_t s = nullptr; bool busy = false; bool waiting = false; unsigned gen = 0; float* user_dist = nullptr; size_t n = 0; bool zero_copy = false; };  // gen: bumped at every acquisition; a ticket is slot | gen << 8, so a stale or repeated wait is refused instead of releasing somebody else's call
  static constexpr int kSlots = 4;
  Slot slot[kSlots];
  std::mutex slot_mu;
  std::condition_variable slot_cv;
  std::atomic<uint64_t> evals_host{0};
  int num_cu = 256;
  // mesher workspace, grow-only, reused by every gsdf_hip_mesh_octree call on this handle
  struct Arena {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
      if (bytes <= cap) return hipSuccess;
      if (p) (void)hipFree(p);
      p = nullptr; cap = 0;
      size_t want = bytes + bytes / 4 + 4096;
      hipError_t e = hipMalloc(&p, want);
      if (e == hipSuccess) cap = want;
      return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  } q0, q1, ctr, spec_pass, dc_grid, dc_dist, dc_fv, dc_nrm, dc_edge, flat_grid, flat_bits, flat_list, rec, hdr;  // flat_bits: sign and near-surface bit planes of the flat renderer's lattice (flat_grid_kernel -> flat_cut_scan_kernel), flat_list: the cut cubes (-> flat_march_list_kernel)  // rec / hdr: cut-leaf records and block headers of the two-kernel leaf phase (the group sums follow the counters in ctr)  // dc_*: dual contouring workspace (index grid: 4 B per lattice cell); flat_grid: FlatRenderer distances
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  // The octree mesher leaves its counters cleared for the NEXT mesh (a memset enqueued behind the readback, executed while
  // the host is between calls) instead of clearing them at the head of its own chain, where the memset and the gap behind it
  // cost ~10 us of every mesh: ctr_clean = bytes of `ctr` known to be zero for work enqueued on ctr_clean_stream (0: unknown).
  size_t ctr_clean = 0;
  hipStream_t ctr_clean_stream = nullptr;
  void ctr_settle() {  // before anyone else writes `ctr`: the pending clear must have run (the stream may be a caller's, and gone)
    if (ctr_clean && hipStreamSynchronize(ctr_clean_stream) != hipSuccess) { (void)hipGetLastError(); (void)hipDeviceSynchronize(); }
    ctr_clean = 0;
  }
  void* h_ctr = nullptr;  // pinned host copy of the device counters (a pageable destination makes the D2H copy a staged, blocking one)
  uint64_t last_tris = 0;  // triangle count of the previous mesh on this handle: sizes the next output buffer
  uint64_t rec_blocks = 0;  // 64-leaf blocks the cut-leaf record arena is sized for (0: first mesh, start with kRecBlocks0)
  uint64_t last_dc_cubes = 0;  // kept cubes of the previous dual contouring pass: sizes the next queues
  // run-time specialised kernels for this program (gsdf_hip_program_specialize): hiprtc module, else interpreter
  hipModule_t spec_mod = nullptr, spec_mod2 = nullptr, spec_mod3 = nullptr, spec_mod4 = nullptr;  // spec_mod4: leaf kernel rebuilt with a larger register budget; spec_mod2: second group, built on first use (see spec_aux); spec_mod3: eval kernel rebuilt for 3 workgroups per CU
  hipFunction_t f_eval = nullptr, f_prune = nullptr, f_prune_spec = nullptr, f_leaf = nullptr;
  hipFunction_t f_dc_origin = nullptr, f_dc_edges = nullptr, f_dc_normals = nullptr, f_normals = nullptr, f_image = nullptr, f_flat_grid = nullptr;
  bool spec_aux_tried = false;
  int spec_eval_k = 0, spec_eval_w = 0, spec_leaf_k = 0, spec_leaf_w = 0;
  bool spec_leaf_both = false;  // the specialised leaf kernel has both column passes in one body (kernels_octree.h: BOTH)
  double spec_compile_s = 0;
  std::string spec_compiler;  // hipcc | hiprtc | cache: what built the specialised kernels
  std::string spec_key;       // key of that build (specialize.cpp: build_key)
  // leaf kernel batching: K = 4 while 3 workgroups still fit the CU's LDS (<= 11 slots); 12..15 slots run K = 2 at 4
  // waves/SIMD instead of K = 4 at 2 (knurled-cylinder: 23.3 vs 23.8 ms). A 4th wave per SIMD is worth more than the
  // ~30 VGPRs it costs (flange 3.28 -> 2.95 ms), but only if 4 workgroups fit the CU's 160 KB of LDS.
  void leaf_config(int* k, int* w, size_t* lds) const;
  bool leaf_nt_in_lds() const {  // see leaf_config
    const int ns = prog.nslots > 0 ? prog.nslots : 1;
    const int lk = (batch_k() == 4 && ns > 11) ? 2 : batch_k();
    const size_t rows_e = (size_t)(ns * lk > 8 ? ns * lk : 8) * BLOCK * sizeof(float);
    return !(4 * rows_e <= 160 * 1024 && 4 * (rows_e + 256) > 160 * 1024);
  }
  size_t lds_bytes(int k = 1) const { return (size_t)(prog.nslots > 0 ? prog.nslots : 1) * k * BLOCK * sizeof(float); }
  // Workgroups per CU the lattice/eval sweeps are compiled for (their W template argument): 4 when the LDS allows it.
  int sweep_waves(int k) const {
    static const int forced = [] { const char* e = getenv("GSDF_HIP_SWEEP_WAVES"); return e ? atoi(e) : 0; }();  // tuning / debugging knob
    if (k != 1 && (forced == 3 || forced == 4)) return forced;
    return (k == 1 || 4 * (lds_bytes(k) + 64) <= (size_t)160 * 1024) ? 4 : 3;
  }
  // Points carried per lane: as many as keep >= 2 workgroups per CU resident (160 KB LDS per CU).
  int batch_k() const {
    static const int forced = [] { const char* e = getenv("GSDF_HIP_BATCH_K"); return e ? atoi(e) : 0; }();  // tuning knob
    if (forced == 1 || forced == 2 || forced == 4) return forced;
    return prog.nslots <= 12 ? 4 : (prog.nslots <= 28 ? 2 : 1);
  }
};

// Leaf phase: two kernels by default (leaf_eval_kernel: evaluation + cut-leaf records, no barrier and no atomic in its
// loop; march_records_kernel: marching cubes over the records); GSDF_HIP_FUSED_LEAF=1 keeps the fused leaf_kernel.
static bool fused_leaf() {
  static const bool f = [] { const char* e = getenv("GSDF_HIP_FUSED_LEAF"); return e && atoi(e) != 0; }();
  return f;
}

void gsdf_program::leaf_config(int* k, int* w, size_t* lds) const {
  if (!fused_leaf()) {  // the evaluating kernel needs the interpreter's columns only
    static const int forced_w = [] { const char* e = getenv("GSDF_HIP_LEAF_WAVES"); return e ? atoi(e) : 0; }();  // tuning knob
    const int ns = prog.nslots > 0 ? prog.nslots : 1;
    const int lk = (batch_k() == 4 && ns > 11) ? 2 : batch_k();
    // (8 rows at least: a brick's distances) + the 256-byte triangles-per-case table, unless it is exactly that table which
    // would cost the fourth workgroup per CU (40 rows = 40 KB): the kernel then reads the counts from the table in global memory
    const size_t rows_e = (size_t)(ns * lk > 8 ? ns * lk : 8) * BLOCK * sizeof(float);
    const size_t lds_e = rows_e + (leaf_nt_in_lds() ? 256 : 0);
    int ww = forced_w ? forced_w : (4 * lds_e <= 160 * 1024 ? 4 : 3);
    if (lk == 4) { if (ww != 2 && ww != 4 && !(ww == 5 && 5 * lds_e <= 160 * 1024)) ww = 3; }
    else if (lk == 2) { if (ww != 4) ww = 3; }
    else ww = 4;
    *k = lk; *w = ww; *lds = lds_e;
    return;
  }
  static const int forced_w = [] { const char* e = getenv("GSDF_HIP_LEAF_WAVES"); return e ? atoi(e) : 0; }();  // tuning knob
  const int ns = prog.nslots;
  const int lk = (batch_k() == 4 && ns > 11) ? 2 : batch_k();
  const size_t lds_m = (size_t)(ns * lk > LEAF_MIN_COLS ? ns * lk : LEAF_MIN_COLS) * BLOCK * sizeof(float) + 4096 + TRI_STAGE * 36 + 64;
  int ww = forced_w ? forced_w : (4 * lds_m <= 160 * 1024 ? 4 : 3);
  if (lk == 4) { if (ww != 2 && ww != 4) ww = 3; }
  else if (lk == 2) { if (ww != 4) ww = 3; }
  else ww = 4;
  *k = lk; *w = ww; *lds = lds_m;
}

// hipModuleLaunchKernel with typed arguments (the specialised kernels take exactly the ahead-of-time kernels' parameters)
template <typename... A>
static hipError_t launch_fn(hipFunction_t f, unsigned grid, unsigned block, size_t lds, hipStream_t s, A... a) {
  void* args[] = {(void*)&a...};
  return hipModuleLaunchKernel(f, grid, 1, 1, block, 1, 1, (unsigned)lds, s, args, nullptr);
}

struct gsdf_mesh {
  int device = 0;
  float* d_tris = nullptr;
  uint64_t cap = 0;
  gsdf_mesh_stats st{};
  hipStream_t stream = nullptr;   // the stream the mesher ran on (the program's or the caller's); the mesher has synchronised it
  hipStream_t rstream = nullptr;  // the mesh's OWN stream for later reads / STL builds: a mesh may outlive its program handle
  bool host_out = false;  // d_tris is pinned, device-mapped HOST memory (gsdf_mesh_opts.host_output): the kernels write across PCIe
  // pinned host copies handed out by gsdf_hip_mesh_host_tris / gsdf_hip_mesh_host_stl (owned by the mesh)
  void* h_tris = nullptr;
  size_t h_tris_cap = 0;
  void* h_stl = nullptr;
  size_t h_stl_cap = 0;
};

// Triangle buffers are recycled through a small per-process pool: hipMalloc/hipFree of the multi-GB
// output buffer would otherwise dominate a mesh call.
namespace {
struct TriBuf { int device; float* p; uint64_t cap; };
std::mutex g_pool_mu;
std::vector<TriBuf> g_pool;
float* pool_take(int device, uint64_t need, uint64_t* cap_out) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  int best = -1;
  for (size_t i = 0; i < g_pool.size(); i++)
    if (g_pool[i].device == device && g_pool[i].cap >= need && (best < 0 || g_pool[i].cap < g_pool[(size_t)best].cap)) best = (int)i;
  if (best < 0) return nullptr;
  TriBuf b = g_pool[(size_t)best];
  g_pool.erase(g_pool.begin() + best);
  *cap_out = b.cap;
  return b.p;
}
// Pinned host buffers for the zero-copy result views: pinning a few hundred MB costs more than the transfer it serves.
struct HostBuf { void* p; size_t cap; };
std::vector<HostBuf> g_hpool;
void* hpool_take(size_t need, size_t* cap_out) {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  int best = -1;
  for (size_t i = 0; i < g_hpool.size(); i++)
    if (g_hpool[i].cap >= need && (best < 0 || g_hpool[i].cap < g_hpool[(size_t)best].cap)) best = (int)i;
  if (best < 0) return nullptr;
  HostBuf b = g_hpool[(size_t)best];
  g_hpool.erase(g_hpool.begin() + best);
  *cap_out = b.cap;
  return b.p;
}
void hpool_give(void* p, size_t cap) {
  if (!p) return;
  std::lock_guard<std::mutex> lk(g_pool_mu);
  if (g_hpool.size() >= 4) {  // drop the smallest
    size_t sm = 0;
    for (size_t i = 1; i < g_hpool.size(); i++) if (g_hpool[i].cap < g_hpool[sm].cap) sm = i;
    if (g_hpool[sm].cap < cap) { (void)hipHostFree(g_hpool[sm].p); g_hpool[sm] = HostBuf{p, cap}; }
    else (void)hipHostFree(p);
    return;
  }
  g_hpool.push_back(HostBuf{p, cap});
}
void pool_give(int device, float* p, uint64_t cap) {
  if (!p) return;
  std::lock_guard<std::mutex> lk(g_pool_mu);
  if (g_pool.size() >= 4) {  // drop the smallest
    size_t sm = 0;
    for (size_t i = 1; i < g_pool.size(); i++) if (g_pool[i].cap < g_pool[sm].cap) sm = i;
    if (g_pool[sm].cap < cap) { (void)hipFree(g_pool[sm].p); g_pool[sm] = TriBuf{device, p, cap}; }
    else (void)hipFree(p);
    return;
  }
  g_pool.push_back(TriBuf{device, p, cap});
}
}  // namespace

// memcpy out of pinned memory into a caller's (usually freshly allocated, not yet faulted-in) buffer: one thread moves
// ~10 GB/s and takes every page fault itself; large results are split over a few threads.
static void big_memcpy(void* dst, const void* src, size_t n) {
  constexpr size_t kChunk = (size_t)16 << 20;
  unsigned nt = (unsigned)std::min<size_t>(8, n / kChunk);
  const unsigned hw = std::thread::hardware_concurrency();
  if (hw && nt > hw) nt = hw;
  if (nt < 2) { std::memcpy(dst, src, n); return; }
  std::vector<std::thread> th;
  const size_t per = ((n / nt) + 4095) & ~(size_t)4095;
  for (unsigned i = 0; i < nt; i++) {
    const size_t off = (size_t)i * per;
    if (off >= n) break;
    const size_t len = std::min(per, n - off);
    th.emplace_back([=] { std::memcpy((char*)dst + off, (const char*)src + off, len); });
  }
  for (auto& t : th) t.join();
}

static int host_buf(void** buf, size_t* cap, size_t need);
// Stream for work issued on a finished mesh (result copies, stl_kernel): created on first use and destroyed with the mesh,
// so that reading a mesh after gsdf_hip_program_destroy (finalisers / garbage collectors run in any order) never touches
// the program's destroyed stream. nullptr (the null stream) if a stream cannot be had.
static hipStream_t mesh_stream(gsdf_mesh* m) {
  if (!m->rstream && hipStreamCreateWithFlags(&m->rstream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); m->rstream = nullptr; }
  return m->rstream;
}
static void release_tris(gsdf_mesh* m) {
  if (m->host_out) hpool_give(m->d_tris, (size_t)m->cap * 36);
  else pool_give(m->device, m->d_tris, m->cap);
  m->d_tris = nullptr; m->cap = 0; m->host_out = false;
}

static unsigned grid_for(uint64_t n, int num_cu, int blocks_per_cu) {
  uint64_t b = (n + BLOCK - 1) / BLOCK;
  uint64_t mx = (uint64_t)num_cu * (uint64_t)blocks_per_cu;
  if (b > mx) b = mx;
  if (b < 1) b = 1;
  return (unsigned)b;
}

extern "C" int gsdf_hip_init(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) return fail(GSDF_ERR_NO_DEVICE, std::string("no HIP device: ") + hipGetErrorString(e));
  if (device >= 0) {
    if (device >= n) return fail(GSDF_ERR_NO_DEVICE, "device index out of range");
    HIP_TRY(hipSetDevice(device));
  }
  return GSDF_OK;
}

extern "C" int gsdf_hip_program_create(const gsdf_tree* tree, gsdf_program** out) {
  if (!tree || !out) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  *out = nullptr;
  gsdf_program* p = new (std::nothrow) gsdf_program();
  if (!p) return fail(GSDF_ERR_BAD_ARGUMENT, "out of memory");
  try {
    p->prog = gsdf_dev::compile(*tree);
  } catch (const std::exception& e) {
    delete p;
    return fail(GSDF_ERR_BAD_TREE, e.what());
  }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    delete p;
    return fail(GSDF_ERR_NO_DEVICE, std::string("no HIP device: ") + hipGetErrorString(e));
  }
  auto cleanup = [&](int code) { gsdf_hip_program_destroy(p); return code; };
  if (hipGetDevice(&p->device) != hipSuccess) return cleanup(fail(GSDF_ERR_HIP, "hipGetDevice failed"));
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, p->device) == hipSuccess) p->num_cu = prop.multiProcessorCount;
  if (p->lds_bytes(p->batch_k()) + 8 * BLOCK * 4 + 4096 + TRI_STAGE * 36 + 64 > 160 * 1024) return cleanup(fail(GSDF_ERR_BAD_TREE, "tree needs more LDS scratch than one CU has"));
  if (hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) != hipSuccess) return cleanup(fail(GSDF_ERR_HIP, "hipStreamCreate failed"));
  size_t bytes = p->prog.code.size() * sizeof(uint32_t);
  if (hipMalloc((void**)&p->d_code, bytes) != hipSuccess) return cleanup(fail(GSDF_ERR_HIP, "hipMalloc(program) failed"));
  if (hipMemcpy(p->d_code, p->prog.code.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return cleanup(fail(GSDF_ERR_HIP, "hipMemcpy(program) failed"));
  *out = p;
  return GSDF_OK;
}

// Test hook: runs dm::div_by_uniform(n, d, RN(1/d)) for all 2^32 numerators n on the GPU and counts results that
// differ from n / d among the numerators the interpreter would send down the fast path.
extern "C" int gsdf_hip_selftest_div(float d, uint64_t* mismatches, uint64_t* fast_path_numerators, float* recip) {
  const float r = gsdf_dev::recip_for(d);
  if (recip) *recip = r;
  if (mismatches) *mismatches = 0;
  if (fast_path_numerators) *fast_path_numerators = 0;
  if (r == 0.f) return GSDF_OK;  // divisor not eligible: the device always uses the IEEE expansion
  unsigned long long* d_c = nullptr;
  HIP_TRY(hipMalloc((void**)&d_c, 16));
  int rc = GSDF_OK;
  do {
    if (hipMemset(d_c, 0, 16) != hipSuccess) { rc = fail(GSDF_ERR_HIP, "memset failed"); break; }
    hipLaunchKernelGGL(div_selftest_kernel, dim3(4096), dim3(BLOCK), 0, nullptr, d, r, d_c, d_c + 1);
    unsigned long long h[2] = {0, 0};
    if (hipMemcpy(h, d_c, 16, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(GSDF_ERR_HIP, "selftest kernel failed"); break; }
    if (mismatches) *mismatches = h[0];
    if (fast_path_numerators) *fast_path_numerators = h[1];
  } while (0);
  (void)hipFree(d_c);
  return rc;
}

// Test hook: dm::circ_sector_fast (the circular array's sector index without the angle) against the reference's
// floor(float32(atan2(y, x)) / angle) over 2^32 points -- all magnitudes, signed zeros, and points hugging the sector
// boundaries -- for angle = float32(2 pi) / ncirc as circarray forms it (cpu_evaluators.go:1047).
extern "C" int gsdf_hip_selftest_circ(float ncirc, uint64_t* mismatches, uint64_t* fast_path_points) {
  if (mismatches) *mismatches = 0;
  if (fast_path_points) *fast_path_points = 0;
  if (!(ncirc >= 1.f)) return fail(GSDF_ERR_BAD_ARGUMENT, "ncirc must be >= 1");
  const float angle = 6.2831853071795862f / ncirc;
  unsigned long long* d_c = nullptr;
  HIP_TRY(hipMalloc((void**)&d_c, 16));
  int rc = GSDF_OK;
  do {
    if (hipMemset(d_c, 0, 16) != hipSuccess) { rc = fail(GSDF_ERR_HIP, "memset failed"); break; }
    hipLaunchKernelGGL(circ_selftest_kernel, dim3(4096), dim3(BLOCK), 0, nullptr, angle, d_c, d_c + 1);
    unsigned long long h[2] = {0, 0};
    if (hipMemcpy(h, d_c, 16, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(GSDF_ERR_HIP, "selftest kernel failed"); break; }
    if (mismatches) *mismatches = h[0];
    if (fast_path_points) *fast_path_points = h[1];
  } while (0);
  (void)hipFree(d_c);
  return rc;
}

// Test hook: dm::sqrt_1to2 against sqrtf for all 8,388,609 floats in [1, 2].
extern "C" int gsdf_hip_selftest_sqrt(uint64_t* mismatches) {
  unsigned long long* d_c = nullptr;
  HIP_TRY(hipMalloc((void**)&d_c, 8));
  int rc = GSDF_OK;
  unsigned long long h = 0;
  if (hipMemset(d_c, 0, 8) != hipSuccess) rc = fail(GSDF_ERR_HIP, "memset failed");
  if (!rc) {
    hipLaunchKernelGGL(sqrt_selftest_kernel, dim3(1024), dim3(BLOCK), 0, nullptr, d_c);
    if (hipMemcpy(&h, d_c, 8, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(GSDF_ERR_HIP, "selftest kernel failed");
  }
  (void)hipFree(d_c);
  if (mismatches) *mismatches = h;
  return rc;
}

static std::string device_arch(int device) {
  hipDeviceProp_t pr;
  if (hipGetDeviceProperties(&pr, device) != hipSuccess) return "gfx950";
  std::string arch = pr.gcnArchName;
  if (arch.find(':') != std::string::npos) arch = arch.substr(0, arch.find(':'));
  return arch;
}

// hiprtc build + module load of `names` for the handle's program; fns receives one function per name.
static int spec_build(gsdf_program* p, const std::vector<std::string>& names, hipModule_t* mod_out, std::vector<hipFunction_t>& fns, double* secs) {
  std::vector<char> co;
  std::vector<std::string> low;
  std::string log;
  const auto t0 = std::chrono::steady_clock::now();
  if (!gsdf_dev::spec_compile(p->prog, device_arch(p->device), names, co, low, log)) return fail(GSDF_ERR_HIP, "specialised build failed:\n" + log);
  if (secs) *secs += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  hipModule_t mod = nullptr;
  HIP_TRY(hipModuleLoadData(&mod, co.data()));
  fns.assign(names.size(), nullptr);
  for (size_t i = 0; i < names.size(); i++) {
    hipError_t e = hipModuleGetFunction(&fns[i], mod, low[i].c_str());
    if (e != hipSuccess) {
      (void)hipModuleUnload(mod);
      return fail(GSDF_ERR_HIP, std::string("hipModuleGetFunction: ") + hipGetErrorString(e));
    }
  }
  *mod_out = mod;
  return GSDF_OK;
}

// Scratch (private segment) bytes per lane of a built kernel. A specialised kernel is used only if this is 0: its code
// shape is new for every tree, and a build that spills registers inside divergent regions has been seen to lose the
// spilled values of the lanes that were inactive at the spill (2-D fuzz tree 708: eval_kernel<2,4,4>, 128 VGPRs + 132 B
// of scratch, wrote the results of 216 points of a ragged last tile to the wrong addresses). The ahead-of-time
// interpreter kernels have ONE code shape each, and that shape is what the whole test suite runs.
static int fn_scratch_bytes(hipFunction_t f) {
  static const bool allow = getenv("GSDF_HIP_EXP_ALLOW_SCRATCH") != nullptr;  // developer experiments only: timing of a spilling build
  if (allow) return 0;
  int v = 0;
  if (hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, f) != hipSuccess) { (void)hipGetLastError(); return 1 << 30; }
  return v;
}
static void spec_report(const char* what, const std::string& name, hipFunction_t f, bool used) {
  if (!getenv("GSDF_HIP_DEBUG")) return;
  int regs = -1;
  (void)hipFuncGetAttribute(&regs, HIP_FUNC_ATTRIBUTE_NUM_REGS, f);
  fprintf(stderr, "gsdf_hip: %s %s: %d registers, %d B scratch per lane -> %s\n", what, name.c_str(), regs, fn_scratch_bytes(f),
          used ? "used" : "not used (interpreter kernel instead)");
}

// Second group of a specialised handle, built the first time one of these entry points runs: the evaluating kernels of
// dual contouring, central-difference normals, the flat renderer's lattice pass and the 2-D image renderer. Failure leaves the interpreter kernels in use.
static void spec_aux(gsdf_program* p) {
  if (!p->spec_mod || p->spec_aux_tried) return;
  p->spec_aux_tried = true;
  const std::string k = std::to_string(p->batch_k());
  const std::string kw = k + ", " + std::to_string(p->sweep_waves(p->batch_k()));
  std::vector<hipFunction_t> f;
  if (p->prog.is2d) {
    if (spec_build(p, {"image2_kernel<" + k + ">"}, &p->spec_mod2, f, &p->spec_compile_s) == GSDF_OK) {
      const bool ok = fn_scratch_bytes(f[0]) == 0;
      spec_report("specialised", "image2_kernel", f[0], ok);
      p->f_image = ok ? f[0] : nullptr;
    }
  } else {
    if (spec_build(p, {"dc_origin_kernel<" + kw + ">", "dc_edges_kernel", "dc_normals_kernel", "normals_kernel", "flat_grid_kernel<" + kw + ">"},
                   &p->spec_mod2, f, &p->spec_compile_s) == GSDF_OK) {
      const char* nm[5] = {"dc_origin_kernel", "dc_edges_kernel", "dc_normals_kernel", "normals_kernel", "flat_grid_kernel"};
      hipFunction_t* dst[5] = {&p->f_dc_origin, &p->f_dc_edges, &p->f_dc_normals, &p->f_normals, &p->f_flat_grid};
      for (int i = 0; i < 5; i++) {
        const bool ok = fn_scratch_bytes(f[(size_t)i]) == 0;
        spec_report("specialised", nm[i], f[(size_t)i], ok);
        *dst[i] = ok ? f[(size_t)i] : nullptr;
      }
    }
  }
}

// Compile and load kernels specialised for this handle's program (specialize.cpp): eval, prune and leaf kernels of
// the configuration the mesher would pick. Afterwards gsdf_hip_eval*/gsdf_hip_mesh_octree launch them instead of the
// interpreter kernels; results are bit-identical (same statements, same compiler flags). Idempotent.
extern "C" int gsdf_hip_program_specialize(gsdf_program* p) {
  if (!p) return fail(GSDF_ERR_BAD_ARGUMENT, "null program");
  if (p->spec_mod) return GSDF_OK;
  // straight-line code grows with the program (multi-evaluation nodes are unrolled at lowering time): beyond a few
  // thousand instructions the build takes minutes and the code no longer fits the instruction cache
  if (gsdf_dev::spec_instruction_count(p->prog) > 4000)
    return fail(GSDF_ERR_BAD_TREE, "program too large to specialise (more than 4000 instructions): the interpreter kernels stay in use");
  HIP_TRY(hipSetDevice(p->device));
  int lk, lw;
  size_t lds_m;
  p->leaf_config(&lk, &lw, &lds_m);
  const int ek = p->batch_k();
  std::vector<std::string> names;
  int both_at = -1;
  const int ew = p->sweep_waves(ek);
  names.push_back(std::string("eval_kernel<") + (p->prog.is2d ? "2" : "3") + ", " + std::to_string(ek) + ", " + std::to_string(ew) + ">");
  if (!p->prog.is2d) {
    names.push_back("prune_kernel");
    names.push_back("prune_spec_kernel");
    names.push_back(std::string(fused_leaf() ? "leaf_kernel<" : "leaf_eval_kernel<") + std::to_string(lk) + ", " + std::to_string(lw) + (fused_leaf() ? ">" : (p->leaf_nt_in_lds() ? ", true, true, false>" : ", true, false, false>")));
    // a fifth workgroup per CU where the LDS has room for it (npt-flange's 7 slots): the 96-register build is taken if the
    // compiler reaches it without scratch (-3 % on the evaluating kernel); built beside the 128-register one, same process
    if (!fused_leaf() && lk == 4 && lw == 4 && 5 * lds_m <= (size_t)160 * 1024) names.push_back("leaf_eval_kernel<4, 5, true, true, false>");
    // both passes of a column brick in one body -- taken, ahead of the others, if the compiler reaches it without scratch: -7 %
    // where much of the program depends on x and y alone (an atan2, several hypots: npt-flange), -1..2 % elsewhere
    static const bool both_off = [] { const char* e = getenv("GSDF_HIP_NO_BOTH_PASSES"); return e && atoi(e) != 0; }();  // developer knob (A/B timing)
    static const int both_min = [] { const char* e = getenv("GSDF_HIP_BOTH_MIN_WEIGHT"); return e ? atoi(e) : 0; }();  // developer knob. 0: always -- programs without x,y-only work gain 1-2 % too (bolt 1.029 -> 1.007 ms, knurled-cylinder 3.19 -> 3.16: one body of eight points schedules a little better than two of four)
    if (!fused_leaf() && !both_off && lk == 4 && gsdf_dev::spec_xy_shared_weight(p->prog) >= both_min) {
      both_at = (int)names.size();
      names.push_back(std::string("leaf_eval_kernel<4, ") + std::to_string(lw) + (p->leaf_nt_in_lds() ? ", true, true, true>" : ", true, false, true>"));
    }
  }
  std::vector<hipFunction_t> f;
  hipModule_t mod = nullptr;
  const int rc = spec_build(p, names, &mod, f, &p->spec_compile_s);
  if (rc != GSDF_OK) return rc;
  p->spec_mod = mod;
  p->spec_compiler = gsdf_dev::spec_last_compiler();
  p->spec_key = gsdf_dev::spec_last_key();
  p->spec_eval_k = ek; p->spec_eval_w = ew; p->spec_leaf_k = lk; p->spec_leaf_w = lw;
  // No scratch, or not used (see fn_scratch_bytes). The eval kernel gets a second chance with the larger register
  // budget of 3 workgroups per CU before the handle falls back to the interpreter kernel for that entry point.
  {
    bool ok = fn_scratch_bytes(f[0]) == 0;
    spec_report("specialised", names[0], f[0], ok);
    p->f_eval = ok ? f[0] : nullptr;
    if (!ok && ew == 4) {
      std::vector<hipFunction_t> f3;
      const std::string n3 = std::string("eval_kernel<") + (p->prog.is2d ? "2" : "3") + ", " + std::to_string(ek) + ", 3>";
      if (spec_build(p, {n3}, &p->spec_mod3, f3, &p->spec_compile_s) == GSDF_OK) {
        ok = fn_scratch_bytes(f3[0]) == 0;
        spec_report("specialised", n3, f3[0], ok);
        if (ok) { p->f_eval = f3[0]; p->spec_eval_w = 3; }
      }
    }
  }
  if (!p->prog.is2d) {
    const bool okp = fn_scratch_bytes(f[1]) == 0, okt = fn_scratch_bytes(f[2]) == 0;
    bool okl = fn_scratch_bytes(f[3]) == 0;
    spec_report("specialised", names[1], f[1], okp);
    spec_report("specialised", names[2], f[2], okt);
    spec_report("specialised", names[3], f[3], okl);
    p->f_prune = okp ? f[1] : nullptr;
    p->f_prune_spec = okt ? f[2] : nullptr;
    p->f_leaf = okl ? f[3] : nullptr;
    if (f.size() > 4 && both_at != 4) {
      const bool ok5 = fn_scratch_bytes(f[4]) == 0;
      spec_report("specialised", names[4], f[4], ok5);
      if (ok5) { p->f_leaf = f[4]; p->spec_leaf_w = 5; okl = true; }
    }
    if (both_at >= 0) {
      const bool okb = fn_scratch_bytes(f[(size_t)both_at]) == 0;
      spec_report("specialised", names[(size_t)both_at], f[(size_t)both_at], okb);
      if (okb) { p->f_leaf = f[(size_t)both_at]; p->spec_leaf_w = lw; p->spec_leaf_both = true; okl = true; }
    }
    // the leaf kernel is where the time goes: before giving it up, trade occupancy for registers (W = workgroups per CU
    // the register budget is sized for; the launch is the same)
    for (int w2 = lw - 1; !okl && w2 >= 2; w2--) {
      std::vector<hipFunction_t> fl;
      hipModule_t m2 = nullptr;
      const std::string nl = std::string(fused_leaf() ? "leaf_kernel<" : "leaf_eval_kernel<") + std::to_string(lk) + ", " + std::to_string(w2) + (fused_leaf() ? ">" : (p->leaf_nt_in_lds() ? ", true, true, false>" : ", true, false, false>"));
      if (spec_build(p, {nl}, &m2, fl, &p->spec_compile_s) != GSDF_OK) break;
      okl = fn_scratch_bytes(fl[0]) == 0;
      spec_report("specialised", nl, fl[0], okl);
      if (okl) { p->f_leaf = fl[0]; p->spec_leaf_w = w2; p->spec_mod4 = m2; }
      else (void)hipModuleUnload(m2);
    }
  }
  return GSDF_OK;
}
/* 1 if the handle runs specialised kernels; compile_seconds (optional) = what the build took */
extern "C" int gsdf_hip_program_is_specialized(const gsdf_program* p, double* compile_seconds) {
  if (compile_seconds) *compile_seconds = p ? p->spec_compile_s : 0.0;
  return p && p->spec_mod ? 1 : 0;
}

/* The kernels this handle launches, e.g. "eval=eval_kernel<3,4,4>:specialised leaf=leaf_kernel<4,3>:specialised
 * prune=prune_kernel:specialised" (":interpreter" for the ahead-of-time kernels): what a profile of the handle shows. */
extern "C" int gsdf_hip_program_kernels(const gsdf_program* p, char* dst, size_t dst_cap) {
  if (!p || !dst || dst_cap == 0) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  int lk, lw;
  size_t lds_m;
  p->leaf_config(&lk, &lw, &lds_m);
  const int ek = p->batch_k();
  const bool se = p->f_eval && p->spec_eval_k == ek, sl = p->f_leaf && p->spec_leaf_k == lk;
  const int ew = se ? p->spec_eval_w : p->sweep_waves(ek);
  // ahead-of-time leaf kernels exist at the scratch-free occupancies only (see gsdf_hip_mesh_octree)
  const int aw = lk == 4 ? (lw == 2 ? 2 : 3) : (lk == 2 ? 3 : 4);
  char buf[384];
  if (p->prog.is2d)
    snprintf(buf, sizeof buf, "eval=eval_kernel<2,%d,%d>:%s", ek, ew, se ? "specialised" : "interpreter");
  else
    snprintf(buf, sizeof buf, "eval=eval_kernel<3,%d,%d>:%s leaf=%s<%d,%d%s>:%s prune=prune_kernel:%s", ek, ew, se ? "specialised" : "interpreter",
             fused_leaf() ? "leaf_kernel" : "leaf_eval_kernel", lk, sl ? p->spec_leaf_w : aw, sl && p->spec_leaf_both ? ",both" : "", sl ? "specialised" : "interpreter",
             p->f_prune ? "specialised" : "interpreter");
  if (p->spec_mod && strlen(buf) + 32 < sizeof buf) { strcat(buf, " compiler="); strcat(buf, p->spec_compiler.c_str()); }
  {  // identity of the code that runs: a stored profile describes this handle's kernels only if it carries the same key
    const std::string key = p->spec_mod ? p->spec_key : gsdf_dev::spec_library_key();
    if (strlen(buf) + 8 + key.size() < sizeof buf) { strcat(buf, " code="); strcat(buf, key.c_str()); }
  }
  if (strlen(buf) + 1 > dst_cap) return fail(GSDF_ERR_SHORT_BUFFER, "short buffer");
  std::memcpy(dst, buf, strlen(buf) + 1);
  return GSDF_OK;
}

// Host-only (no GPU): the generated evaluator source of a tree's specialised build, and a hiprtc compile of the
// specialised kernels for gfx950 that stops before loading them (proves the generated code builds).
extern "C" int gsdf_hip_specialize_source(const gsdf_tree* tree, char* dst, size_t dst_cap, size_t* len) {
  if (!tree) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  try {
    const std::string s = gsdf_dev::spec_source(gsdf_dev::compile(*tree));
    if (len) *len = s.size();
    if (dst) {
      if (dst_cap < s.size() + 1) return fail(GSDF_ERR_SHORT_BUFFER, "short buffer");
      std::memcpy(dst, s.c_str(), s.size() + 1);
    }
    return GSDF_OK;
  } catch (const std::exception& e) {
    return fail(GSDF_ERR_BAD_TREE, e.what());
  }
}
extern "C" int gsdf_hip_specialize_check(const gsdf_tree* tree, size_t* code_object_bytes) {
  if (!tree) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  try {
    const gsdf_dev::Program pr = gsdf_dev::compile(*tree);
    std::vector<char> co;
    std::vector<std::string> low;
    std::string log;
    const std::vector<std::string> names = pr.is2d ? std::vector<std::string>{"eval_kernel<2, 4, 4>"}
                                                   : std::vector<std::string>{"eval_kernel<3, 4, 4>", "prune_kernel", "prune_spec_kernel", "leaf_eval_kernel<4, 4, true, true, false>", "leaf_kernel<4, 4>", "flat_grid_kernel<4, 4>"};
    if (!gsdf_dev::spec_compile(pr, "gfx950", names, co, low, log)) return fail(GSDF_ERR_HIP, "specialised build failed:\n" + log);
    if (code_object_bytes) *code_object_bytes = co.size();
    return GSDF_OK;
  } catch (const std::exception& e) {
    return fail(GSDF_ERR_BAD_TREE, e.what());
  }
}

// Host-only (no GPU): lower a tree to the device instruction stream, for inspection/tests.
extern "C" int gsdf_hip_lower(const gsdf_tree* tree, uint32_t* code_out, uint32_t code_cap, uint32_t* code_words, uint32_t* lds_slots) {
  if (!tree) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  try {
    gsdf_dev::Program pr = gsdf_dev::compile(*tree);
    if (code_words) *code_words = (uint32_t)pr.code.size();
    if (lds_slots) *lds_slots = (uint32_t)pr.nslots;
    if (code_out) {
      if (code_cap < pr.code.size()) return fail(GSDF_ERR_SHORT_BUFFER, "short buffer");
      std::memcpy(code_out, pr.code.data(), pr.code.size() * sizeof(uint32_t));
    }
    return GSDF_OK;
  } catch (const std::exception& e) {
    return fail(GSDF_ERR_BAD_TREE, e.what());
  }
}

// Host-only test hook: the lower-bound region the compiler claims for the subtree of `node` (0 none, 1 box, 2 z-cylinder).
extern "C" int gsdf_hip_lower_region(const gsdf_tree* tree, uint32_t node, int* kind, float params[8]) {
  if (!tree || !kind || !params) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  try {
    *kind = gsdf_dev::region_of(*tree, node, params);
    return GSDF_OK;
  } catch (const std::exception& e) {
    return fail(GSDF_ERR_BAD_TREE, e.what());
  }
}

extern "C" void gsdf_hip_program_destroy(gsdf_program* p) {
  if (!p) return;
  if (p->d_code) (void)hipFree(p->d_code);
  if (p->d_pos) (void)hipFree(p->d_pos);
  if (p->d_dist) (void)hipFree(p->d_dist);
  if (p->h_pos) (void)hipHostFree(p->h_pos);
  if (p->h_dist) (void)hipHostFree(p->h_dist);
  for (auto& sl : p->slot) {
    if (sl.h_pos) (void)hipHostFree(sl.h_pos);
    if (sl.h_dist) (void)hipHostFree(sl.h_dist);
    if (sl.s) (void)hipStreamDestroy(sl.s);
  }
  if (p->spec_mod) (void)hipModuleUnload(p->spec_mod);
  if (p->spec_mod2) (void)hipModuleUnload(p->spec_mod2);
  if (p->spec_mod3) (void)hipModuleUnload(p->spec_mod3);
  if (p->spec_mod4) (void)hipModuleUnload(p->spec_mod4);
  p->q0.release(); p->q1.release(); p->ctr.release();
  p->rec.release(); p->hdr.release();
  p->flat_grid.release(); p->flat_bits.release(); p->flat_list.release(); p->dc_grid.release(); p->dc_dist.release(); p->dc_fv.release(); p->dc_nrm.release(); p->dc_edge.release();
  for (auto e : p->ev) if (e) (void)hipEventDestroy(e);
  if (p->h_ctr) (void)hipHostFree(p->h_ctr);
  if (p->stream) (void)hipStreamDestroy(p->stream);
  delete p;
}

extern "C" int gsdf_hip_program_bounds(const gsdf_program* p, float bb[6]) {
  if (!p || !bb) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  std::memcpy(bb, p->prog.bb, sizeof(float) * 6);
  return GSDF_OK;
}
extern "C" int gsdf_hip_program_is2d(const gsdf_program* p) { return p && p->prog.is2d ? 1 : 0; }
extern "C" int gsdf_hip_program_info(const gsdf_program* p, uint32_t* code_words, uint32_t* lds_slots) {
  if (!p) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  if (code_words) *code_words = (uint32_t)p->prog.code.size();
  if (lds_slots) *lds_slots = (uint32_t)p->prog.nslots;
  return GSDF_OK;
}
extern "C" uint64_t gsdf_hip_evaluations(const gsdf_program* p) { return p ? p->evals + p->evals_host.load() : 0; }

static int eval_dev(gsdf_program* p, int dim, const void* d_pos, size_t stride_bytes, float* d_dist, size_t n, hipStream_t s, bool count = true) {
  if (stride_bytes % 4 != 0 || stride_bytes < (size_t)dim * 4) return fail(GSDF_ERR_BAD_ARGUMENT, "bad position stride");
  const int k = p->batch_k();
  static const int eval_bpc = [] { const char* e = getenv("GSDF_HIP_EVAL_BPC"); return e ? atoi(e) : 64; }();  // tuning knob: finer grids drain evenly (8 -> 64 per CU: +10 % on npt-flange)
  const unsigned grid = grid_for((n + k - 1) / k, p->num_cu, eval_bpc);
  const uint32_t sf = (uint32_t)(stride_bytes / 4);
  const float* q = (const float*)d_pos;
  const uint64_t nn = (uint64_t)n;
  const int w = p->sweep_waves(k);
#define LAUNCH_EVAL(D, KK, WW) hipLaunchKernelGGL((eval_kernel<D, KK, WW>), dim3(grid), dim3(BLOCK), p->lds_bytes(KK), s, p->d_code, q, sf, d_dist, nn)
  if (p->f_eval && p->spec_eval_k == k) {  // whichever W the specialised kernel was built for: same launch
    HIP_TRY(launch_fn(p->f_eval, grid, BLOCK, p->lds_bytes(k), s, (const uint32_t*)p->d_code, q, sf, d_dist, nn));
  } else
  if (dim == 3) {
    if (k == 4) { if (w == 4) LAUNCH_EVAL(3, 4, 4); else LAUNCH_EVAL(3, 4, 3); }
    else if (k == 2) { if (w == 4) LAUNCH_EVAL(3, 2, 4); else LAUNCH_EVAL(3, 2, 3); }
    else LAUNCH_EVAL(3, 1, 4);
  } else {
    if (k == 4) { if (w == 4) LAUNCH_EVAL(2, 4, 4); else LAUNCH_EVAL(2, 4, 3); }
    else if (k == 2) { if (w == 4) LAUNCH_EVAL(2, 2, 4); else LAUNCH_EVAL(2, 2, 3); }
    else LAUNCH_EVAL(2, 1, 4);
  }
#undef LAUNCH_EVAL
  HIP_TRY(hipGetLastError());
  if (count) p->evals += n;  // (concurrent host-buffer callers count through evals_host instead)
  return GSDF_OK;
}

// ---- caller buffers the GPU can reach directly (pinned + device-mapped): no staging copy at all ----------------------
// Process-wide table of host ranges registered through gsdf_hip_host_alloc / gsdf_hip_host_register. A call whose
// positions AND distances lie inside such ranges runs the kernel straight on the caller's memory across PCIe.
namespace {
struct HostRange { char* p; size_t n; bool owned; };
std::mutex g_reg_mu;
std::vector<HostRange> g_reg;
void* reg_device_ptr(const void* h, size_t n) {
  std::lock_guard<std::mutex> lk(g_reg_mu);
  for (const HostRange& r : g_reg)
    if ((const char*)h >= r.p && (const char*)h + n <= r.p + r.n) {
      void* d = nullptr;
      if (hipHostGetDevicePointer(&d, r.p, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
      return (char*)d + ((const char*)h - r.p);
    }
  return nullptr;
}
}  // namespace
extern "C" void* gsdf_hip_host_alloc(size_t bytes) {
  void* h = nullptr;
  if (bytes == 0 || hipHostMalloc(&h, bytes, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  std::lock_guard<std::mutex> lk(g_reg_mu);
  g_reg.push_back(HostRange{(char*)h, bytes, true});
  return h;
}
extern "C" int gsdf_hip_host_register(void* ptr, size_t bytes) {
  if (!ptr || bytes == 0) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterMapped | hipHostRegisterPortable));
  std::lock_guard<std::mutex> lk(g_reg_mu);
  g_reg.push_back(HostRange{(char*)ptr, bytes, false});
  return GSDF_OK;
}
extern "C" int gsdf_hip_host_release(void* ptr) {  // gsdf_hip_host_alloc'ed: freed; gsdf_hip_host_register'ed: unregistered
  if (!ptr) return GSDF_OK;
  HostRange r{nullptr, 0, false};
  {
    std::lock_guard<std::mutex> lk(g_reg_mu);
    for (size_t i = 0; i < g_reg.size(); i++)
      if (g_reg[i].p == (char*)ptr) { r = g_reg[i]; g_reg.erase(g_reg.begin() + (long)i); break; }
  }
  if (!r.p) return fail(GSDF_ERR_BAD_ARGUMENT, "not a registered host buffer");
  if (r.owned) HIP_TRY(hipHostFree(r.p));
  else HIP_TRY(hipHostUnregister(r.p));
  return GSDF_OK;
}

static constexpr size_t kSmallPos = (size_t)1 << 20, kSmallDist = (size_t)1 << 18;

static int slot_acquire(gsdf_program* p, int* idx) {
  std::unique_lock<std::mutex> lk(p->slot_mu);
  for (;;) {
    for (int i = 0; i < gsdf_program::kSlots; i++)
      if (!p->slot[i].busy) { p->slot[i].busy = true; p->slot[i].waiting = false; p->slot[i].gen = (p->slot[i].gen + 1u) & 0x7fffffu; *idx = i; return GSDF_OK; }
    p->slot_cv.wait(lk);
  }
}
static void slot_release(gsdf_program* p, int idx) {
  { std::lock_guard<std::mutex> lk(p->slot_mu); p->slot[idx].busy = false; }
  p->slot_cv.notify_one();
}

// Enqueue one host-buffer evaluation on a staging slot (no wait). Small calls (what the reference's renderers issue:
// <= 32768 points, gsdfaux.go:89,113) make no DMA round trips: the kernel reads the positions from -- and writes the
// distances to -- pinned, device-mapped host memory across PCIe itself: the caller's own buffers when they are registered
// (zero copy), else the slot's staging buffers (one memcpy in, one out).
static int eval_submit(gsdf_program* p, int dim, const void* pos, size_t stride, size_t n_pos, float* dist, size_t n_dist, int* ticket) {
  if (!p) return fail(GSDF_ERR_BAD_ARGUMENT, "null program");
  if (n_pos != n_dist) return fail(GSDF_ERR_LENGTH_MISMATCH, "position and distance buffer length mismatch");
  if (n_pos == 0) return fail(GSDF_ERR_EMPTY_BUFFERS, "empty buffers");
  if (!pos || !dist) return fail(GSDF_ERR_BAD_ARGUMENT, "null buffer");
  if (p->prog.is2d != (dim == 2)) return fail(GSDF_ERR_DIMENSION, dim == 2 ? "program is 3D, eval2 called" : "program is 2D, eval3 called");
  HIP_TRY(hipSetDevice(p->device));
  const size_t pbytes = n_pos * stride;
  int si = -1;
  int rc = slot_acquire(p, &si);
  if (rc) return rc;
  gsdf_program::Slot& sl = p->slot[si];
  auto bail = [&](int code) { slot_release(p, si); return code; };
  if (!sl.s && hipStreamCreateWithFlags(&sl.s, hipStreamNonBlocking) != hipSuccess) return bail(fail(GSDF_ERR_HIP, "hipStreamCreate failed"));
  void* dp = reg_device_ptr(pos, pbytes);
  void* dd = dp ? reg_device_ptr(dist, n_pos * sizeof(float)) : nullptr;
  sl.zero_copy = dp && dd;
  sl.user_dist = dist;
  sl.n = n_pos;
  if (!sl.zero_copy) {
    if (pbytes > kSmallPos || n_pos > kSmallDist) return bail(fail(GSDF_ERR_BAD_ARGUMENT, "internal: large call on the small path"));
    if (!sl.h_pos && hipHostMalloc(&sl.h_pos, kSmallPos, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) return bail(fail(GSDF_ERR_HIP, "hipHostMalloc failed"));
    if (!sl.h_dist && hipHostMalloc((void**)&sl.h_dist, kSmallDist * sizeof(float), hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) return bail(fail(GSDF_ERR_HIP, "hipHostMalloc failed"));
    std::memcpy(sl.h_pos, pos, pbytes);
    if (hipHostGetDevicePointer(&dp, sl.h_pos, 0) != hipSuccess || hipHostGetDevicePointer(&dd, sl.h_dist, 0) != hipSuccess) return bail(fail(GSDF_ERR_HIP, "hipHostGetDevicePointer failed"));
  }
  rc = eval_dev(p, dim, dp, stride, (float*)dd, n_pos, sl.s, /*count=*/false);
  if (rc) return bail(rc);
  p->evals_host.fetch_add(n_pos);
  *ticket = si | (int)(sl.gen << 8);
  return GSDF_OK;
}
static int eval_wait(gsdf_program* p, int tk) {
  const int ticket = tk & 0xff;
  if (!p || tk < 0 || ticket >= gsdf_program::kSlots) return fail(GSDF_ERR_BAD_ARGUMENT, "bad evaluation ticket");
  {  // the slot must be in flight for THIS ticket, and nobody else may be waiting on it
    std::lock_guard<std::mutex> lk(p->slot_mu);
    gsdf_program::Slot& s0 = p->slot[ticket];
    if (!s0.busy || s0.waiting || s0.gen != ((unsigned)tk >> 8)) return fail(GSDF_ERR_BAD_ARGUMENT, "bad evaluation ticket (stale, or waited for twice)");
    s0.waiting = true;
  }
  gsdf_program::Slot& sl = p->slot[ticket];
  hipError_t e = hipStreamSynchronize(sl.s);
  if (e == hipSuccess && !sl.zero_copy) std::memcpy(sl.user_dist, sl.h_dist, sl.n * sizeof(float));
  slot_release(p, ticket);
  if (e != hipSuccess) return fail(GSDF_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
  return GSDF_OK;
}

static int eval_host(gsdf_program* p, int dim, const void* pos, size_t stride, size_t n_pos, float* dist, size_t n_dist) {
  if (!p) return fail(GSDF_ERR_BAD_ARGUMENT, "null program");
  const size_t pbytes = n_pos * stride;
  const bool small = pbytes <= kSmallPos && n_pos <= kSmallDist;
  if (small || (pos && dist && n_pos == n_dist && n_pos && reg_device_ptr(pos, pbytes) && reg_device_ptr(dist, n_pos * sizeof(float)))) {
    int t = -1;
    const int rc = eval_submit(p, dim, pos, stride, n_pos, dist, n_dist, &t);
    return rc ? rc : eval_wait(p, t);
  }
  if (n_pos != n_dist) return fail(GSDF_ERR_LENGTH_MISMATCH, "position and distance buffer length mismatch");
  if (n_pos == 0) return fail(GSDF_ERR_EMPTY_BUFFERS, "empty buffers");
  if (!pos || !dist) return fail(GSDF_ERR_BAD_ARGUMENT, "null buffer");
  if (p->prog.is2d != (dim == 2)) return fail(GSDF_ERR_DIMENSION, dim == 2 ? "program is 3D, eval2 called" : "program is 2D, eval3 called");
  HIP_TRY(hipSetDevice(p->device));
  // large calls: DMA in, kernel, DMA out on the program's stream (one caller at a time, as before)
  static std::mutex big_mu;
  std::lock_guard<std::mutex> lk(big_mu);
  if (pbytes > p->cap_pos_bytes) {
    if (p->d_pos) (void)hipFree(p->d_pos);
    p->d_pos = nullptr; p->cap_pos_bytes = 0;
    HIP_TRY(hipMalloc(&p->d_pos, pbytes));
    p->cap_pos_bytes = pbytes;
  }
  if (n_pos > p->cap_dist) {
    if (p->d_dist) (void)hipFree(p->d_dist);
    p->d_dist = nullptr; p->cap_dist = 0;
    HIP_TRY(hipMalloc((void**)&p->d_dist, n_pos * sizeof(float)));
    p->cap_dist = n_pos;
  }
  HIP_TRY(hipMemcpyAsync(p->d_pos, pos, pbytes, hipMemcpyHostToDevice, p->stream));
  int rc = eval_dev(p, dim, p->d_pos, stride, p->d_dist, n_pos, p->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(dist, p->d_dist, n_pos * sizeof(float), hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return GSDF_OK;
}

// Pipelined form of the host-buffer API: submit returns at once with a ticket (at most 4 in flight per program: a fifth
// submit waits for a free slot), wait blocks until that call's distances are in `dist`. For callers that can prepare the
// next batch while the previous one is on the GPU. Batches of up to 2^18 points (2^20 position bytes), or any size in
// registered memory.
extern "C" int gsdf_hip_eval3_submit(gsdf_program* p, const void* pos, size_t stride, size_t n_pos, float* dist, size_t n_dist, int* ticket) {
  if (!ticket) return fail(GSDF_ERR_BAD_ARGUMENT, "null ticket");
  if (n_pos * stride > kSmallPos || n_pos > kSmallDist) {
    if (!(pos && dist && reg_device_ptr(pos, n_pos * stride) && reg_device_ptr(dist, n_pos * sizeof(float))))
      return fail(GSDF_ERR_BAD_ARGUMENT, "submit takes at most 262144 points per call unless both buffers are registered host memory");
  }
  return eval_submit(p, 3, pos, stride, n_pos, dist, n_dist, ticket);
}
extern "C" int gsdf_hip_eval_wait(gsdf_program* p, int ticket) { return eval_wait(p, ticket); }

extern "C" int gsdf_hip_eval3(gsdf_program* p, const void* pos, size_t stride, size_t n_pos, float* dist, size_t n_dist) {
  return eval_host(p, 3, pos, stride, n_pos, dist, n_dist);
}
extern "C" int gsdf_hip_eval2(gsdf_program* p, const void* pos, size_t stride, size_t n_pos, float* dist, size_t n_dist) {
  return eval_host(p, 2, pos, stride, n_pos, dist, n_dist);
}
extern "C" int gsdf_hip_eval3_dev(gsdf_program* p, const void* d_pos, size_t stride, float* d_dist, size_t n, void* stream) {
  if (!p || !d_pos || !d_dist) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  if (n == 0) return fail(GSDF_ERR_EMPTY_BUFFERS, "empty buffers");
  if (p->prog.is2d) return fail(GSDF_ERR_DIMENSION, "program is 2D, eval3 called");
  return eval_dev(p, 3, d_pos, stride, d_dist, n, stream ? (hipStream_t)stream : p->stream);
}
extern "C" int gsdf_hip_eval2_dev(gsdf_program* p, const void* d_pos, size_t stride, float* d_dist, size_t n, void* stream) {
  if (!p || !d_pos || !d_dist) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  if (n == 0) return fail(GSDF_ERR_EMPTY_BUFFERS, "empty buffers");
  if (!p->prog.is2d) return fail(GSDF_ERR_DIMENSION, "program is 3D, eval2 called");
  return eval_dev(p, 2, d_pos, stride, d_dist, n, stream ? (hipStream_t)stream : p->stream);
}

extern "C" int gsdf_hip_normals3(gsdf_program* p, const float* pos, float* normals, size_t n, float step) {
  if (!p || !pos || !normals) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  step *= 0.5f;
  if (!(step > 0)) return fail(GSDF_ERR_BAD_ARGUMENT, "invalid step");
  if (n == 0) return fail(GSDF_ERR_EMPTY_BUFFERS, "empty buffers");
  if (p->prog.is2d) return fail(GSDF_ERR_DIMENSION, "program is 2D");
  HIP_TRY(hipSetDevice(p->device));
  float *d_p = nullptr, *d_n = nullptr;
  HIP_TRY(hipMalloc((void**)&d_p, n * 12));
  if (hipMalloc((void**)&d_n, n * 12) != hipSuccess) { (void)hipFree(d_p); return fail(GSDF_ERR_HIP, "hipMalloc failed"); }
  int rc = GSDF_OK;
  do {
    if (hipMemcpyAsync(d_p, pos, n * 12, hipMemcpyHostToDevice, p->stream) != hipSuccess) { rc = fail(GSDF_ERR_HIP, "H2D copy failed"); break; }
    spec_aux(p);
    if (p->f_normals) {
      if (launch_fn(p->f_normals, grid_for(n, p->num_cu, 8), BLOCK, p->lds_bytes(2), p->stream, (const uint32_t*)p->d_code, (const float*)d_p, (float*)d_n,
                    (uint64_t)n, (float)step) != hipSuccess) { rc = fail(GSDF_ERR_HIP, "kernel launch failed"); break; }
    } else
    hipLaunchKernelGGL(normals_kernel, dim3(grid_for(n, p->num_cu, 8)), dim3(BLOCK), p->lds_bytes(2), p->stream, p->d_code, d_p, d_n, (uint64_t)n, step);
    if (hipGetLastError() != hipSuccess) { rc = fail(GSDF_ERR_HIP, "normals kernel launch failed"); break; }
    if (hipMemcpyAsync(normals, d_n, n * 12, hipMemcpyDeviceToHost, p->stream) != hipSuccess) { rc = fail(GSDF_ERR_HIP, "D2H copy failed"); break; }
    if (hipStreamSynchronize(p->stream) != hipSuccess) { rc = fail(GSDF_ERR_HIP, "stream sync failed"); break; }
    p->evals += 6 * n;
  } while (0);
  (void)hipFree(d_p);
  (void)hipFree(d_n);
  return rc;
}

// ---- mesher -----------------------------------------------------------------------------------
namespace {
// ms3.Box.ScaleCentered(1.01) = NewCenteredBox(Center(), MulElem(scale, Size())) [external]; float32, unfused.
void scale_centered(const float bb[6], float s, float mn[3], float mx[3]) {
  for (int a = 0; a < 3; a++) {
    const float c = 0.5f * (bb[a] + bb[a + 3]);
    const float size = bb[a + 3] - bb[a];
    float sz = fmaxf(s, 0.f) * size;
    const float half = 0.5f * fmaxf(sz, 0.f);
    mn[a] = c - half;
    mx[a] = c + half;
  }
}
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
};
}  // namespace

extern "C" int gsdf_hip_mesh_octree(gsdf_program* p, float res, const gsdf_mesh_opts* opts_in, gsdf_mesh** out) {
  if (!p || !out) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  *out = nullptr;
  if (p->prog.is2d) return fail(GSDF_ERR_DIMENSION, "program is 2D");
  if (!(res > 0) || std::isnan(res) || std::isinf(res)) return fail(GSDF_ERR_RESOLUTION, "invalid renderer cube resolution");
  gsdf_mesh_opts opts{};
  opts.prune = 1; opts.shard_rank = 0; opts.shard_count = 1; opts.share_corners = 0;
  if (opts_in) opts = *opts_in;
  if (opts.shard_count < 1 || opts.shard_rank < 0 || opts.shard_rank >= opts.shard_count) return fail(GSDF_ERR_BAD_ARGUMENT, "bad shard rank/count");
  HIP_TRY(hipSetDevice(p->device));
  hipStream_t s = opts.stream ? (hipStream_t)opts.stream : p->stream;

  // Octree.Reset (octreerenderer.go:71-128) + makeICube (:222-235)
  float mn[3], mx[3];
  scale_centered(p->prog.bb, 1.01f, mn, mx);
  const float longAxis = fmaxf(mx[0] - mn[0], fmaxf(mx[1] - mn[1], mx[2] - mn[2]));
  const float l2 = (float)std::log2((double)(longAxis / res));
  const int levels = (int)std::ceil(l2) + 1;
  if (levels <= 1) return fail(GSDF_ERR_RESOLUTION, "resolution not fine enough for marching cubes");
  if (levels > 17) return fail(GSDF_ERR_RESOLUTION, "resolution too fine: more than 17 octree levels");
  const float ox = mn[0], oy = mn[1], oz = mn[2];

  gsdf_mesh* m = new (std::nothrow) gsdf_mesh();
  if (!m) return fail(GSDF_ERR_BAD_ARGUMENT, "out of memory");
  m->device = p->device;
  m->stream = s;
  m->st.levels = levels;
  m->st.res = res;
  m->st.origin[0] = ox; m->st.origin[1] = oy; m->st.origin[2] = oz;
  auto bail = [&](int code) { gsdf_hip_mesh_destroy(m); return code; };
#define HIP_TRYM(expr)                                                                                          \
  do {                                                                                                          \
    hipError_t _e = (expr);                                                                                     \
    if (_e != hipSuccess) return bail(fail(GSDF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e))); \
  } while (0)

  if (!p->h_ctr) HIP_TRYM(hipHostMalloc(&p->h_ctr, 4096, hipHostMallocDefault));
  static_assert(sizeof(MeshCounters) <= 2048, "counters: first half of the pinned block (second half: DCCounters)");
  for (auto& e : p->ev)
    if (!e) HIP_TRYM(hipEventCreate(&e));
  hipEvent_t ev0 = p->ev[0], ev1 = p->ev[1], ev2 = p->ev[2];

  // ---- level-synchronous descent from the top cube to level lq = min(levels, 3). The whole chain
  // (memset, one prune launch per level, the leaf kernel) is enqueued without a host round trip: each
  // kernel reads its input count from the previous level's counter in device memory.
  const int lq = levels < 3 ? levels : 3;
  // multi-GPU: bricks of level ls are dealt to ranks by a hash of their coordinates (brick_owner).
  const int ls = levels < lq + 2 ? levels : lq + 2;
  const int prune_cols = p->prog.nslots + p->prog.lip_depth;  // interval mode: two points per lane + the interval stack
  const size_t lds_prune = (size_t)(prune_cols > 0 ? prune_cols : 1) * 2 * BLOCK * sizeof(float) + PRUNE_STAGE * sizeof(Cube) + 64;
  int pmask = opts.prune & ~GSDF_PRUNE_ASSUME_SDF;  // levels to test: 0 none, 1 all, else bit L = Level L
  if ((opts.prune & GSDF_PRUNE_ASSUME_SDF) && pmask == 0) pmask = 1;
  const int ptest = (opts.prune & GSDF_PRUNE_ASSUME_SDF) ? 2 : 1;
  int lk, lw;
  size_t lds_m;
  p->leaf_config(&lk, &lw, &lds_m);
  uint64_t qcap = p->q0.cap / sizeof(Cube);
  {
    // 1 M cubes (8 MB) per queue to start with; GSDF_HIP_QCAP_MIN lowers it so that tests can drive the
    // overflow -> grow -> rerun path (the arenas only ever grow, so a handle that already meshed keeps its size)
    const char* e = getenv("GSDF_HIP_QCAP_MIN");
    const uint64_t qmin = e ? (uint64_t)strtoull(e, nullptr, 10) : ((uint64_t)1 << 20);
    if (qcap < qmin) qcap = qmin;
    if (qcap < 64) qcap = 64;
  }
  uint64_t want = opts.max_tris;
  MeshCounters& hc = *(MeshCounters*)p->h_ctr;
  hc = MeshCounters{};
  MeshCounters* d_ctr = nullptr;
  constexpr size_t kCtrBytes = (sizeof(MeshCounters) + 255) & ~(size_t)255;  // the group sums follow the counters: one memset clears both
  bool used_brick = false, two_kernel = false, ctr_on_host = false;
  int chain_first = levels;  // levels <= chain_first were tested by prune_kernel (one launch per level), the ones above speculatively
  static const bool ctr_from_kernel = [] { const char* e = getenv("GSDF_HIP_CTR_FROM_KERNEL"); return !e || atoi(e) != 0; }();  // developer knob (A/B timing)
  float ms01 = 0, ms12 = 0, ms13 = 0;
  for (int attempt = 0;; attempt++) {
    ctr_on_host = false;
    HIP_TRYM(p->q0.ensure(qcap * sizeof(Cube)));
    HIP_TRYM(p->q1.ensure(qcap * sizeof(Cube)));
    const uint64_t cap0 = p->q0.cap / sizeof(Cube), cap1 = p->q1.cap / sizeof(Cube);
    gsdf_program::Arena* q[2] = {&p->q0, &p->q1};
    const uint64_t capq[2] = {cap0, cap1};
    // triangle buffer: caller's size, else a pooled buffer, else a guess that is corrected by one exact rerun
    if (!m->d_tris) {
      uint64_t need = want ? want : (p->last_tris ? p->last_tris + p->last_tris / 16 + 1024 : (uint64_t)1 << 20);
      if (opts.host_output) {
        // triangles straight into pinned host memory: the stage flushes of leaf_kernel are 4.6 KB coalesced bursts,
        // which PCIe takes well; the transfer then overlaps the kernel instead of following it
        void* hb = nullptr;
        size_t hcap = 0;
        if (int rc = host_buf(&hb, &hcap, (size_t)need * 36)) return bail(rc);
        m->d_tris = (float*)hb;
        m->cap = hcap / 36;
        m->host_out = true;
      } else {
        m->d_tris = pool_take(p->device, need, &m->cap);
        if (!m->d_tris) {
          HIP_TRYM(hipMalloc((void**)&m->d_tris, need * 36));
          m->cap = need;
        }
      }
    }
    // 64-leaf blocks the queue capacity allows for, and their groups (two-kernel leaf phase)
    uint64_t lbound = capq[lq & 1] << (3 * (lq - 1));
    {
      const uint64_t full = (levels - lq) * 3 >= 40 ? UINT64_MAX : ((uint64_t)1 << (3 * (levels - 1)));
      if (lbound > full) lbound = full;
    }
    // The record arena (2 560 B per block) is sized for the blocks a mesh of this handle actually had (+ 1/8), not for the queue
    // capacity: a million-cube queue would ask for 2.7 GB, and each queue regrowth for four times more. First mesh: 384 K
    // blocks (1 GB); a mesh that needs more says so through its survivor count and is repeated once with the exact size.
    constexpr uint64_t kRecBlocks0 = (uint64_t)384 << 10;
    const uint64_t nblk_q = (lbound + 63) / 64;
    uint64_t nblk = p->rec_blocks ? p->rec_blocks : kRecBlocks0;
    if (nblk > nblk_q) nblk = nblk_q;
    const uint64_t ngrp = (nblk + MARCH_GROUP - 1) / MARCH_GROUP;
    bool want_two = !fused_leaf() && !(lq == 3 && lk == 4 && opts.share_corners);
    if (want_two && (p->hdr.ensure(nblk * sizeof(uint32_t)) != hipSuccess || p->rec.ensure(nblk * (size_t)REC_BLOCK * sizeof(uint32_t)) != hipSuccess)) {
      (void)hipGetLastError();  // no room for the records: the fused kernel needs none
      p->hdr.release(); p->rec.release();
      want_two = false;
    }
    const size_t clear_bytes = kCtrBytes + (want_two ? ngrp * sizeof(unsigned long long) : 0);
    {
      const void* before = p->ctr.p;
      HIP_TRYM(p->ctr.ensure(clear_bytes));
      if (p->ctr.p != before) p->ctr_clean = 0;
    }
    d_ctr = (MeshCounters*)p->ctr.p;
    if (p->ctr_clean && p->ctr_clean_stream != s) p->ctr_settle();  // cleared on another stream: let that finish, do not rely on it
    if (p->ctr_clean < clear_bytes) HIP_TRYM(hipMemsetAsync(d_ctr, 0, clear_bytes, s));
    p->ctr_clean = 0;  // dirty from here on
    HIP_TRYM(hipEventRecord(ev0, s));
    static const int prune_bpc = [] { const char* e = getenv("GSDF_HIP_PRUNE_BPC"); return e ? atoi(e) : 4; }();  // tuning knob
    // The first S levels (at most 7: 299,593 cubes) are centre-tested speculatively, every cube of the complete octree at once,
    // and resolved by a second launch (kernels.h: prune_spec_kernel / prune_resolve_kernel): two launches instead of a chain of
    // S dependent ones. GSDF_HIP_PRUNE_SPEC=0 keeps one launch per level (cross-check in the tests).
    static const bool use_spec = [] { const char* e = getenv("GSDF_HIP_PRUNE_SPEC"); return !e || atoi(e) != 0; }();
    int first_level = levels;  // first level of the per-level chain
    chain_first = levels;
    unsigned* spec_part = nullptr;  // statistics rows of the speculative top, for the first per-level launch to add up
    unsigned spec_rows = 0, spec_mask = 0;
    int spec_top_S = 0;
    auto test_mask_of = [](int pm) { return pm == 1 ? 0xffffffffu : (unsigned)pm; };
    if (use_spec) {
      const int S = levels - lq + 1 < 7 ? levels - lq + 1 : 7;
      const int last_spec = levels - (S - 1);
      unsigned n_spec = 0;
      for (int j = 0; j < S; j++) n_spec += 1u << (3 * j);
      // the resolve stage's statistics: a row of 16 counts per workgroup behind the pass bytes, added up by the first per-level
      // launch -- if there is one (else the resolve stage issues its atomics itself)
      const unsigned rrows = (n_spec + SPEC_STAGE - 1) / SPEC_STAGE;
      const size_t part_off = ((size_t)n_spec + 63) & ~(size_t)63;
      HIP_TRYM(p->spec_pass.ensure(part_off + (size_t)rrows * 16 * sizeof(unsigned)));
      const bool chain_follows = last_spec - 1 >= lq;
      spec_part = chain_follows ? (unsigned*)((char*)p->spec_pass.p + part_off) : nullptr;
      spec_rows = rrows;
      spec_top_S = levels | (S << 8);
      spec_mask = test_mask_of(pmask);
      const unsigned test_mask = test_mask_of(pmask);
      const int shard_level = opts.shard_count > 1 ? ls : -1;
      const unsigned sgrid = grid_for(n_spec, p->num_cu, 8);
      if (p->f_prune_spec) {
        HIP_TRYM(launch_fn(p->f_prune_spec, sgrid, BLOCK, lds_prune, s, (const uint32_t*)p->d_code, (int)levels, (unsigned)n_spec, (int)prune_cols,
                           (int)p->prog.nslots, ox, oy, oz, res, (unsigned)test_mask, (int)ptest, (int)shard_level, (unsigned)opts.shard_rank,
                           (unsigned)opts.shard_count, (uint8_t*)p->spec_pass.p));
      } else {
        hipLaunchKernelGGL(prune_spec_kernel, dim3(sgrid), dim3(BLOCK), lds_prune, s, p->d_code, levels, n_spec, prune_cols, p->prog.nslots, ox, oy,
                           oz, res, test_mask, ptest, shard_level, (unsigned)opts.shard_rank, (unsigned)opts.shard_count, (uint8_t*)p->spec_pass.p);
      }
      HIP_TRYM(hipGetLastError());
      hipLaunchKernelGGL(prune_resolve_kernel, dim3((n_spec + SPEC_STAGE - 1) / SPEC_STAGE), dim3(BLOCK), 0, s, (const uint8_t*)p->spec_pass.p, levels, S,
                         n_spec, test_mask, (Cube*)q[last_spec & 1]->p, (unsigned long long)capq[last_spec & 1], d_ctr, spec_part);
      HIP_TRYM(hipGetLastError());
      first_level = last_spec - 1;
      chain_first = first_level;
    }
    for (int level = first_level; level >= lq; level--) {
      const int expand = level != levels;
      const int do_test = (level >= 3 && (pmask == 1 || (pmask > 1 && ((pmask >> level) & 1)))) ? ptest : 0;
      // upper bound of candidates at this level (for the grid only): 8^(levels-level), capped by the queue
      uint64_t bound = (levels - level) * 3 >= 40 ? UINT64_MAX : ((uint64_t)1 << (3 * (levels - level)));
      if (bound > capq[(level + 1) & 1] * 8) bound = capq[(level + 1) & 1] * 8;
      if (p->f_prune) {
        HIP_TRYM(launch_fn(p->f_prune, grid_for(bound, p->num_cu, prune_bpc), BLOCK, lds_prune, s, (const uint32_t*)p->d_code,
                           (const Cube*)q[(level + 1) & 1]->p, (unsigned long long)capq[(level + 1) & 1], (int)expand, (int)level,
                           (int)prune_cols, (int)p->prog.nslots, ox, oy, oz, res, (int)do_test,
                           (Cube*)q[level & 1]->p, (unsigned long long)capq[level & 1], (int)((opts.shard_count > 1 && level == ls) ? 1 : 0),
                           (unsigned)opts.shard_rank, (unsigned)opts.shard_count, d_ctr, (const unsigned*)(level == first_level ? spec_part : nullptr),
                           (unsigned)spec_rows, (int)spec_top_S, (unsigned)spec_mask));
      } else
      hipLaunchKernelGGL(prune_kernel, dim3(grid_for(bound, p->num_cu, prune_bpc)), dim3(BLOCK), lds_prune, s, p->d_code,
                         (const Cube*)q[(level + 1) & 1]->p, (unsigned long long)capq[(level + 1) & 1], expand, level, prune_cols,
                         p->prog.nslots, ox, oy, oz, res, do_test, (Cube*)q[level & 1]->p,
                         (unsigned long long)capq[level & 1], (opts.shard_count > 1 && level == ls) ? 1 : 0,
                         (unsigned)opts.shard_rank, (unsigned)opts.shard_count, d_ctr, (const unsigned*)(level == first_level ? spec_part : nullptr),
                         spec_rows, spec_top_S, spec_mask);
      HIP_TRYM(hipGetLastError());
    }
    HIP_TRYM(hipEventRecord(ev1, s));
    {
      const uint64_t bound = lbound;
      const unsigned long long tcap = opts.max_tris ? opts.max_tris : m->cap;
      static const int leaf_bpc = [] { const char* e = getenv("GSDF_HIP_LEAF_BPC"); return e ? atoi(e) : 64; }();  // grid = up to 64 workgroups per CU (4 resident): a few grid-stride iterations each, so the CUs drain evenly at the end (8 per CU: +8 % kernel time; one iteration per workgroup: +10 %)
#define LAUNCH_LEAF(KK, WW)                                                                                           \
  hipLaunchKernelGGL((leaf_kernel<KK, WW>), dim3(grid_for(bound, p->num_cu, leaf_bpc)), dim3(BLOCK), lds_m, s, p->d_code,      \
                     (const Cube*)q[lq & 1]->p, (unsigned long long)capq[lq & 1], lq, p->prog.nslots, ox, oy, oz, res,  \
                     m->d_tris, tcap, d_ctr)
      if (want_two) {
        // two kernels: evaluation + cut-leaf records, then marching cubes over the records
        uint32_t* d_hdr = (uint32_t*)p->hdr.p;
        uint32_t* d_rec = (uint32_t*)p->rec.p;
        unsigned long long* d_psum = (unsigned long long*)((char*)p->ctr.p + kCtrBytes);  // cleared with the counters
#define LAUNCH_LEAF_EVAL_U(KK, WW, UU, NN, LDS)                                                                                    \
  hipLaunchKernelGGL((leaf_eval_kernel<KK, WW, UU, NN>), dim3(grid_for(bound, p->num_cu, leaf_bpc)), dim3(BLOCK), LDS, s, p->d_code, \
                     (const Cube*)q[lq & 1]->p, (unsigned long long)capq[lq & 1], lq, p->prog.nslots, ox, oy, oz, res, d_hdr,   \
                     d_rec, d_psum, (unsigned long long)nblk, d_ctr)
        // lq == 3 (three levels or more): a wave pass is one level-3 cube (column bricks, scalar prefetched cube load), its
        // case-count table in LDS unless that costs a workgroup per CU; else a few leaves (occupancy is no concern: table in LDS)
#define LAUNCH_LEAF_EVAL(KK, WW)                                                             \
  do {                                                                                       \
    if (lq != 3) LAUNCH_LEAF_EVAL_U(KK, WW, false, true, lds_m + 256);                       \
    else if (p->leaf_nt_in_lds()) LAUNCH_LEAF_EVAL_U(KK, WW, true, true, lds_m);             \
    else LAUNCH_LEAF_EVAL_U(KK, WW, true, false, lds_m);                                     \
  } while (0)
        if (p->f_leaf && p->spec_leaf_k == lk && lq == 3) {
          HIP_TRYM(launch_fn(p->f_leaf, grid_for(bound, p->num_cu, leaf_bpc), BLOCK, lds_m, s, (const uint32_t*)p->d_code, (const Cube*)q[lq & 1]->p,
                             (unsigned long long)capq[lq & 1], (int)lq, (int)p->prog.nslots, ox, oy, oz, res, d_hdr, d_rec, d_psum,
                             (unsigned long long)nblk, d_ctr));
        } else {
          // ahead-of-time kernels exist at the scratch-free occupancies only (tests/test_kernel_resources.py)
          if (lk == 4) { if (lw == 2) LAUNCH_LEAF_EVAL(4, 2); else LAUNCH_LEAF_EVAL(4, 3); }
          else if (lk == 2) LAUNCH_LEAF_EVAL(2, 3);
          else LAUNCH_LEAF_EVAL(1, 4);
        }
#undef LAUNCH_LEAF_EVAL
#undef LAUNCH_LEAF_EVAL_U
        HIP_TRYM(hipGetLastError());
        HIP_TRYM(hipEventRecord(p->ev[3], s));
        two_kernel = true;
        // one wave of workgroups: each takes an equal share of the records (computed on device from the group sums)
        static const int march_bpc = [] { const char* e = getenv("GSDF_HIP_MARCH_BPC"); return e ? atoi(e) : 7; }();  // tuning knob (7 fit a CU)
        const size_t lds_march = (size_t)11 * BLOCK * 4 + 5 * BLOCK * 4 + 256 * 16 + (BLOCK + 1) * 4 + 8 * 4 + 8 + 14 * 8;
        hipLaunchKernelGGL(march_records_kernel, dim3(grid_for(nblk_q, p->num_cu, march_bpc)), dim3(BLOCK), lds_march, s, d_hdr, d_rec,
                           d_psum, (unsigned long long)nblk, lq, ox, oy, oz, res, m->d_tris, (uint64_t)tcap, d_ctr,
                           ctr_from_kernel ? (MeshCounters*)p->h_ctr : (MeshCounters*)nullptr);
        ctr_on_host = ctr_from_kernel;
      } else if (lq == 3 && lk == 4 && opts.share_corners) {
        // exact corner sharing: one wave per level-3 brick
        const size_t lds_b = (size_t)(p->prog.nslots * 4) * BLOCK * sizeof(float) + 4096 + TRI_STAGE * 36 + 32 + 4 * 512 * 4 + 4 * 24 * 4;
        hipLaunchKernelGGL((leaf_brick_kernel<4, 2>), dim3(grid_for(capq[lq & 1] * 64 < bound ? capq[lq & 1] * 64 : bound, p->num_cu, 8)),
                           dim3(BLOCK), lds_b, s, p->d_code, (const Cube*)q[lq & 1]->p, (unsigned long long)capq[lq & 1],
                           p->prog.nslots, ox, oy, oz, res, m->d_tris, tcap, d_ctr);
        used_brick = true;
      } else if (p->f_leaf && p->spec_leaf_k == lk) {  // whichever W it was built for: same launch
        HIP_TRYM(launch_fn(p->f_leaf, grid_for(bound, p->num_cu, leaf_bpc), BLOCK, lds_m, s, (const uint32_t*)p->d_code, (const Cube*)q[lq & 1]->p,
                           (unsigned long long)capq[lq & 1], (int)lq, (int)p->prog.nslots, ox, oy, oz, res, m->d_tris,
                           (unsigned long)tcap, d_ctr));
      } else {
        // Ahead-of-time (interpreter) leaf kernels exist only at occupancies the compiler reaches WITHOUT scratch
        // (tests/test_kernel_resources.py reads the shipped code object): at 4 workgroups per CU (128 VGPRs) the K = 4 and
        // K = 2 interpreter builds spill, and a spilling build is not trusted (see fn_scratch_bytes).
        if (lk == 4) { if (lw == 2) LAUNCH_LEAF(4, 2); else LAUNCH_LEAF(4, 3); }
        else if (lk == 2) LAUNCH_LEAF(2, 3);
        else LAUNCH_LEAF(1, 4);
      }
#undef LAUNCH_LEAF
      HIP_TRYM(hipGetLastError());
    }
    HIP_TRYM(hipEventRecord(ev2, s));
    if (!ctr_on_host) HIP_TRYM(hipMemcpyAsync(&hc, d_ctr, sizeof(hc), hipMemcpyDeviceToHost, s));  // (else march_records_kernel wrote them)
    hipEvent_t evr = p->ev[4];
    HIP_TRYM(hipEventRecord(evr, s));
    static const bool clear_ahead = [] { const char* e = getenv("GSDF_HIP_CLEAR_AHEAD"); return !e || atoi(e) != 0; }();
    if (clear_ahead) HIP_TRYM(hipMemsetAsync(d_ctr, 0, clear_bytes, s));  // for the next mesh (or the rerun below); not waited for
    HIP_TRYM(hipEventSynchronize(evr));
    if (clear_ahead) { p->ctr_clean = clear_bytes; p->ctr_clean_stream = s; }
    if (hc.q_overflow) {  // a cube queue was too small: double and redo (exact: nothing was dropped silently)
      if (attempt >= 8) return bail(fail(GSDF_ERR_CAPACITY, "octree queue capacity exceeded"));
      qcap *= 4;
      continue;
    }
    if (two_kernel) {  // more blocks than the record arena holds: the blocks beyond it were skipped -- repeat with room for all
      const uint64_t need = ((hc.n_level[lq] << (3 * (lq - 1))) + 63) / 64;
      if (need > nblk) {
        if (attempt >= 8) return bail(fail(GSDF_ERR_CAPACITY, "cut-leaf record arena capacity exceeded"));
        p->rec_blocks = need + need / 8 + 1024;
        two_kernel = false;
        continue;
      }
    }
    if (hc.overflow) {  // triangle buffer too small: the kernel kept counting, so the exact size is known
      if (opts.max_tris) return bail(fail(GSDF_ERR_CAPACITY, "device triangle buffer capacity exceeded"));
      if (attempt >= 8) return bail(fail(GSDF_ERR_CAPACITY, "device triangle buffer capacity exceeded"));
      release_tris(m);
      want = hc.n_tris + hc.n_tris / 16 + 1024;
      continue;
    }
    break;
  }
  HIP_TRYM(hipEventElapsedTime(&ms01, ev0, ev1));
  HIP_TRYM(hipEventElapsedTime(&ms12, ev1, ev2));
  if (two_kernel) HIP_TRYM(hipEventElapsedTime(&ms13, ev1, p->ev[3]));  // the evaluating kernel alone
  uint64_t evals_prune = 0, pruned = 0;
  for (int level = levels; level >= lq; level--) {
    evals_prune += hc.n_items[level];
    // cubes that passed their test: counted apart only where "passed" and "kept by this rank" differ (prune_kernel)
    const bool dealt_here = opts.shard_count > 1 && level == ls;
    const uint64_t passed = (level <= chain_first && !dealt_here) ? hc.n_level[level] : hc.n_pass[level];
    if (hc.n_items[level]) pruned += (hc.n_items[level] - passed) << (3 * (level - 1));  // DecomposesTo(1) = 8^(level-1)
  }
  const uint64_t n_leaves = hc.n_level[lq] << (3 * (lq - 1));
  const uint64_t evals_leaf = used_brick ? hc.n_points  // distinct lattice points evaluated once each
                                         : n_leaves * (uint64_t)lk + (uint64_t)(8 - lk) * (two_kernel && lq == 3 ? n_leaves : hc.n_cont);  // evaluations actually executed (a column brick always evaluates all eight rows: leaf_eval_kernel counts nothing)
  m->st.n_tris = hc.n_tris;
  m->st.evals = evals_prune + evals_leaf;
  m->st.evals_prune = evals_prune;
  m->st.evals_leaf = evals_leaf;
  m->st.pruned_leaves = pruned;
  m->st.leaf_cubes = n_leaves;
  m->st.active_leaves = hc.n_active;
  m->st.ms_prune = ms01;
  m->st.ms_leaf = ms12;
  m->st.ms_march = two_kernel ? ms13 : ms12;
  m->st.ms_emit = two_kernel ? ms12 - ms13 : 0.0;
  m->st.cut_leaves = hc.n_cut;
  m->st.ms_total = (double)ms01 + (double)ms12;
  p->evals += m->st.evals;
  p->last_tris = hc.n_tris;
  *out = m;
  return GSDF_OK;
#undef HIP_TRYM
}

// glrender.ImageRendererSDF2.Render for a 2D program: w x h pixels over Bounds(); host outputs (either may be NULL).
extern "C" int gsdf_hip_image2(gsdf_program* p, int w, int h, float* dist_out, uint8_t* rgba_out) {
  if (!p) return fail(GSDF_ERR_BAD_ARGUMENT, "null program");
  if (!p->prog.is2d) return fail(GSDF_ERR_DIMENSION, "program is 3D, image2 called");
  if (w <= 0 || h <= 0) return fail(GSDF_ERR_BAD_ARGUMENT, "bad image size");
  HIP_TRY(hipSetDevice(p->device));
  const size_t n = (size_t)w * (size_t)h;
  // image.go:82-88: dx = sz.X/dxi ; bb.Min += (dx/2, dy/2) ; y = bb.Max.Y - j*dy ; x = i*dx + bb.Min.X
  const float szx = p->prog.bb[3] - p->prog.bb[0], szy = p->prog.bb[4] - p->prog.bb[1];
  const float dx = szx / (float)w, dy = szy / (float)h;
  const float xmin = p->prog.bb[0] + dx / 2, ymax = p->prog.bb[4];
  DevBuf dd, dc;
  HIP_TRY(dd.alloc(n * 4));
  HIP_TRY(dc.alloc(n * 4));
  const int k = p->batch_k();
  const unsigned grid = grid_for((n + k - 1) / k, p->num_cu, 8);
#define LAUNCH_IMG(KK) hipLaunchKernelGGL((image2_kernel<KK>), dim3(grid), dim3(BLOCK), p->lds_bytes(KK), p->stream, p->d_code, w, h, xmin, ymax, dx, dy, (float*)dd.p, (uint32_t*)dc.p)
  spec_aux(p);
  if (p->f_image) HIP_TRY(launch_fn(p->f_image, grid, BLOCK, p->lds_bytes(k), p->stream, (const uint32_t*)p->d_code, (int)w, (int)h, xmin, ymax, dx, dy, (float*)dd.p, (uint32_t*)dc.p));
  else
  if (k == 4) LAUNCH_IMG(4); else if (k == 2) LAUNCH_IMG(2); else LAUNCH_IMG(1);
#undef LAUNCH_IMG
  HIP_TRY(hipGetLastError());
  if (dist_out) HIP_TRY(hipMemcpyAsync(dist_out, dd.p, n * 4, hipMemcpyDeviceToHost, p->stream));
  if (rgba_out) HIP_TRY(hipMemcpyAsync(rgba_out, dc.p, n * 4, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  p->evals += n;
  return GSDF_OK;
}

// glrender.DualContourRenderer.Reset + RenderAll with DualContourLeastSquares on device.
extern "C" int gsdf_hip_mesh_dualcontour(gsdf_program* p, float res, int chiseled, int shard_rank, int shard_count, void* stream,
                                         gsdf_mesh** out) {
  if (!p || !out) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  *out = nullptr;
  if (p->prog.is2d) return fail(GSDF_ERR_DIMENSION, "program is 2D");
  if (!(res > 0) || std::isnan(res) || std::isinf(res)) return fail(GSDF_ERR_RESOLUTION, "invalid renderer cube resolution");
  if (shard_count < 1 || shard_rank < 0 || shard_rank >= shard_count) return fail(GSDF_ERR_BAD_ARGUMENT, "bad shard rank/count");
  HIP_TRY(hipSetDevice(p->device));
  hipStream_t s = stream ? (hipStream_t)stream : p->stream;
  // Reset (dual_contour.go:26-41): bounds shifted by -res/2, makeICube
  const float sub = res / 2;
  float mn[3], mx[3];
  for (int a = 0; a < 3; a++) { mn[a] = p->prog.bb[a] + -sub; mx[a] = p->prog.bb[a + 3] + -sub; }
  const float longAxis = fmaxf(mx[0] - mn[0], fmaxf(mx[1] - mn[1], mx[2] - mn[2]));
  const int levels = (int)std::ceil((float)std::log2((double)(longAxis / res))) + 1;
  if (levels <= 1) return fail(GSDF_ERR_RESOLUTION, "resolution not fine enough for marching cubes");
  // the neighbour lookup is a dense int32 index grid over the 2^(levels-1) cube lattice: 4.3 GB at 11 levels, 34 GB at 12 --
  // what one 288 GB device holds beside the rest of the workspace (13 levels would be 275 GB)
  if (levels > 12) return fail(GSDF_ERR_RESOLUTION, "dual contouring lattice too large: more than 12 octree levels");
  const int nshift = levels - 1;
  const uint64_t ncell = (uint64_t)1 << (3 * nshift);
  const float ox = mn[0], oy = mn[1], oz = mn[2];
  // multi-GPU: z-slabs. A rank owns the quads of cubes with z in [zown_lo, zown_hi); their vertices need the cubes of
  // [zown_lo-1, zown_hi) placed, which in turn need the distances and normals of [zown_lo-1, zown_hi+1). Every stage is
  // a pure function of the lattice cell, so the halo is recomputed instead of exchanged (no data-path collective).
  const unsigned nz = 1u << nshift;
  uint32_t zown_lo = 0, zown_hi = 0;
  gsdf_hip_slab_range(nz, (uint32_t)shard_rank, (uint32_t)shard_count, &zown_lo, &zown_hi);
  const unsigned zlo = zown_lo > 0 ? zown_lo - 1 : 0, zhi = zown_hi < nz ? zown_hi + 1 : nz;
  const uint64_t nslab = (uint64_t)(zhi - zlo) << (2 * nshift);

  gsdf_mesh* m = new (std::nothrow) gsdf_mesh();
  if (!m) return fail(GSDF_ERR_BAD_ARGUMENT, "out of memory");
  m->device = p->device; m->stream = s;
  m->st.levels = levels; m->st.res = res;
  m->st.origin[0] = ox; m->st.origin[1] = oy; m->st.origin[2] = oz;
  auto bail = [&](int code) { gsdf_hip_mesh_destroy(m); return code; };
#define HIP_TRYM(expr)                                                                                          \
  do {                                                                                                          \
    hipError_t _e = (expr);                                                                                     \
    if (_e != hipSuccess) return bail(fail(GSDF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e))); \
  } while (0)
  for (auto& e : p->ev)
    if (!e) HIP_TRYM(hipEventCreate(&e));
  // Workspace lives in the program handle (grow-only): the index grid alone is 4.3 GB at 1024^3 cells, and a
  // hipMalloc/hipFree pair of that size per mesh cost more wall time than the whole device pass.
  gsdf_program::Arena &grid = p->dc_grid, &d2 = p->dc_dist, &f2 = p->dc_fv, &n2 = p->dc_nrm, &e2 = p->dc_edge;
  HIP_TRYM(grid.ensure(ncell * sizeof(int)));
  p->ctr_settle();  // (the octree mesher's clear-ahead)
  HIP_TRYM(p->ctr.ensure(sizeof(MeshCounters) > sizeof(DCCounters) ? sizeof(MeshCounters) : sizeof(DCCounters)));
  DCCounters* d_ctr = (DCCounters*)p->ctr.p;
  const int lk = p->batch_k();
  // Kept cubes hug the surface: ~ c * n^2 of the n^3 lattice. Start from the previous pass on this handle, else from
  // 12 n^2 (an overflow repeats the full-lattice origin pass, so be generous: 84 B per cube).
  uint64_t ccap = p->last_dc_cubes ? p->last_dc_cubes + p->last_dc_cubes / 8 + 4096 : (uint64_t)12 << (2 * nshift);
  if (ccap < (1u << 20)) ccap = 1u << 20;
  DCCounters hc{};
  const float h = (chiseled ? (float)1e-4 : (float)2e-8) * 0.5f;  // NormalsCentralDiff: step *= 0.5
  const float sqrtLambda = chiseled ? (float)(std::sqrt(1e-5) * 1e-4) : (float)std::sqrt(1e-5);
  for (int attempt = 0;; attempt++) {
    if (ccap > nslab) ccap = nslab;
    const uint64_t ecap = 3 * ccap, tcap = 2 * ecap;
    HIP_TRYM(p->q0.ensure(ccap * sizeof(Cube)));
    HIP_TRYM(d2.ensure(ccap * sizeof(float4)));
    HIP_TRYM(f2.ensure(ccap * 12));
    HIP_TRYM(n2.ensure(ccap * 36));
    HIP_TRYM(e2.ensure(ecap * sizeof(unsigned)));
    if (!m->d_tris || m->cap < tcap) {
      pool_give(p->device, m->d_tris, m->cap);
      m->d_tris = pool_take(p->device, tcap, &m->cap);
      if (!m->d_tris) { HIP_TRYM(hipMalloc((void**)&m->d_tris, tcap * 36)); m->cap = tcap; }
    }
    HIP_TRYM(hipMemsetAsync(d_ctr, 0, sizeof(DCCounters), s));
    if (shard_count > 1) HIP_TRYM(hipMemsetAsync(grid.p, 0xff, ncell * sizeof(int), s));  // cells outside the slab read as empty
    HIP_TRYM(hipEventRecord(p->ev[0], s));
    const unsigned g1 = grid_for((nslab + lk - 1) / lk, p->num_cu, 32);
#define LAUNCH_O(KK, WW) hipLaunchKernelGGL((dc_origin_kernel<KK, WW>), dim3(g1), dim3(BLOCK), p->lds_bytes(KK) + 32, s, p->d_code, p->prog.nslots, nshift, ox, oy, oz, res, (int*)grid.p, (Cube*)p->q0.p, (unsigned long long)ccap, zlo, zhi, ub, eb[0], eb[1], eb[2], eb[3], eb[4], eb[5], t0[0], t0[1], t0[2], tn[0], tn[1], tn[2], d_ctr)
    spec_aux(p);
    const int ub = p->prog.has_exact_bb ? 1 : 0;
    const float* eb = p->prog.exact_bb;
    // tile range of the origin sweep (8 x 8 x 4K cells per tile, z tiles counted from zlo)
    const unsigned ncell1 = 1u << nshift, tzk = 4u * (unsigned)lk;
    unsigned t0[3] = {0, 0, 0}, tn[3] = {(ncell1 + 7u) >> 3, (ncell1 + 7u) >> 3, (zhi - zlo + tzk - 1u) / tzk};
    if (ub) {
      const float org[3] = {ox, oy, oz}, grow = res * 2 * 1.001f + 2 * res;
      const unsigned lo_lim[3] = {0, 0, zlo}, hi_lim[3] = {ncell1, ncell1, zhi}, tsz[3] = {8, 8, tzk};
      for (int a = 0; a < 3; a++) {
        double c0 = std::floor(((double)eb[a] - grow - org[a]) / res) - 1, c1 = std::ceil(((double)eb[a + 3] + grow - org[a]) / res) + 2;
        if (c0 < lo_lim[a]) c0 = lo_lim[a];
        if (c1 > hi_lim[a]) c1 = hi_lim[a];
        if (c1 <= c0) { c0 = lo_lim[a]; c1 = lo_lim[a]; }  // nothing of the box in this slab
        const unsigned first = ((unsigned)c0 - lo_lim[a]) / tsz[a], last = ((unsigned)c1 - lo_lim[a] + tsz[a] - 1) / tsz[a];
        t0[a] = first;
        tn[a] = last > first ? last - first : 0;
      }
    }
    if (p->f_dc_origin) HIP_TRYM(launch_fn(p->f_dc_origin, g1, BLOCK, p->lds_bytes(lk) + 32, s, (const uint32_t*)p->d_code, (int)p->prog.nslots, (int)nshift, ox, oy, oz, res, (int*)grid.p, (Cube*)p->q0.p, (unsigned long long)ccap, (unsigned)zlo, (unsigned)zhi, ub, eb[0], eb[1], eb[2], eb[3], eb[4], eb[5], t0[0], t0[1], t0[2], tn[0], tn[1], tn[2], d_ctr));
    else
    if (lk == 4) LAUNCH_O(4, 3);  // <4, 4> needs scratch: not built (see fn_scratch_bytes)
    else if (lk == 2) { if (p->sweep_waves(2) == 4) LAUNCH_O(2, 4); else LAUNCH_O(2, 3); }
    else LAUNCH_O(1, 4);
#undef LAUNCH_O
    HIP_TRYM(hipGetLastError());
    if (p->lds_bytes(4) > 150 * 1024) return bail(fail(GSDF_ERR_BAD_TREE, "tree needs too much LDS scratch for the dual contouring edge pass"));
    if (p->f_dc_edges) HIP_TRYM(launch_fn(p->f_dc_edges, grid_for(ccap, p->num_cu, 8), BLOCK, p->lds_bytes(4), s, (const uint32_t*)p->d_code, (const Cube*)p->q0.p,
                       (unsigned long long)ccap, ox, oy, oz, res, (float4*)d2.p, (float*)f2.p, (unsigned*)e2.p, (unsigned long long)ecap, d_ctr));
    else
    hipLaunchKernelGGL(dc_edges_kernel, dim3(grid_for(ccap, p->num_cu, 8)), dim3(BLOCK), p->lds_bytes(4), s, p->d_code, (const Cube*)p->q0.p,
                       (unsigned long long)ccap, ox, oy, oz, res, (float4*)d2.p, (float*)f2.p, (unsigned*)e2.p, (unsigned long long)ecap, d_ctr);
    HIP_TRYM(hipGetLastError());
    if (p->f_dc_normals) HIP_TRYM(launch_fn(p->f_dc_normals, grid_for(ecap, p->num_cu, 8), BLOCK, p->lds_bytes(2), s, (const uint32_t*)p->d_code, (const Cube*)p->q0.p,
                       (const float4*)d2.p, (const unsigned*)e2.p, (unsigned long long)ecap, ox, oy, oz, res, h, (float*)n2.p, d_ctr));
    else
    hipLaunchKernelGGL(dc_normals_kernel, dim3(grid_for(ecap, p->num_cu, 8)), dim3(BLOCK), p->lds_bytes(2), s, p->d_code, (const Cube*)p->q0.p,
                       (const float4*)d2.p, (const unsigned*)e2.p, (unsigned long long)ecap, ox, oy, oz, res, h, (float*)n2.p, d_ctr);
    HIP_TRYM(hipGetLastError());
    hipLaunchKernelGGL(dc_place_kernel, dim3(grid_for(ccap * 4, p->num_cu, 16)), dim3(DC_BLOCK), 0, s, (const Cube*)p->q0.p,
                       (unsigned long long)ccap, (const float4*)d2.p, (const int*)grid.p, (const float*)n2.p, nshift, ox, oy, oz, res,
                       sqrtLambda, (float*)f2.p, zown_hi, d_ctr);
    HIP_TRYM(hipGetLastError());
    hipLaunchKernelGGL(dc_quads_kernel, dim3(grid_for(ecap, p->num_cu, 8)), dim3(BLOCK), 0, s, (const Cube*)p->q0.p, (const float4*)d2.p,
                       (const unsigned*)e2.p, (unsigned long long)ecap, (const int*)grid.p, (const float*)f2.p, nshift, zown_lo, zown_hi,
                       m->d_tris, (unsigned long long)m->cap, d_ctr);
    HIP_TRYM(hipGetLastError());
    HIP_TRYM(hipEventRecord(p->ev[1], s));
    HIP_TRYM(hipMemcpyAsync(&hc, d_ctr, sizeof(hc), hipMemcpyDeviceToHost, s));
    HIP_TRYM(hipStreamSynchronize(s));
    if (hc.q_overflow || hc.t_overflow) {
      if (attempt >= 8 || ccap >= nslab) return bail(fail(GSDF_ERR_CAPACITY, "dual contouring queue capacity exceeded"));
      // the origin pass keeps counting past the capacity, so the exact number of kept cubes is known
      ccap = hc.n_cubes > ccap ? hc.n_cubes + hc.n_cubes / 16 + 4096 : ccap * 2;
      continue;
    }
    break;
  }
  float ms = 0;
  HIP_TRYM(hipEventElapsedTime(&ms, p->ev[0], p->ev[1]));
  p->last_dc_cubes = hc.n_cubes;
  m->st.n_tris = 2 * hc.n_tris;  // quads -> 2 triangles
  m->st.evals = hc.n_origin_evals + 4 * hc.n_cubes + 6 * hc.n_edges;
  m->st.evals_prune = hc.n_origin_evals;  // of the nslab lattice cells; the rest lay outside the exact box by > 2 res
  m->st.evals_leaf = 4 * hc.n_cubes + 6 * hc.n_edges;
  m->st.pruned_leaves = nslab - hc.n_cubes;
  m->st.leaf_cubes = hc.n_cubes;
  m->st.active_leaves = hc.n_edges;
  m->st.ms_total = ms;
  p->evals += m->st.evals;
  *out = m;
  return GSDF_OK;
#undef HIP_TRYM
}

// glrender.FlatRenderer (flatrenderer.go:36-256) on device: Reset's lattice, evalGrid into a dense grid in HBM,
// ReadTriangles as one marching-cubes pass over every cube. Multi-GPU: z-slabs of cubes like the reference's goroutines
// (:120-122); a rank evaluates the planes its cubes touch (one shared plane per boundary is recomputed, nothing exchanged).
extern "C" int gsdf_hip_mesh_flat(gsdf_program* p, float res, int shard_rank, int shard_count, void* stream, gsdf_mesh** out) {
  if (!p || !out) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  *out = nullptr;
  if (p->prog.is2d) return fail(GSDF_ERR_DIMENSION, "program is 2D");
  if (!(res > 0) || std::isnan(res) || std::isinf(res)) return fail(GSDF_ERR_RESOLUTION, "invalid renderer cube resolution");
  if (shard_count < 1 || shard_rank < 0 || shard_rank >= shard_count) return fail(GSDF_ERR_BAD_ARGUMENT, "bad shard rank/count");
  HIP_TRY(hipSetDevice(p->device));
  hipStream_t s = stream ? (hipStream_t)stream : p->stream;
  // Reset (:36-80)
  float mn[3], mx[3];
  scale_centered(p->prog.bb, 1.01f, mn, mx);
  double nd[3];
  for (int a = 0; a < 3; a++) nd[a] = (double)std::ceil((mx[a] - mn[a]) / res);
  if (!(nd[0] > 0) || !(nd[1] > 0) || !(nd[2] > 0)) return fail(GSDF_ERR_RESOLUTION, "resolution not fine enough for marching cubes");
  if (nd[0] > 65534 || nd[1] > 65534 || nd[2] > 1e9 || (nd[0] + 1) * (nd[1] + 1) >= 4294967296.0)
    return fail(GSDF_ERR_RESOLUTION, "resolution too fine for the flat renderer's lattice");
  const unsigned nx = (unsigned)nd[0], ny = (unsigned)nd[1], nz = (unsigned)nd[2];
  const unsigned sx = nx + 1, sy = ny + 1;
  const uint64_t sxy = (uint64_t)sx * sy;
  const uint64_t pxy = (uint64_t)FLAT_PITCH(sx) * sy;  // the grid's rows are padded to 256 bytes
  const float ox = mn[0], oy = mn[1], oz = mn[2];
  // this rank's cubes in z and the lattice planes they touch
  uint32_t c0 = 0, c1 = 0;
  gsdf_hip_slab_range(nz, (uint32_t)shard_rank, (uint32_t)shard_count, &c0, &c1);
  const unsigned ncz = c1 - c0, nk = ncz ? ncz + 1 : 0;

  gsdf_mesh* m = new (std::nothrow) gsdf_mesh();
  if (!m) return fail(GSDF_ERR_BAD_ARGUMENT, "out of memory");
  m->device = p->device; m->stream = s;
  m->st.levels = 0; m->st.res = res;
  m->st.origin[0] = ox; m->st.origin[1] = oy; m->st.origin[2] = oz;
  auto bail = [&](int code) { gsdf_hip_mesh_destroy(m); return code; };
#define HIP_TRYM(expr)                                                                                          \
  do {                                                                                                          \
    hipError_t _e = (expr);                                                                                     \
    if (_e != hipSuccess) return bail(fail(GSDF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e))); \
  } while (0)
  if (ncz == 0) { *out = m; return GSDF_OK; }  // more ranks than cube planes: nothing for this one
  for (auto& e : p->ev)
    if (!e) HIP_TRYM(hipEventCreate(&e));
  p->ctr_settle();  // (the octree mesher's clear-ahead)
  HIP_TRYM(p->ctr.ensure(sizeof(MeshCounters)));
  MeshCounters* d_ctr = (MeshCounters*)p->ctr.p;
  {
    // refuse up front what cannot fit (a failed multi-hundred-GB hipMalloc is slow and leaves the allocator fragmented)
    size_t mfree = 0, mtotal = 0;
    const double need = (double)pxy * (double)nk * sizeof(float);
    if (hipMemGetInfo(&mfree, &mtotal) == hipSuccess && need > (double)mfree + (double)p->flat_grid.cap)
      return bail(fail(GSDF_ERR_CAPACITY, "flat renderer: the distance grid (" + std::to_string((unsigned long long)(need / 1e9)) +
                                              " GB) does not fit the device memory; use the octree renderer at this resolution"));
  }
  if (p->flat_grid.ensure(pxy * nk * sizeof(float)) != hipSuccess) {
    (void)hipGetLastError();
    return bail(fail(GSDF_ERR_CAPACITY, "flat renderer: no device memory for the distance grid (" + std::to_string(pxy * nk * 4) + " bytes)"));
  }
  float* grid = (float*)p->flat_grid.p;
  // two bit planes per lattice plane beside it (1/32 + 1/32 of the grid's size): "d < 0" and "|d| <= 2 sqrt3 res" per corner,
  // in whole words per flat_grid_kernel pass (BLOCK corners = BLOCK / 64 words)
  const unsigned wpp = (unsigned)((sxy + (uint64_t)BLOCK - 1) / (uint64_t)BLOCK) * (BLOCK / 64);
  if (p->flat_bits.ensure((size_t)2 * wpp * nk * sizeof(unsigned long long)) != hipSuccess) {
    (void)hipGetLastError();
    return bail(fail(GSDF_ERR_CAPACITY, "flat renderer: no device memory for the lattice's bit planes"));
  }
  unsigned long long* negbits = (unsigned long long*)p->flat_bits.p;
  unsigned long long* nearbits = negbits + (size_t)wpp * nk;
  const int ek = p->batch_k();
  spec_aux(p);
  HIP_TRYM(hipEventRecord(p->ev[0], s));
  {
    const uint64_t npass = ((sxy + (uint64_t)BLOCK - 1) / (uint64_t)BLOCK) * ((nk + ek - 1) / ek);
    static const int bpc = [] { const char* e = getenv("GSDF_HIP_EVAL_BPC"); return e ? atoi(e) : 64; }();
    const uint64_t gmax = (uint64_t)p->num_cu * (uint64_t)(bpc > 0 ? bpc : 64);
    const unsigned g = (unsigned)(npass < gmax ? npass : gmax);
    if (p->f_flat_grid) HIP_TRYM(launch_fn(p->f_flat_grid, g, BLOCK, p->lds_bytes(ek), s, (const uint32_t*)p->d_code, ox, oy, oz, res, sx, sy, c0, nk, grid, negbits, nearbits));
#define LAUNCH_FG(KK, WW) hipLaunchKernelGGL((flat_grid_kernel<KK, WW>), dim3(g), dim3(BLOCK), p->lds_bytes(KK), s, p->d_code, ox, oy, oz, res, sx, sy, c0, nk, grid, negbits, nearbits)
    else if (ek == 4) { if (p->sweep_waves(4) == 4) LAUNCH_FG(4, 4); else LAUNCH_FG(4, 3); }
    else if (ek == 2) { if (p->sweep_waves(2) == 4) LAUNCH_FG(2, 4); else LAUNCH_FG(2, 3); }
    else LAUNCH_FG(1, 4);
#undef LAUNCH_FG
    HIP_TRYM(hipGetLastError());
  }
  HIP_TRYM(hipEventRecord(p->ev[1], s));
  // ReadTriangles: the triangle count is not known in advance; the pass is cheap (HBM-bound over the grid), so a
  // buffer that turns out too small is replaced by one of the exact size and only this pass is repeated.
  MeshCounters hc{};
  uint64_t want = p->last_tris ? p->last_tris + p->last_tris / 16 + 1024 : (uint64_t)1 << 20;
  float ms_march = 0;
  for (int attempt = 0;; attempt++) {
    if (!m->d_tris) {
      m->d_tris = pool_take(p->device, want, &m->cap);
      if (!m->d_tris) { HIP_TRYM(hipMalloc((void**)&m->d_tris, want * 36)); m->cap = want; }
    }
    HIP_TRYM(hipMemsetAsync(d_ctr, 0, sizeof(MeshCounters), s));
    HIP_TRYM(hipEventRecord(p->ev[2], s));
    // default: the marching pass over the bit planes (flat_cut_scan_kernel + flat_march_list_kernel); GSDF_HIP_FLAT_STREAM=1:
    // the pass that streams the float grid (flat_march_kernel, the kernel of rounds 1-2), kept for comparison -- same triangles
    static const bool stream_march = [] { const char* e = getenv("GSDF_HIP_FLAT_STREAM"); return e && atoi(e) != 0; }();
    if (nx >= 65536u || ny >= 65536u || (uint64_t)c0 + ncz >= 65536u)  // record coordinates are 16-bit
      return bail(fail(GSDF_ERR_RESOLUTION, "resolution too fine for the flat renderer's lattice"));
    if (stream_march) {
      const uint64_t npass = (uint64_t)((nx + FLAT_TX - 1) / FLAT_TX) * ((ny + FLAT_ROWS - 1) / FLAT_ROWS) * ncz;  // wave passes: FLAT_TX x FLAT_ROWS cubes each
      if ((double)npass + 1e6 >= 4294967296.0) return bail(fail(GSDF_ERR_RESOLUTION, "resolution too fine for the flat renderer's lattice"));
      // four workgroups per CU are resident (36 KB of LDS each): a grid of exactly those, ~200 passes per wave, measured best
      // (0.63 ms; 8 per CU 0.66, 32 per CU 0.74, 6 per CU 0.82 -- the stride between a wave's passes matters)
      static const int mbpc = [] { const char* e = getenv("GSDF_HIP_FLAT_BPC"); return e ? atoi(e) : 4; }();  // tuning knob
      const uint64_t gmax = (uint64_t)p->num_cu * (uint64_t)(mbpc > 0 ? mbpc : 4);
      const uint64_t nwg = (npass + 3) / 4;
      const size_t lds = (size_t)256 * 16 + (size_t)4 * FLAT_WAVE_RECS * REC_WORDS * 4 + (size_t)4 * 5 * FLAT_WAVE_RECS * 2;
      hipLaunchKernelGGL(flat_march_kernel, dim3((unsigned)(nwg < gmax ? (nwg ? nwg : 1) : gmax)), dim3(BLOCK), lds, s, (const float*)grid, nx, ny, ncz, c0,
                         ox, oy, oz, res, m->d_tris, (uint64_t)m->cap, d_ctr);
    } else {
      // a cut cube has at least one triangle: a list of the triangle buffer's capacity overflows only if that does
      if (p->flat_list.ensure((size_t)m->cap * sizeof(unsigned long long)) != hipSuccess) {
        (void)hipGetLastError();
        return bail(fail(GSDF_ERR_CAPACITY, "flat renderer: no device memory for the list of cut cubes"));
      }
      unsigned long long* list = (unsigned long long*)p->flat_list.p;
      const uint64_t npass = (((uint64_t)sx * ny + 4095u) >> 12) * ncz;  // wave passes of the scan: 4096 cubes each
      static const int sbpc = [] { const char* e = getenv("GSDF_HIP_FLAT_SCAN_BPC"); return e ? atoi(e) : 16; }();   // tuning knobs
      static const int lbpc = [] { const char* e = getenv("GSDF_HIP_FLAT_LIST_BPC"); return e ? atoi(e) : 6; }();
      const uint64_t gmax = (uint64_t)p->num_cu * (uint64_t)(sbpc > 0 ? sbpc : 16), nwg = (npass + 3) / 4;
      hipLaunchKernelGGL(flat_cut_scan_kernel, dim3((unsigned)(nwg < gmax ? (nwg ? nwg : 1) : gmax)), dim3(BLOCK), 0, s, (const unsigned long long*)negbits,
                         (const unsigned long long*)nearbits, wpp, nx, ny, ncz, list, (uint64_t)m->cap, d_ctr);
      hipLaunchKernelGGL(flat_march_list_kernel, dim3((unsigned)p->num_cu * (unsigned)(lbpc > 0 ? lbpc : 6)), dim3(BLOCK), FLATB_LDS_BYTES, s, (const float*)grid,
                         (const unsigned long long*)list, (uint64_t)m->cap, nx, ny, c0, ox, oy, oz, res, m->d_tris, (uint64_t)m->cap, d_ctr);
    }
    HIP_TRYM(hipGetLastError());
    HIP_TRYM(hipEventRecord(p->ev[3], s));
    HIP_TRYM(hipMemcpyAsync(&hc, d_ctr, sizeof(hc), hipMemcpyDeviceToHost, s));
    HIP_TRYM(hipStreamSynchronize(s));
    if (hc.overflow) {
      // The float-stream pass keeps counting: n_tris is exact. So does the bit-plane pass as long as its cut-cube list held
      // every cut cube (n_cut <= capacity); if not, n_cut is still exact and n_tris covers the listed cubes only: a cut cube
      // has 1 to 5 triangles (2.2 on the configs' surfaces), so room for 3 per cut cube holds the list for certain and
      // the triangles nearly always -- one more exact rerun otherwise.
      if (attempt >= 5) return bail(fail(GSDF_ERR_CAPACITY, "device triangle buffer capacity exceeded"));
      const bool list_short = hc.n_cut > m->cap;
      pool_give(p->device, m->d_tris, m->cap);
      m->d_tris = nullptr; m->cap = 0;
      want = hc.n_tris + hc.n_tris / 16 + 1024;
      if (list_short && want < 3 * hc.n_cut) want = 3 * hc.n_cut;
      continue;
    }
    break;
  }
  float ms_grid = 0;
  HIP_TRYM(hipEventElapsedTime(&ms_march, p->ev[2], p->ev[3]));  // the last (successful) marching pass
  HIP_TRYM(hipEventElapsedTime(&ms_grid, p->ev[0], p->ev[1]));
  m->st.n_tris = hc.n_tris;
  m->st.evals = sxy * nk;  // FlatRenderer.Evaluations(): every lattice corner once
  m->st.evals_prune = 0;
  m->st.evals_leaf = m->st.evals;
  m->st.pruned_leaves = 0;
  m->st.leaf_cubes = (uint64_t)nx * ny * ncz;
  m->st.active_leaves = hc.n_active;
  m->st.ms_prune = 0;
  m->st.ms_leaf = ms_grid;
  m->st.ms_march = ms_march;
  m->st.ms_total = (double)ms_grid + (double)ms_march;
  p->evals += m->st.evals;
  p->last_tris = hc.n_tris;
  *out = m;
  return GSDF_OK;
#undef HIP_TRYM
}

// Pure host helper (no GPU): owner rank of the brick (x,y,z) under the multi-GPU partition that
// gsdf_hip_mesh_octree applies on device (same function, SURVEY 8(e): no data-path collective).
// Pure host helper (no GPU): the z-slab [lo, hi) of n lattice planes that rank `rank` of `count` owns in the flat
// renderer and in dual contouring -- contiguous, disjoint, covering [0, n); ranks beyond n get empty slabs.
extern "C" void gsdf_hip_slab_range(uint32_t n, uint32_t rank, uint32_t count, uint32_t* lo, uint32_t* hi) {
  if (!count || rank >= count) { if (lo) *lo = 0; if (hi) *hi = 0; return; }
  if (lo) *lo = (uint32_t)(((uint64_t)n * (uint64_t)rank) / (uint64_t)count);
  if (hi) *hi = (uint32_t)(((uint64_t)n * (uint64_t)(rank + 1)) / (uint64_t)count);
}

extern "C" uint32_t gsdf_hip_brick_owner(uint32_t x, uint32_t y, uint32_t z, uint32_t count) {
  return count ? brick_owner(x, y, z, count) : 0;
}

// ---------------------------------------------------------------------------------------------
// multi-GPU: RCCL all-gatherv of the ranks' triangle buffers
// ---------------------------------------------------------------------------------------------
// One process per GPU. The mesher shards with no data-path collective (brick_owner / z-slabs); the one exchange is the
// final variable-length gather, done here on RCCL directly so that a Go (or C) caller of this ABI has the multi-GPU path
// without any Python: ncclAllGather of the counts, then ONE ncclGroup of ncclBroadcast's, root r sending exactly
// count_r * 36 bytes straight from its mesh's triangle buffer into every rank's output at offset sum(count_<r) -- no
// padding, no staging copies. librccl is loaded at first use (dlopen by soname: the process-wide copy), so the library
// itself has no link-time dependency on it and single-GPU users never load it.
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {
struct RcclApi {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};
RcclApi* rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) { api.h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (api.h) break; }
    if (!api.h) { api.err = std::string("librccl not found: ") + (dlerror() ? dlerror() : ""); return; }
#define GSDF_RCCL_SYM(field, sym)                                                          \
  api.field = (decltype(api.field))dlsym(api.h, sym);                                      \
  if (!api.field && api.err.empty()) api.err = std::string("librccl lacks ") + sym;
    GSDF_RCCL_SYM(GetUniqueId, "ncclGetUniqueId") GSDF_RCCL_SYM(CommInitRank, "ncclCommInitRank") GSDF_RCCL_SYM(CommDestroy, "ncclCommDestroy")
    GSDF_RCCL_SYM(AllGather, "ncclAllGather") GSDF_RCCL_SYM(AllReduce, "ncclAllReduce") GSDF_RCCL_SYM(Broadcast, "ncclBroadcast")
    GSDF_RCCL_SYM(Send, "ncclSend") GSDF_RCCL_SYM(Recv, "ncclRecv")
    GSDF_RCCL_SYM(GroupStart, "ncclGroupStart") GSDF_RCCL_SYM(GroupEnd, "ncclGroupEnd") GSDF_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef GSDF_RCCL_SYM
  });
  return &api;
}
}  // namespace

struct gsdf_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  hipStream_t stream = nullptr;
  unsigned long long* d_counts = nullptr;  // [world + 1]: the gathered counts, then this rank's own
  unsigned long long* h_counts = nullptr;  // pinned mirror
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};  // around the counts exchange and the payload, on `stream`
};

// A gather whose payload is on its way (gsdf_hip_mesh_gatherv_start): the result mesh, the counts, and the events that time it.
struct gsdf_gather {
  gsdf_comm* c = nullptr;
  gsdf_mesh* g = nullptr;  // result (NULL on the ranks that receive nothing)
  std::vector<uint64_t> counts;
  gsdf_gather_stats st{};
  hipEvent_t ev_payload0 = nullptr, ev_payload1 = nullptr;
  float ms_counts = 0;
};

#define RCCL_TRY(expr)                                                                                                   \
  do {                                                                                                                   \
    ncclResult_t _r = (expr);                                                                                            \
    if (_r != ncclSuccess) return fail(GSDF_ERR_HIP, std::string(#expr) + ": " + (R->GetErrorString ? R->GetErrorString(_r) : "rccl error")); \
  } while (0)

extern "C" int gsdf_hip_comm_unique_id(uint8_t id[GSDF_COMM_ID_BYTES]) {
  if (!id) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  RcclApi* R = rccl();
  if (!R->err.empty()) return fail(GSDF_ERR_HIP, R->err);
  static_assert(sizeof(ncclUniqueId) <= GSDF_COMM_ID_BYTES, "ncclUniqueId larger than GSDF_COMM_ID_BYTES");
  ncclUniqueId u;
  RCCL_TRY(R->GetUniqueId(&u));
  std::memset(id, 0, GSDF_COMM_ID_BYTES);
  std::memcpy(id, &u, sizeof u);
  return GSDF_OK;
}

extern "C" void gsdf_hip_comm_destroy(gsdf_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->comm && rccl()->CommDestroy) (void)rccl()->CommDestroy(c->comm);
  if (c->d_counts) (void)hipFree(c->d_counts);
  if (c->h_counts) (void)hipHostFree(c->h_counts);
  for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

extern "C" int gsdf_hip_comm_create(const uint8_t id[GSDF_COMM_ID_BYTES], int rank, int world, gsdf_comm** out) {
  if (!id || !out) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world) return fail(GSDF_ERR_BAD_ARGUMENT, "bad rank / world size");
  RcclApi* R = rccl();
  if (!R->err.empty()) return fail(GSDF_ERR_HIP, R->err);
  gsdf_comm* c = new (std::nothrow) gsdf_comm();
  if (!c) return fail(GSDF_ERR_BAD_ARGUMENT, "out of memory");
  c->rank = rank; c->world = world;
  auto bail = [&](int code) { gsdf_hip_comm_destroy(c); return code; };
  if (hipGetDevice(&c->device) != hipSuccess) return bail(fail(GSDF_ERR_HIP, "hipGetDevice failed"));
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(GSDF_ERR_HIP, "hipStreamCreate failed"));
  if (hipMalloc((void**)&c->d_counts, sizeof(unsigned long long) * (size_t)(world + 1)) != hipSuccess) return bail(fail(GSDF_ERR_HIP, "hipMalloc(counts) failed"));
  if (hipHostMalloc((void**)&c->h_counts, sizeof(unsigned long long) * (size_t)(world + 1), hipHostMallocDefault) != hipSuccess) return bail(fail(GSDF_ERR_HIP, "hipHostMalloc(counts) failed"));
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  ncclResult_t r = R->CommInitRank(&c->comm, world, u, rank);
  if (r != ncclSuccess) { c->comm = nullptr; return bail(fail(GSDF_ERR_HIP, std::string("ncclCommInitRank: ") + R->GetErrorString(r))); }
  *out = c;
  return GSDF_OK;
}
extern "C" int gsdf_hip_comm_rank(const gsdf_comm* c) { return c ? c->rank : -1; }
extern "C" int gsdf_hip_comm_world(const gsdf_comm* c) { return c ? c->world : 0; }

// Sum of `n` host uint64 values over all ranks, in place (Evaluations(), TotalPruned(), triangle totals).
extern "C" int gsdf_hip_comm_allreduce_sum_u64(gsdf_comm* c, uint64_t* vals, size_t n) {
  if (!c || !vals) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  if (n == 0) return GSDF_OK;
  RcclApi* R = rccl();
  HIP_TRY(hipSetDevice(c->device));
  unsigned long long* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, n * 8));
  int rc = GSDF_OK;
  do {
    if (hipMemcpyAsync(d, vals, n * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess) { rc = fail(GSDF_ERR_HIP, "H2D copy failed"); break; }
    ncclResult_t r = R->AllReduce(d, d, n, ncclUint64, ncclSum, c->comm, c->stream);
    if (r != ncclSuccess) { rc = fail(GSDF_ERR_HIP, std::string("ncclAllReduce: ") + R->GetErrorString(r)); break; }
    if (hipMemcpyAsync(vals, d, n * 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { rc = fail(GSDF_ERR_HIP, "D2H copy failed"); break; }
  } while (0);
  (void)hipFree(d);
  return rc;
}

// Gather of triangle buffers. mode ALL: every rank ends up with the triangles of all ranks in rank order (ncclAllGather of the
// counts, then ONE group of ncclBroadcast's, root r sending exactly count_r * 36 bytes straight out of its mesh into every
// rank's result at offset sum(count_<r): no padding, no staging copies). mode ROOT: only `root` does (one group of ncclSend /
// ncclRecv: a rank's link carries its own triangles only, 1/world of what ALL puts on it). mode NONE: the counts only -- every
// rank keeps its shard (a caller that writes per-rank files, or consumes the shards where they are). _start returns once the
// counts are known and the payload is enqueued on the communicator's stream: the caller may mesh the next part while it
// moves (the source mesh must stay alive until _wait); _wait returns the result, a mesh like any other (read / host views /
// STL / destroy as usual), NULL on ranks that received nothing.
extern "C" int gsdf_hip_mesh_gatherv_start(const gsdf_mesh* m, gsdf_comm* c, int mode, int root, gsdf_gather** out) {
  if (!m || !c || !out) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  *out = nullptr;
  if (mode != GSDF_GATHER_ALL && mode != GSDF_GATHER_ROOT && mode != GSDF_GATHER_NONE) return fail(GSDF_ERR_BAD_ARGUMENT, "bad gather mode");
  if (mode == GSDF_GATHER_ROOT && (root < 0 || root >= c->world)) return fail(GSDF_ERR_BAD_ARGUMENT, "bad root rank");
  if (m->host_out) return fail(GSDF_ERR_BAD_ARGUMENT, "gatherv needs device-resident triangles (host_output meshes live in host memory)");
  if (m->device != c->device) return fail(GSDF_ERR_BAD_ARGUMENT, "mesh and communicator are on different devices");
  RcclApi* R = rccl();
  HIP_TRY(hipSetDevice(c->device));
  for (auto& e : c->ev) if (!e) HIP_TRY(hipEventCreate(&e));
  const int W = c->world;
  // 1. counts
  HIP_TRY(hipEventRecord(c->ev[0], c->stream));
  c->h_counts[W] = m->st.n_tris;
  HIP_TRY(hipMemcpyAsync(c->d_counts + W, c->h_counts + W, 8, hipMemcpyHostToDevice, c->stream));
  RCCL_TRY(R->AllGather(c->d_counts + W, c->d_counts, 1, ncclUint64, c->comm, c->stream));
  HIP_TRY(hipMemcpyAsync(c->h_counts, c->d_counts, 8 * (size_t)W, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipEventRecord(c->ev[1], c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  gsdf_gather* p = new (std::nothrow) gsdf_gather();
  if (!p) return fail(GSDF_ERR_BAD_ARGUMENT, "out of memory");
  p->c = c;
  p->counts.assign(c->h_counts, c->h_counts + W);
  (void)hipEventElapsedTime(&p->ms_counts, c->ev[0], c->ev[1]);
  uint64_t total = 0;
  for (int r = 0; r < W; r++) total += p->counts[(size_t)r];
  auto bail = [&](int code) { if (p->g) gsdf_hip_mesh_destroy(p->g); if (p->ev_payload0) (void)hipEventDestroy(p->ev_payload0); if (p->ev_payload1) (void)hipEventDestroy(p->ev_payload1); delete p; return code; };
  const bool receives = mode == GSDF_GATHER_ALL || (mode == GSDF_GATHER_ROOT && c->rank == root);
  const uint64_t mine = p->counts[(size_t)c->rank];
  if (receives) {
    gsdf_mesh* g = new (std::nothrow) gsdf_mesh();
    if (!g) return bail(fail(GSDF_ERR_BAD_ARGUMENT, "out of memory"));
    p->g = g;
    g->device = c->device;
    g->st = m->st;  // resolution, origin, levels; per-rank counters stay per-rank (sum them with gsdf_hip_comm_allreduce_sum_u64)
    g->st.n_tris = total;
    if (total) {
      g->d_tris = pool_take(c->device, total, &g->cap);
      if (!g->d_tris) {
        if (hipMalloc((void**)&g->d_tris, total * 36) != hipSuccess) { (void)hipGetLastError(); return bail(fail(GSDF_ERR_HIP, "hipMalloc of the gathered triangle buffer failed")); }
        g->cap = total;
      }
    }
    p->st.bytes_received = (total - mine) * 36;
  }
  if (mode == GSDF_GATHER_ALL) p->st.bytes_sent = W > 1 ? mine * 36 : 0;  // (a broadcast: the ring / tree forwards it; one copy leaves this rank)
  else if (mode == GSDF_GATHER_ROOT) p->st.bytes_sent = c->rank == root ? 0 : mine * 36;
  if (hipEventCreate(&p->ev_payload0) != hipSuccess || hipEventCreate(&p->ev_payload1) != hipSuccess) return bail(fail(GSDF_ERR_HIP, "hipEventCreate failed"));
  if (hipEventRecord(p->ev_payload0, c->stream) != hipSuccess) return bail(fail(GSDF_ERR_HIP, "hipEventRecord failed"));
  // 2. payload: one grouped launch
  if (total && mode != GSDF_GATHER_NONE) {
    ncclResult_t r0 = R->GroupStart();
    if (r0 != ncclSuccess) return bail(fail(GSDF_ERR_HIP, std::string("ncclGroupStart: ") + R->GetErrorString(r0)));
    uint64_t off = 0;
    ncclResult_t rb = ncclSuccess;
    for (int r = 0; r < W && rb == ncclSuccess; r++) {
      const uint64_t n = p->counts[(size_t)r];
      if (n) {
        if (mode == GSDF_GATHER_ALL) {
          rb = R->Broadcast(r == c->rank ? (const void*)m->d_tris : (const void*)(p->g->d_tris + off * 9), p->g->d_tris + off * 9, (size_t)n * 9, ncclFloat32, r, c->comm, c->stream);
        } else if (c->rank == root) {  // ROOT, on the root: its own share by a device copy, everybody else's by a receive
          if (r == root) { if (hipMemcpyAsync(p->g->d_tris + off * 9, m->d_tris, (size_t)n * 36, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) rb = ncclUnhandledCudaError; }
          else rb = R->Recv(p->g->d_tris + off * 9, (size_t)n * 9, ncclFloat32, r, c->comm, c->stream);
        } else if (r == c->rank) {     // ROOT, elsewhere: send mine
          rb = R->Send(m->d_tris, (size_t)n * 9, ncclFloat32, root, c->comm, c->stream);
        }
      }
      off += n;
    }
    ncclResult_t r1 = R->GroupEnd();
    if (rb != ncclSuccess) return bail(fail(GSDF_ERR_HIP, std::string("gather payload: ") + R->GetErrorString(rb)));
    if (r1 != ncclSuccess) return bail(fail(GSDF_ERR_HIP, std::string("ncclGroupEnd: ") + R->GetErrorString(r1)));
  }
  if (hipEventRecord(p->ev_payload1, c->stream) != hipSuccess) return bail(fail(GSDF_ERR_HIP, "hipEventRecord failed"));
  *out = p;
  return GSDF_OK;
}

extern "C" int gsdf_hip_mesh_gatherv_wait(gsdf_gather* p, gsdf_mesh** out, uint64_t* counts, gsdf_gather_stats* st) {
  if (!p) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  if (out) *out = nullptr;
  (void)hipSetDevice(p->c->device);
  hipError_t e = hipEventSynchronize(p->ev_payload1);
  int rc = GSDF_OK;
  if (e != hipSuccess) rc = fail(GSDF_ERR_HIP, std::string("gatherv: ") + hipGetErrorString(e));
  float ms = 0;
  if (rc == GSDF_OK) (void)hipEventElapsedTime(&ms, p->ev_payload0, p->ev_payload1);
  p->st.ms_counts = p->ms_counts;
  p->st.ms_payload = ms;
  if (counts) for (size_t r = 0; r < p->counts.size(); r++) counts[r] = p->counts[r];
  if (st) *st = p->st;
  if (rc == GSDF_OK && out) { *out = p->g; p->g = nullptr; }
  if (p->g) gsdf_hip_mesh_destroy(p->g);
  (void)hipEventDestroy(p->ev_payload0);
  (void)hipEventDestroy(p->ev_payload1);
  delete p;
  return rc;
}

// All-gatherv in one call (gsdf_hip_mesh_gatherv_start + _wait, mode ALL): every rank gets every triangle.
extern "C" int gsdf_hip_mesh_gatherv(const gsdf_mesh* m, gsdf_comm* c, gsdf_mesh** out, uint64_t* counts) {
  if (!out) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  gsdf_gather* p = nullptr;
  if (int rc = gsdf_hip_mesh_gatherv_start(m, c, GSDF_GATHER_ALL, 0, &p)) return rc;
  return gsdf_hip_mesh_gatherv_wait(p, out, counts, nullptr);
}

extern "C" int gsdf_hip_mesh_stats_get(const gsdf_mesh* m, gsdf_mesh_stats* st) {
  if (!m || !st) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  *st = m->st;
  return GSDF_OK;
}
extern "C" int gsdf_hip_mesh_read(const gsdf_mesh* m, uint64_t first, uint64_t count, float* dst) {
  if (!m || (!dst && count)) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  if (first + count > m->st.n_tris) return fail(GSDF_ERR_BAD_ARGUMENT, "triangle range out of bounds");
  if (count == 0) return GSDF_OK;
  // The reference's pull loop (glrender.RenderAll: 4096 triangles per ReadTriangles call) would issue ~1700 small
  // device-to-host copies at resdiv 1600. Partial reads are served from the mesh's pinned host copy instead: one DMA
  // on the first call, plain memcpy afterwards.
  // A read of (nearly) everything takes the same route, with a multi-threaded copy out of the pinned buffer (a pageable
  // hipMemcpy is staged by the runtime at ~12 GB/s).
  {
    const float* h = nullptr;
    if (gsdf_hip_mesh_host_tris(const_cast<gsdf_mesh*>(m), &h) == GSDF_OK) {
      big_memcpy(dst, h + first * 9, count * 36);  // single memcpy below 32 MB
      return GSDF_OK;
    }
  }
  HIP_TRY(hipSetDevice(m->device));  // no pinned memory to be had: plain copy from the device
  HIP_TRY(hipMemcpy(dst, m->d_tris + first * 9, count * 36, hipMemcpyDeviceToHost));
  return GSDF_OK;
}
extern "C" const float* gsdf_hip_mesh_dev_tris(const gsdf_mesh* m) { return m ? m->d_tris : nullptr; }

extern "C" int gsdf_hip_mesh_stl(const gsdf_mesh* m, uint8_t* dst, size_t dst_cap) {
  if (!m || !dst) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  const uint64_t n = m->st.n_tris;
  if (n == 0) return fail(GSDF_ERR_EMPTY_BUFFERS, "empty triangle slice");
  if (n > 0xffffffffull) return fail(GSDF_ERR_BAD_ARGUMENT, "amount of triangles in model exceeds STL design limits");
  const size_t bytes = 84 + 50 * (size_t)n;
  if (dst_cap < bytes) return fail(GSDF_ERR_SHORT_BUFFER, "short buffer");
  // built on device and moved into the mesh's pinned host buffer (gsdf_hip_mesh_host_stl), then copied out
  const uint8_t* h = nullptr;
  size_t len = 0;
  const int rc = gsdf_hip_mesh_host_stl(const_cast<gsdf_mesh*>(m), &h, &len);
  if (rc) return rc;
  big_memcpy(dst, h, len);
  return GSDF_OK;
}

// Zero-copy result views. The reference's consumers take the mesh through ReadTriangles into pageable memory; at
// 6.8 M triangles that is 23 ms of page faults and staged copies for a 1.7 ms mesh (and 99 ms for the STL through the
// copying call). Here the whole result is moved once, by DMA, into pinned host memory that the mesh owns and the caller
// reads in place ([]ms3.Triangle / []byte over the pointer; valid until gsdf_hip_mesh_destroy).
static int host_buf(void** buf, size_t* cap, size_t need) {
  if (*buf && *cap >= need) return GSDF_OK;
  hpool_give(*buf, *cap);
  *buf = hpool_take(need, cap);
  if (*buf) return GSDF_OK;
  *cap = 0;
  const size_t want = need + need / 16 + 4096;
  hipError_t e = hipHostMalloc(buf, want, hipHostMallocPortable | hipHostMallocMapped);  // the pool is shared by all devices of the process; mapped: may serve as a kernel's output buffer
  if (e != hipSuccess) { *buf = nullptr; (void)hipGetLastError(); return fail(GSDF_ERR_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e)); }
  *cap = want;
  return GSDF_OK;
}

extern "C" int gsdf_hip_mesh_host_tris(gsdf_mesh* m, const float** tris) {
  if (!m || !tris) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  *tris = nullptr;
  const uint64_t n = m->st.n_tris;
  if (n == 0) return GSDF_OK;
  if (m->host_out) {  // the mesher wrote them there
    *tris = m->d_tris;
    return GSDF_OK;
  }
  HIP_TRY(hipSetDevice(m->device));
  if (!m->h_tris) {
    const int rc = host_buf(&m->h_tris, &m->h_tris_cap, (size_t)n * 36);
    if (rc) return rc;
    hipStream_t rs = mesh_stream(m);
    hipError_t e = hipMemcpyAsync(m->h_tris, m->d_tris, (size_t)n * 36, hipMemcpyDeviceToHost, rs);
    if (e == hipSuccess) e = hipStreamSynchronize(rs);
    if (e != hipSuccess) { hpool_give(m->h_tris, m->h_tris_cap); m->h_tris = nullptr; m->h_tris_cap = 0; return fail(GSDF_ERR_HIP, std::string("D2H triangles: ") + hipGetErrorString(e)); }
  }
  *tris = (const float*)m->h_tris;
  return GSDF_OK;
}

extern "C" int gsdf_hip_mesh_host_stl(gsdf_mesh* m, const uint8_t** stl, size_t* len) {
  if (!m || !stl || !len) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  *stl = nullptr; *len = 0;
  const uint64_t n = m->st.n_tris;
  if (n == 0) return fail(GSDF_ERR_EMPTY_BUFFERS, "empty triangle slice");
  if (n > 0xffffffffull) return fail(GSDF_ERR_BAD_ARGUMENT, "amount of triangles in model exceeds STL design limits");
  const size_t bytes = 84 + 50 * (size_t)n;
  HIP_TRY(hipSetDevice(m->device));
  if (!m->h_stl) {
    int rc = host_buf(&m->h_stl, &m->h_stl_cap, bytes);
    if (rc) return rc;
    // device scratch for the records from the triangle-buffer pool (sized in 36-byte units)
    const uint64_t units = (bytes + 4 + 35) / 36;
    uint64_t dcap = 0;
    float* d_out = pool_take(m->device, units, &dcap);
    if (!d_out) {
      if (hipMalloc((void**)&d_out, units * 36) != hipSuccess) { (void)hipGetLastError(); return fail(GSDF_ERR_HIP, "hipMalloc of the STL scratch failed"); }
      dcap = units;
    }
    uint8_t* hdr = (uint8_t*)m->h_stl;  // pinned: a valid source for the async header upload
    std::memset(hdr, 0, 84);
    const uint32_t cnt = (uint32_t)n;
    std::memcpy(hdr + 80, &cnt, 4);
    hipStream_t rs = mesh_stream(m);
    hipError_t e = hipMemcpyAsync(d_out, hdr, 84, hipMemcpyHostToDevice, rs);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(stl_kernel, dim3(grid_for(n, 256, 8)), dim3(BLOCK), 0, rs, m->d_tris, n, (uint8_t*)d_out);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(m->h_stl, d_out, bytes, hipMemcpyDeviceToHost, rs);
    if (e == hipSuccess) e = hipStreamSynchronize(rs);
    pool_give(m->device, d_out, dcap);
    if (e != hipSuccess) { hpool_give(m->h_stl, m->h_stl_cap); m->h_stl = nullptr; m->h_stl_cap = 0; return fail(GSDF_ERR_HIP, std::string("STL build/transfer: ") + hipGetErrorString(e)); }
  }
  *stl = (const uint8_t*)m->h_stl;
  *len = bytes;
  return GSDF_OK;
}

extern "C" void gsdf_hip_mesh_destroy(gsdf_mesh* m) {
  if (!m) return;
  release_tris(m);
  hpool_give(m->h_tris, m->h_tris_cap);
  hpool_give(m->h_stl, m->h_stl_cap);
  if (m->rstream) (void)hipStreamDestroy(m->rstream);
  delete m;
}

// ---- gleval.BlockCachedSDF3 (gleval/gleval.go:110-218) over a HIP program ---------------------------------------
// Host-side wrapper, as in the reference: a lossy cache keyed by the lattice cell of the position
// (int(mul * (p - bb.Min)) per axis, mul = 1/res); misses are evaluated in ONE batch by the wrapped evaluator.
struct gsdf_blockcache {
  gsdf_program* sdf = nullptr;
  float mul[3] = {0, 0, 0};
  struct Key { long long x, y, z; bool operator==(const Key& o) const { return x == o.x && y == o.y && z == o.z; } };
  struct Hash {
    size_t operator()(const Key& k) const {
      unsigned long long h = (unsigned long long)k.x * 0x9E3779B97F4A7C15ull;
      h ^= (unsigned long long)k.y + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
      h ^= (unsigned long long)k.z + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
      return (size_t)h;
    }
  };
  std::unordered_map<Key, float, Hash> m;
  std::vector<float> posbuf, distbuf;
  std::vector<size_t> idxbuf;
  uint64_t hits = 0, evals = 0;
};

extern "C" int gsdf_hip_blockcache_reset(gsdf_blockcache* c, gsdf_program* sdf, float resx, float resy, float resz) {
  if (!c || !sdf) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  if (resx <= 0 || resy <= 0 || resz <= 0 || std::isnan(resx) || std::isnan(resy) || std::isnan(resz))
    return fail(GSDF_ERR_RESOLUTION, "invalid resolution for BlockCachedSDF3");  // gleval.go:127-129
  if (sdf->prog.is2d) return fail(GSDF_ERR_DIMENSION, "program is 2D");
  c->m.clear();
  c->sdf = sdf;
  c->mul[0] = 1.0f / resx; c->mul[1] = 1.0f / resy; c->mul[2] = 1.0f / resz;  // DivElem({1,1,1}, res)
  c->posbuf.clear(); c->distbuf.clear(); c->idxbuf.clear();
  c->hits = 0; c->evals = 0;  // Reset also resets the statistics (gleval.go:124)
  return GSDF_OK;
}
extern "C" int gsdf_hip_blockcache_create(gsdf_program* sdf, float resx, float resy, float resz, gsdf_blockcache** out) {
  if (!out) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  *out = nullptr;
  gsdf_blockcache* c = new (std::nothrow) gsdf_blockcache();
  if (!c) return fail(GSDF_ERR_BAD_ARGUMENT, "out of memory");
  const int rc = gsdf_hip_blockcache_reset(c, sdf, resx, resy, resz);
  if (rc) { delete c; return rc; }
  *out = c;
  return GSDF_OK;
}
extern "C" void gsdf_hip_blockcache_destroy(gsdf_blockcache* c) { delete c; }
extern "C" uint64_t gsdf_hip_blockcache_hits(const gsdf_blockcache* c) { return c ? c->hits : 0; }
extern "C" uint64_t gsdf_hip_blockcache_evaluations(const gsdf_blockcache* c) { return c ? c->evals : 0; }

// (*BlockCachedSDF3).Evaluate (gleval.go:154-211). pos: n x 3 float32 with the given byte stride (12 or 16).
extern "C" int gsdf_hip_blockcache_eval3(gsdf_blockcache* c, const void* pos, size_t stride, size_t n_pos, float* dist, size_t n_dist) {
  if (!c || !c->sdf) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  if (n_pos != n_dist) return fail(GSDF_ERR_LENGTH_MISMATCH, "position and distance buffer length mismatch");
  if (n_pos == 0) return fail(GSDF_ERR_EMPTY_BUFFERS, "empty buffers");
  if (!pos || !dist) return fail(GSDF_ERR_BAD_ARGUMENT, "null buffer");
  if (stride % 4 != 0 || stride < 12) return fail(GSDF_ERR_BAD_ARGUMENT, "bad position stride");
  const float* bbmin = c->sdf->prog.bb;
  auto key_of = [&](const float* p) {
    // tp = MulElem(mul, Sub(p, bb.Min)); int(tp.X) truncates toward zero (Go float->int conversion)
    gsdf_blockcache::Key k;
    k.x = (long long)(c->mul[0] * (p[0] - bbmin[0]));
    k.y = (long long)(c->mul[1] * (p[1] - bbmin[1]));
    k.z = (long long)(c->mul[2] * (p[2] - bbmin[2]));
    return k;
  };
  c->posbuf.clear();
  c->idxbuf.clear();
  const char* base = (const char*)pos;
  for (size_t i = 0; i < n_pos; i++) {
    const float* p = (const float*)(base + i * stride);
    auto it = c->m.find(key_of(p));
    if (it != c->m.end()) {
      dist[i] = it->second;
    } else {
      c->posbuf.insert(c->posbuf.end(), p, p + 3);
      c->idxbuf.push_back(i);
    }
  }
  const size_t nseek = c->idxbuf.size();
  if (nseek > 0) {
    c->distbuf.resize(nseek);
    const int rc = gsdf_hip_eval3(c->sdf, c->posbuf.data(), 12, nseek, c->distbuf.data(), nseek);
    if (rc) return rc;
    for (size_t i = 0; i < nseek; i++) c->m[key_of(&c->posbuf[3 * i])] = c->distbuf[i];  // later entries of a cell overwrite
    for (size_t i = 0; i < nseek; i++) dist[c->idxbuf[i]] = c->distbuf[i];
  }
  c->evals += n_pos;
  c->hits += n_pos - nseek;
  return GSDF_OK;
}
