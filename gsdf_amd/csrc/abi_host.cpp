// abi_host.cpp -- host-side plumbing shared by the C ABI's translation units (abi_host.h): the per-thread error text, the
// triangle-buffer and pinned-buffer pools, the multi-threaded result copy.
#include "abi_host.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

static thread_local std::string g_err;
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
extern "C" const char* gsdf_hip_last_error(void) { return g_err.c_str(); }

// Triangle buffers are recycled through a small per-process pool: hipMalloc/hipFree of the multi-GB
// output buffer would otherwise dominate a mesh call.
namespace {
struct TriBuf { int device; float* p; uint64_t cap; };
std::mutex g_pool_mu;
std::vector<TriBuf> g_pool;
struct HostBuf { void* p; size_t cap; };
std::vector<HostBuf> g_hpool;
}  // namespace
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
// How many idle device buffers the pool keeps. Three meshes in flight per handle, a records payload each, a gather with its receive
// buffer and the gathered mesh of the step before: eight buffers circulate on a rank of an N > 1 run (bench.py), and with room for
// four every step freed one and allocated another -- hipFree waits for the device -- and, over the ipc test transport, a peer's
// cached mapping of a freed buffer outlived it (a recycled address faulted: round 6, HISTORY item 21). GSDF_HIP_POOL_MAX overrides.
static size_t pool_max() {
  static const size_t v = [] { const char* e = getenv("GSDF_HIP_POOL_MAX"); const long n = e ? atol(e) : 16; return (size_t)(n < 1 ? 1 : n); }();
  return v;
}
// Buffers whose hipIpcMemHandle has left the process (the ipc test transport, abi_comm.cpp): a peer keeps its mapping of such a buffer
// cached, so the pool never frees one while a transport of that kind lives -- a recycled address would be read through the stale mapping.
namespace {
std::vector<void*> g_exported;
int g_exporters = 0;
bool exported_locked(const void* p) { return std::find(g_exported.begin(), g_exported.end(), p) != g_exported.end(); }
}  // namespace
void pool_exporter_opened() { std::lock_guard<std::mutex> lk(g_pool_mu); g_exporters++; }
void pool_exporter_closed() { std::lock_guard<std::mutex> lk(g_pool_mu); if (--g_exporters <= 0) { g_exporters = 0; g_exported.clear(); } }
void pool_note_exported(void* base) { std::lock_guard<std::mutex> lk(g_pool_mu); if (!exported_locked(base)) g_exported.push_back(base); }
void pool_give(int device, float* p, uint64_t cap) {
  if (!p) return;
  std::lock_guard<std::mutex> lk(g_pool_mu);
  if (g_pool.size() >= pool_max()) {  // drop the smallest (of those no peer process may have mapped)
    int sm = -1;
    for (size_t i = 0; i < g_pool.size(); i++)
      if (!exported_locked(g_pool[i].p) && (sm < 0 || g_pool[i].cap < g_pool[(size_t)sm].cap)) sm = (int)i;
    const bool p_exported = exported_locked(p);
    if (sm >= 0 && (p_exported || g_pool[(size_t)sm].cap < cap)) { (void)hipFree(g_pool[(size_t)sm].p); g_pool[(size_t)sm] = TriBuf{device, p, cap}; }
    else if (!p_exported) (void)hipFree(p);
    else g_pool.push_back(TriBuf{device, p, cap});  // (everything idle is exported: the pool grows past its limit rather than free one)
    return;
  }
  g_pool.push_back(TriBuf{device, p, cap});
}

// memcpy out of pinned memory into a caller's (usually freshly allocated, not yet faulted-in) buffer: one thread moves
// ~10 GB/s and takes every page fault itself; large results are split over a few threads.
void big_memcpy(void* dst, const void* src, size_t n) {
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

// Stream for work issued on a finished mesh (result copies, stl_kernel): created on first use and destroyed with the mesh,
// so that reading a mesh after gsdf_hip_program_destroy (finalisers / garbage collectors run in any order) never touches
// the program's destroyed stream. nullptr (the null stream) if a stream cannot be had.
hipStream_t mesh_stream(gsdf_mesh* m) {
  if (!m->rstream && hipStreamCreateWithFlags(&m->rstream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); m->rstream = nullptr; }
  return m->rstream;
}
void release_tris(gsdf_mesh* m) {
  if (m->host_out) hpool_give(m->d_tris, (size_t)m->cap * 36);
  else pool_give(m->device, m->d_tris, m->cap);
  m->d_tris = nullptr; m->cap = 0; m->host_out = false;
}

// Pinned host buffers for the zero-copy result views (gsdf_hip_mesh_host_tris / _host_stl), from the pool.
int host_buf(void** buf, size_t* cap, size_t need) {
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
