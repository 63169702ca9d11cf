// abi_program.h -- the program handle and the launch helpers of the translation units that own kernels (abi_eval.hip,
// abi_mesh.hip). Include after the kernel headers: the handle's LDS arithmetic uses their constants (BLOCK, TRI_STAGE, ...).
#pragma once
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "abi_host.h"
#include "compile.h"
#include "specialize.h"

struct gsdf_program {
  gsdf_dev::Program prog;
  int device = 0;
  uint32_t* d_code = nullptr;
  hipStream_t stream = nullptr;
  hipStream_t stream_b = nullptr;  // the octree mesher's second chain in flight (created on first use)
  hipStream_t stream_c = nullptr;  // its third
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
  struct Slot { void* h_pos = nullptr; float* h_dist = nullptr; hipStream_t s = nullptr; bool busy = false; bool waiting = false; unsigned gen = 0; float* user_dist = nullptr; size_t n = 0; bool zero_copy = false; unsigned* h_flag = nullptr; unsigned* d_flag = nullptr; unsigned flag_val = 0; bool flagged = false; };  // gen: bumped at every acquisition; a ticket is slot | gen << 8, so a stale or repeated wait is refused instead of releasing somebody else's call
  // h_flag / d_flag: a word of pinned, device-mapped memory the stream writes flag_val into behind the kernel (completion by
  // polling: a blocking call of 4 096 points is 26 us through hipStreamSynchronize, of which the kernel is a few)
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
  } q0, q1, ctr, spec_pass, dc_tile, dc_grid, dc_dist, dc_fv, dc_nrm, dc_edge, dc_erun, dc_flag, flat_grid, flat_bits, flat_list, rec, hdr, grp, b_q0, b_q1, b_ctr, b_spec_pass, b_rec, b_hdr, b_grp, c_q0, c_q1, c_ctr, c_spec_pass, c_rec, c_hdr, c_grp;  // b_* / c_*: the octree mesher's second and third workspace (its further chains in flight run on stream_b / stream_c, beside the first: gsdf_hip_mesh_octree_start)  // flat_bits: sign and near-surface bit planes of the flat renderer's lattice (flat_grid_kernel -> flat_cut_scan_kernel), flat_list: the cut cubes (-> flat_march_list_kernel)  // rec / hdr: cut-leaf records and block headers of the two-kernel leaf phase (the group sums follow the counters in ctr)  // dc_*: dual contouring workspace (index grid: 4 B per lattice cell); flat_grid: FlatRenderer distances
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_b[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // second set: the octree mesher has up to three chains in flight (gsdf_hip_mesh_octree_start)
  hipEvent_t ev_c[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // third set
  // Three chains: measured at npt-flange resdiv 1600 with one, two, three, four and six meshes in flight (tools/gpu_pipe_depth.py,
  // round 6): 0.452, 0.365-0.369, 0.340, 0.374, 0.358 ms per mesh -- a third chain fills what two leave idle around the evaluating
  // kernel (the five dependent launches of the centre tests, the marching kernel's latency-bound passes), a fourth only adds contention.
  static constexpr int kJobs = 3;
  bool job_busy[kJobs] = {false, false, false};
  hipStream_t job_stream[kJobs] = {nullptr, nullptr, nullptr};
  bool mesh_in_flight() const { return job_busy[0] || job_busy[1] || job_busy[2]; }
  void* h_ctr = nullptr;  // pinned host copy of the device counters (a pageable destination makes the D2H copy a staged, blocking one)
  uint64_t last_tris = 0;  // triangle count of the previous mesh on this handle: sizes the next output buffer
  uint64_t last_recs = 0;  // cut leaves of the previous mesh with payload = records: sizes the next payload buffer
  uint64_t rec_blocks = 0;  // 64-leaf blocks the cut-leaf record arena is sized for (0: first mesh, start with kRecBlocks0)
  uint64_t last_dc_cubes = 0;  // kept cubes of the previous dual contouring pass: sizes the next queues
  // run-time specialised kernels for this program (gsdf_hip_program_specialize): hiprtc module, else interpreter
  hipModule_t spec_mod = nullptr, spec_mod2 = nullptr, spec_mod3 = nullptr, spec_mod4 = nullptr;  // spec_mod4: leaf kernel rebuilt with a larger register budget; spec_mod2: second group, built on first use (see spec_aux); spec_mod3: eval kernel rebuilt for 3 workgroups per CU
  hipFunction_t f_eval_k1 = nullptr;  // eval_kernel<D, 1, 4>: one point per lane, for host-mapped calls (abi_eval.hip: eval_dev, spec_eval_k1)
  hipModule_t spec_mod_k1 = nullptr;
  bool spec_k1_tried = false;
  hipFunction_t f_eval = nullptr, f_prune = nullptr, f_prune_spec = nullptr, f_leaf = nullptr;
  hipFunction_t f_dc_block_test = nullptr;  // the interval test of dual contouring's origin sweep, block by block (kernels_dc.h: dc_block_test_kernel)
  hipFunction_t f_dc_origin = nullptr, f_dc_edges = nullptr, f_dc_normals = nullptr, f_normals = nullptr, f_image = nullptr, f_flat_grid = nullptr;
  bool spec_aux_tried = false;
  int spec_eval_k = 0, spec_eval_w = 0, spec_leaf_k = 0, spec_leaf_w = 0;
  bool spec_leaf_both = false;  // the specialised leaf kernel has both column passes in one body (kernels_octree.h: BOTH)
  // distinct z rows (gsdf_mesh_opts.share_corners = 2; kernels_octree.h: DZ): built the first time a mesh asks for it (spec_leaf_dz)
  hipModule_t spec_mod_dz = nullptr;
  hipFunction_t f_leaf_dz = nullptr;
  bool spec_dz_tried = false, spec_leaf_dz_both = false;
  int spec_leaf_dz_w = 0;
  // distinct lattice points (share_corners = 1; kernels_octree.h: leaf_dense_kernel): likewise built on first use (spec_leaf_dense)
  hipModule_t spec_mod_dense = nullptr;
  hipFunction_t f_leaf_dense = nullptr;
  bool spec_dense_tried = false;
  int spec_leaf_dense_w = 0;
  size_t lds_dense() const { return (size_t)(prog.nslots * 4 > 8 ? prog.nslots * 4 : 8) * BLOCK * sizeof(float) + 256 + 4 * 512 * sizeof(float) + 4 * 24 * sizeof(float) + 16; }
  double spec_compile_s = 0;
  // Specialisation in the background (gsdf_hip_program_specialize_async): a thread builds and loads the kernels on a shadow handle
  // (this program, this device, nothing else) while the interpreter kernels serve; the entry points adopt the result at their
  // next call (abi_eval.hip: spec_adopt). 0 idle | 1 building | 2 built, to be adopted | 3 failed (spec_async_err)
  std::atomic<int> spec_async{0};
  std::thread spec_thread;
  gsdf_program* spec_shadow = nullptr;
  std::mutex spec_async_mu;
  std::string spec_async_err;
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
inline bool fused_leaf() {
  static const bool f = [] { const char* e = getenv("GSDF_HIP_FUSED_LEAF"); return e && atoi(e) != 0; }();
  return f;
}

inline void gsdf_program::leaf_config(int* k, int* w, size_t* lds) const {
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

inline unsigned grid_for(uint64_t n, int num_cu, int blocks_per_cu) {
  uint64_t b = (n + BLOCK - 1) / BLOCK;
  uint64_t mx = (uint64_t)num_cu * (uint64_t)blocks_per_cu;
  if (b > mx) b = mx;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// Second group of a specialised handle (dual contouring, normals, flat lattice pass, image renderer), built on first use:
// abi_eval.hip.
void spec_aux(gsdf_program* p);
// A background build (gsdf_hip_program_specialize_async) that has finished is taken over here: called at the top of the entry
// points that launch evaluating kernels. One relaxed load when there is nothing to adopt.
void spec_adopt_slow(gsdf_program* p);
inline void spec_adopt(gsdf_program* p) { if (p->spec_async.load(std::memory_order_acquire) == 2) spec_adopt_slow(p); }
// The evaluating kernel with distinct z rows for a specialised handle, built on first use: abi_eval.hip.
void spec_leaf_dz(gsdf_program* p);
void spec_leaf_dense(gsdf_program* p);

namespace {
// ms3.Box.ScaleCentered(1.01) = NewCenteredBox(Center(), MulElem(scale, Size())) [external]; float32, unfused.
inline void scale_centered(const float bb[6], float s, float mn[3], float mx[3]) {
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
