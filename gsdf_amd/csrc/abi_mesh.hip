// abi_mesh.hip -- C ABI (include/gsdf_hip.h), mesher side: glrender.Octree + marchCubes, FlatRenderer and
// DualContourRenderer on device, and the accessors of the resulting mesh (ReadTriangles drain, STL, pinned host views).
// Kernels: kernels_octree.h, kernels_flat.h, kernels_dc.h, kernels_stl.h.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "kernels_common.h"
#include "kernels_octree.h"
#include "kernels_flat.h"
#include "kernels_dc.h"
#include "kernels_stl.h"
#include "kernels_minecraft.h"
#include "abi_program.h"
#include "host_math.h"

// ---- glrender.Octree on device ----------------------------------------------------------------------------------------------
// One mesh = one chain of launches with no host round trip (each kernel reads its input count from the previous one's counter in
// device memory), then one wait. The chain and the wait are two entry points -- gsdf_hip_mesh_octree_start / _wait -- so that a
// caller with several meshes to make enqueues the next chain before it waits for the previous one: the ~30 us between the last
// kernel of a mesh and the first kernel of the next (completion wake-up, the caller's own bookkeeping, launch latency) then pass
// under a running chain instead of an idle GPU. Up to two chains per handle; they run back to back on the handle's stream, so the
// workspace arenas are reused in stream order and only the pinned counter block and the timing events exist twice.
namespace {
constexpr size_t kCtrBytes = (sizeof(MeshCounters) + 255) & ~(size_t)255;  // the group sums follow the counters: one memset clears both
const bool ctr_from_kernel = [] { const char* e = getenv("GSDF_HIP_CTR_FROM_KERNEL"); return !e || atoi(e) != 0; }();  // developer knobs (A/B timing)
}  // namespace

#define HIP_TRYM(expr)                                                                                   \
  do {                                                                                                   \
    hipError_t _e = (expr);                                                                              \
    if (_e != hipSuccess) return fail(GSDF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

struct gsdf_mesh_job {
  gsdf_program* p = nullptr;
  gsdf_mesh* m = nullptr;
  gsdf_mesh_opts opts{};
  float res = 0, ox = 0, oy = 0, oz = 0;
  hipStream_t s = nullptr;
  int levels = 0, lq = 0, ls = 0, prune_cols = 0, pmask = 0, ptest = 0, lk = 0, lw = 0;
  size_t lds_prune = 0, lds_m = 0;
  bool want_recs = false;
  uint64_t qcap = 0, want = 0, want_rec_n = 0;  // capacities of the next attempt
  // the attempt in flight
  int attempt = 0, slot;
  bool used_brick = false, two_kernel = false, ctr_on_host = false, used_dz = false, used_dense = false;
  int chain_first = 0;  // levels <= chain_first were tested by prune_kernel (one launch per level), the ones above speculatively
  uint64_t nblk = 0;
  size_t clear_bytes = 0;
  MeshCounters* d_ctr = nullptr;
  MeshCounters* hcp = nullptr;  // this job's pinned counter block
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr, evr = nullptr, ev_done = nullptr;
  // this job's workspace: the handle has two, so that two chains in flight do not share one (they run on two streams, the
  // latency-bound top and tail of one under the leaf kernel of the other)
  struct Ws { gsdf_program::Arena &q0, &q1, &ctr, &spec_pass, &rec, &hdr, &grp; };
  Ws w;
  explicit gsdf_mesh_job(gsdf_program* pp, int sl)
      : p(pp), slot(sl), w(sl == 0 ? Ws{pp->q0, pp->q1, pp->ctr, pp->spec_pass, pp->rec, pp->hdr, pp->grp}
                           : sl == 1 ? Ws{pp->b_q0, pp->b_q1, pp->b_ctr, pp->b_spec_pass, pp->b_rec, pp->b_hdr, pp->b_grp}
                                     : Ws{pp->c_q0, pp->c_q1, pp->c_ctr, pp->c_spec_pass, pp->c_rec, pp->c_hdr, pp->c_grp}) {}

  int enqueue();            // one attempt: the whole chain onto the stream, nothing waited for
  int finish(bool* again);  // waits for it; *again: a capacity was short -- enqueue() once more (nothing was dropped silently)
  int stats();              // the mesh's statistics from the counters
};

int gsdf_mesh_job::enqueue() {
    ctr_on_host = false;
    used_dz = false;
    used_dense = false;
    MeshCounters& hc = *hcp;
    (void)hc;
    HIP_TRYM(w.q0.ensure(qcap * sizeof(Cube)));
    HIP_TRYM(w.q1.ensure(qcap * sizeof(Cube)));
    const uint64_t cap0 = w.q0.cap / sizeof(Cube), cap1 = w.q1.cap / sizeof(Cube);
    gsdf_program::Arena* q[2] = {&w.q0, &w.q1};
    const uint64_t capq[2] = {cap0, cap1};
    // records payload: room for the packed records instead (sized like the triangles: previous mesh, else a guess + one exact rerun)
    if (want_recs && !m->d_recs) {
      const uint64_t nrec = want_rec_n ? want_rec_n : (p->last_recs ? p->last_recs + p->last_recs / 16 + 1024 : (uint64_t)1 << 20);
      const uint64_t units = (dense_bytes(nrec) + 35) / 36;
      m->d_recs = (uint8_t*)pool_take(p->device, units, &m->recs_cap36);
      if (!m->d_recs) { HIP_TRYM(hipMalloc((void**)&m->d_recs, units * 36)); m->recs_cap36 = units; }
    }
    // triangle buffer: caller's size, else a pooled buffer, else a guess that is corrected by one exact rerun
    if (!want_recs && !m->d_tris) {
      uint64_t need = want ? want : (p->last_tris ? p->last_tris + p->last_tris / 16 + 1024 : (uint64_t)1 << 20);
      if (opts.host_output) {
        // triangles straight into pinned host memory: the stage flushes of leaf_kernel are 4.6 KB coalesced bursts,
        // which PCIe takes well; the transfer then overlaps the kernel instead of following it
        void* hb = nullptr;
        size_t hcap = 0;
        if (int rc = host_buf(&hb, &hcap, (size_t)need * 36)) return rc;
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
    nblk = p->rec_blocks ? p->rec_blocks : kRecBlocks0;
    if (nblk > nblk_q) nblk = nblk_q;
    const uint64_t ngrp = (nblk + MARCH_GROUP - 1) / MARCH_GROUP;
    bool want_two = !fused_leaf();
    if (want_two && (w.hdr.ensure(nblk * sizeof(uint32_t)) != hipSuccess || w.rec.ensure(nblk * (size_t)REC_BLOCK * sizeof(uint32_t)) != hipSuccess)) {
      (void)hipGetLastError();  // no room for the records: the fused kernel needs none
      w.hdr.release(); w.rec.release();
      want_two = false;
    }
    if (want_recs && (!want_two || w.grp.ensure(ngrp * sizeof(unsigned long long)) != hipSuccess)) {
      (void)hipGetLastError();
      return (fail(GSDF_ERR_CAPACITY, "no device memory for the cut-leaf record arena (payload = records needs it)"));
    }
    clear_bytes = kCtrBytes + (want_two ? ngrp * sizeof(unsigned long long) : 0);
    HIP_TRYM(w.ctr.ensure(clear_bytes));
    d_ctr = (MeshCounters*)w.ctr.p;
    // the counters and the group sums are cleared by the chain's first kernel (prune_spec_kernel); one launch per level: a memset
    static const bool use_spec = [] { const char* e = getenv("GSDF_HIP_PRUNE_SPEC"); return !e || atoi(e) != 0; }();
    if (!use_spec) HIP_TRYM(hipMemsetAsync(d_ctr, 0, clear_bytes, s));
    HIP_TRYM(hipEventRecord(ev0, s));
    static const int prune_bpc = [] { const char* e = getenv("GSDF_HIP_PRUNE_BPC"); return e ? atoi(e) : 2; }();  // tuning knob: 2 workgroups per CU (round 4: 4 -> 2 is -3 us on a blocking mesh and leaves the CUs to the other mesh in flight: 0.503 -> 0.486 ms per mesh, two in flight)
    // The first S levels (at most 7: 299,593 cubes) are centre-tested speculatively, every cube of the complete octree at once,
    // and resolved by a second launch (kernels.h: prune_spec_kernel / prune_resolve_kernel): two launches instead of a chain of
    // S dependent ones. GSDF_HIP_PRUNE_SPEC=0 keeps one launch per level (cross-check in the tests).
    int first_level = levels;  // first level of the per-level chain
    chain_first = levels;
    // Brick masks (dev_ops.h: D_SKIP): valid when the last level of cubes, which the leaf kernels continue from, is a level-3 brick
    // that was centre-tested by interval evaluation -- that test leaves, in every surviving cube's w field, which operand
    // subtrees of the program cannot matter anywhere in the cube. GSDF_HIP_NO_BRICK_MASKS=1: no numbers in the program at all.
    const bool masks_valid = lq == 3 && ptest == 1 && p->prog.n_skip_ids > 0 && (pmask == 1 || (pmask > 1 && ((pmask >> lq) & 1)));
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
      // brick masks of the block's last level, when that is the level the leaf kernels continue from (meshes of up to nine levels):
      // two bytes per cube of that level, behind the statistics rows
      const unsigned n_last = 1u << (3 * (S - 1));
      const size_t mask_off_b = (part_off + (size_t)rrows * 16 * sizeof(unsigned) + 63) & ~(size_t)63;
      const bool spec_masks = masks_valid && last_spec == lq;
      HIP_TRYM(w.spec_pass.ensure(mask_off_b + (spec_masks ? (size_t)n_last * sizeof(uint16_t) : 0)));
      uint16_t* d_mask16 = spec_masks ? (uint16_t*)((char*)w.spec_pass.p + mask_off_b) : nullptr;
      const bool chain_follows = last_spec - 1 >= lq;
      spec_part = chain_follows ? (unsigned*)((char*)w.spec_pass.p + part_off) : nullptr;
      spec_rows = rrows;
      spec_top_S = levels | (S << 8);
      spec_mask = test_mask_of(pmask);
      const unsigned test_mask = test_mask_of(pmask);
      const int shard_level = opts.shard_count > 1 ? ls : -1;
      const unsigned sgrid = grid_for(n_spec, p->num_cu, 8);
      if (p->f_prune_spec) {
        HIP_TRYM(launch_fn(p->f_prune_spec, sgrid, BLOCK, lds_prune, s, (const uint32_t*)p->d_code, (int)levels, (unsigned)n_spec, (int)prune_cols,
                           (int)p->prog.nslots, ox, oy, oz, res, (unsigned)test_mask, (int)ptest, (int)shard_level, (unsigned)opts.shard_rank,
                           (unsigned)opts.shard_count, (uint8_t*)w.spec_pass.p, (unsigned*)d_ctr, (unsigned)(clear_bytes / 4), d_mask16,
                           (unsigned)(n_spec - n_last)));
      } else {
        hipLaunchKernelGGL(prune_spec_kernel, dim3(sgrid), dim3(BLOCK), lds_prune, s, p->d_code, levels, n_spec, prune_cols, p->prog.nslots, ox, oy,
                           oz, res, test_mask, ptest, shard_level, (unsigned)opts.shard_rank, (unsigned)opts.shard_count, (uint8_t*)w.spec_pass.p,
                           (unsigned*)d_ctr, (unsigned)(clear_bytes / 4), d_mask16, n_spec - n_last);
      }
      HIP_TRYM(hipGetLastError());
      hipLaunchKernelGGL(prune_resolve_kernel, dim3((n_spec + SPEC_STAGE - 1) / SPEC_STAGE), dim3(BLOCK), 0, s, (const uint8_t*)w.spec_pass.p, levels, S,
                         n_spec, test_mask, (Cube*)q[last_spec & 1]->p, (unsigned long long)capq[last_spec & 1], d_ctr, spec_part,
                         (const uint16_t*)d_mask16);
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
      static const int leaf_bpc = [] { const char* e = getenv("GSDF_HIP_LEAF_BPC"); return e ? atoi(e) : 32; }();  // (round 4: 32 -- the same for one mesh at a time, 0.503 -> 0.475 ms per mesh with two in flight: the other chain's small kernels are dispatched sooner behind a shorter queue of workgroups; the sentence that follows describes the round-3 sweep)  // grid = up to 64 workgroups per CU (4 resident): a few grid-stride iterations each, so the CUs drain evenly at the end (8 per CU: +8 % kernel time; one iteration per workgroup: +10 %)
#define LAUNCH_LEAF(KK, WW)                                                                                           \
  hipLaunchKernelGGL((leaf_kernel<KK, WW>), dim3(grid_for(bound, p->num_cu, leaf_bpc)), dim3(BLOCK), lds_m, s, p->d_code,      \
                     (const Cube*)q[lq & 1]->p, (unsigned long long)capq[lq & 1], lq, p->prog.nslots, ox, oy, oz, res,  \
                     m->d_tris, tcap, d_ctr)
      if (want_two) {
        // two kernels: evaluation + cut-leaf records, then marching cubes over the records
        uint32_t* d_hdr = (uint32_t*)w.hdr.p;
        uint32_t* d_rec = (uint32_t*)w.rec.p;
        unsigned long long* d_psum = (unsigned long long*)((char*)w.ctr.p + kCtrBytes);  // cleared with the counters
        // share_corners = 1: every bitwise-distinct lattice point of a brick once (kernels_octree.h: leaf_dense_kernel)
        used_dense = opts.share_corners == 1 && lq == 3 && lk == 4;
        if (used_dense) spec_leaf_dense(p);
        // share_corners = 2: the distinct z rows of a brick once each (kernels_octree.h: DZ) -- column bricks at four points per lane
        used_dz = opts.share_corners == 2 && lq == 3 && lk == 4;
        if (used_dz) spec_leaf_dz(p);
        if (used_dz && !p->f_leaf_dz && p->f_leaf && p->spec_leaf_k == lk) used_dz = false;  // no scratch-free build of it for this tree: every row, through the specialised kernel
#define LAUNCH_LEAF_EVAL_U(KK, WW, UU, NN, LDS)                                                                                    \
  hipLaunchKernelGGL((leaf_eval_kernel<KK, WW, UU, NN>), dim3(grid_for(bound, p->num_cu, leaf_bpc)), dim3(BLOCK), LDS, s, p->d_code, \
                     (const Cube*)q[lq & 1]->p, (unsigned long long)capq[lq & 1], lq, p->prog.nslots, ox, oy, oz, res, d_hdr,   \
                     d_rec, d_psum, (unsigned long long)nblk, d_ctr, (unsigned)(masks_valid ? 1u : 0u))
        // lq == 3 (three levels or more): a wave pass is one level-3 cube (column bricks, scalar prefetched cube load), its
        // case-count table in LDS unless that costs a workgroup per CU; else a few leaves (occupancy is no concern: table in LDS)
#define LAUNCH_LEAF_EVAL(KK, WW)                                                             \
  do {                                                                                       \
    if (lq != 3) LAUNCH_LEAF_EVAL_U(KK, WW, false, true, lds_m + 256);                       \
    else if (p->leaf_nt_in_lds()) LAUNCH_LEAF_EVAL_U(KK, WW, true, true, lds_m);             \
    else LAUNCH_LEAF_EVAL_U(KK, WW, true, false, lds_m);                                     \
  } while (0)
#define LAUNCH_LEAF_EVAL_DZ(WW, NN)                                                                                                 \
  hipLaunchKernelGGL((leaf_eval_kernel<4, WW, true, NN, false, true>), dim3(grid_for(bound, p->num_cu, leaf_bpc)), dim3(BLOCK), lds_m, s, p->d_code, \
                     (const Cube*)q[lq & 1]->p, (unsigned long long)capq[lq & 1], lq, p->prog.nslots, ox, oy, oz, res, d_hdr,   \
                     d_rec, d_psum, (unsigned long long)nblk, d_ctr, (unsigned)(masks_valid ? 1u : 0u))
        if (used_dense) {
          const size_t lds_d = p->lds_dense();
          if (p->f_leaf_dense) {
            HIP_TRYM(launch_fn(p->f_leaf_dense, grid_for(bound, p->num_cu, leaf_bpc), BLOCK, lds_d, s, (const uint32_t*)p->d_code, (const Cube*)q[lq & 1]->p,
                               (unsigned long long)capq[lq & 1], (int)p->prog.nslots, ox, oy, oz, res, d_hdr, d_rec, d_psum, (unsigned long long)nblk, d_ctr, (unsigned)(masks_valid ? 1u : 0u)));
          } else if (3 * lds_d <= (size_t)160 * 1024) {  // the interpreter's kernels: scratch-free occupancies only, passes of four points per lane
            hipLaunchKernelGGL((leaf_dense_kernel<3, true, false>), dim3(grid_for(bound, p->num_cu, leaf_bpc)), dim3(BLOCK), lds_d, s, p->d_code, (const Cube*)q[lq & 1]->p,
                               (unsigned long long)capq[lq & 1], p->prog.nslots, ox, oy, oz, res, d_hdr, d_rec, d_psum, (unsigned long long)nblk, d_ctr, (unsigned)(masks_valid ? 1u : 0u));
          } else {
            hipLaunchKernelGGL((leaf_dense_kernel<2, true, false>), dim3(grid_for(bound, p->num_cu, leaf_bpc)), dim3(BLOCK), lds_d, s, p->d_code, (const Cube*)q[lq & 1]->p,
                               (unsigned long long)capq[lq & 1], p->prog.nslots, ox, oy, oz, res, d_hdr, d_rec, d_psum, (unsigned long long)nblk, d_ctr, (unsigned)(masks_valid ? 1u : 0u));
          }
        } else if (used_dz && p->f_leaf_dz) {
          HIP_TRYM(launch_fn(p->f_leaf_dz, grid_for(bound, p->num_cu, leaf_bpc), BLOCK, lds_m, s, (const uint32_t*)p->d_code, (const Cube*)q[lq & 1]->p,
                             (unsigned long long)capq[lq & 1], (int)lq, (int)p->prog.nslots, ox, oy, oz, res, d_hdr, d_rec, d_psum,
                             (unsigned long long)nblk, d_ctr, (unsigned)(masks_valid ? 1u : 0u)));
        } else if (used_dz) {  // the interpreter's kernels (scratch-free occupancies only, as below)
          if (p->leaf_nt_in_lds()) { if (lw == 2) LAUNCH_LEAF_EVAL_DZ(2, true); else LAUNCH_LEAF_EVAL_DZ(3, true); }
          else { if (lw == 2) LAUNCH_LEAF_EVAL_DZ(2, false); else LAUNCH_LEAF_EVAL_DZ(3, false); }
        } else if (p->f_leaf && p->spec_leaf_k == lk && lq == 3) {
          HIP_TRYM(launch_fn(p->f_leaf, grid_for(bound, p->num_cu, leaf_bpc), BLOCK, lds_m, s, (const uint32_t*)p->d_code, (const Cube*)q[lq & 1]->p,
                             (unsigned long long)capq[lq & 1], (int)lq, (int)p->prog.nslots, ox, oy, oz, res, d_hdr, d_rec, d_psum,
                             (unsigned long long)nblk, d_ctr, (unsigned)(masks_valid ? 1u : 0u)));
        } else {
          // ahead-of-time kernels exist at the scratch-free occupancies only (tests/test_kernel_resources.py)
          if (lk == 4) { if (lw == 2) LAUNCH_LEAF_EVAL(4, 2); else LAUNCH_LEAF_EVAL(4, 3); }
          else if (lk == 2) LAUNCH_LEAF_EVAL(2, 3);
          else LAUNCH_LEAF_EVAL(1, 4);
        }
#undef LAUNCH_LEAF_EVAL
#undef LAUNCH_LEAF_EVAL_DZ
#undef LAUNCH_LEAF_EVAL_U
        HIP_TRYM(hipGetLastError());
        HIP_TRYM(hipEventRecord(ev3, s));
        two_kernel = true;
        if (want_recs) {
          // no marching here: the records are packed for the gather (scan_groups_kernel + pack_records_kernel), marching cubes
          // runs where they arrive (march_dense_kernel)
          uint64_t rec_cap = m->recs_cap36 * 36 / 41;  // 40 B per record + 4 B per 256 of them, rounded: stays inside the buffer
          if (rec_cap > 2) rec_cap -= 2;
          hipLaunchKernelGGL(scan_groups_kernel, dim3(1), dim3(1024), 0, s, (const unsigned long long*)d_psum, (unsigned long long)nblk, lq, d_ctr,
                             (unsigned long long*)w.grp.p, m->d_recs, (unsigned long long)rec_cap,
                             ctr_from_kernel ? hcp : (MeshCounters*)nullptr);
          HIP_TRYM(hipGetLastError());
          const uint64_t ngrp_q = (nblk_q + MARCH_GROUP - 1) / MARCH_GROUP;
          const uint64_t gmax = (uint64_t)p->num_cu * 8;
          hipLaunchKernelGGL(pack_records_kernel, dim3((unsigned)(ngrp_q < gmax ? (ngrp_q ? ngrp_q : 1) : gmax)), dim3(BLOCK), 0, s, (const uint32_t*)d_hdr,
                             (const uint32_t*)d_rec, (const unsigned long long*)w.grp.p, (unsigned long long)nblk, lq, (const MeshCounters*)d_ctr,
                             m->d_recs, (unsigned long long)rec_cap);
          ctr_on_host = ctr_from_kernel;
        } else {
        // one wave of workgroups: each takes an equal share of the records (computed on device from the group sums)
        static const int march_bpc = [] { const char* e = getenv("GSDF_HIP_MARCH_BPC"); return e ? atoi(e) : 4; }();  // tuning knob (7 fit a CU; 4 leave LDS and wave slots to the other mesh in flight: 0.503 -> 0.495 ms per mesh, and a blocking mesh loses nothing)
        const size_t lds_march = MARCH_LDS_BYTES;
        hipLaunchKernelGGL(march_records_kernel, dim3(grid_for(nblk_q, p->num_cu, march_bpc)), dim3(BLOCK), lds_march, s, d_hdr, d_rec,
                           d_psum, (unsigned long long)nblk, lq, ox, oy, oz, res, m->d_tris, (uint64_t)tcap, d_ctr,
                           ctr_from_kernel ? hcp : (MeshCounters*)nullptr);
        ctr_on_host = ctr_from_kernel;
        }
      } else if (lq == 3 && lk == 4 && opts.share_corners == 1) {
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
    // the completion event: ev2 itself when the last kernel has handed the counters to the host (nothing sits between them)
    if (!ctr_on_host) {
      HIP_TRYM(hipMemcpyAsync(&hc, d_ctr, sizeof(hc), hipMemcpyDeviceToHost, s));
      HIP_TRYM(hipEventRecord(evr, s));
    }
    ev_done = ctr_on_host ? ev2 : evr;
    return GSDF_OK;
}

int gsdf_mesh_job::finish(bool* again) {
    MeshCounters& hc = *hcp;
    HIP_TRYM(hipEventSynchronize(ev_done));
    if (hc.q_overflow) {  // a cube queue was too small: double and redo (exact: nothing was dropped silently)
      if (attempt >= 8) return (fail(GSDF_ERR_CAPACITY, "octree queue capacity exceeded"));
      qcap *= 4;
      *again = true;
      return GSDF_OK;
    }
    if (two_kernel) {  // more blocks than the record arena holds: the blocks beyond it were skipped -- repeat with room for all
      const uint64_t need = ((hc.n_level[lq] << (3 * (lq - 1))) + 63) / 64;
      if (need > nblk) {
        if (attempt >= 8) return (fail(GSDF_ERR_CAPACITY, "cut-leaf record arena capacity exceeded"));
        p->rec_blocks = need + need / 8 + 1024;
        two_kernel = false;
        *again = true;
      return GSDF_OK;
      }
    }
    if (hc.overflow && want_recs) {  // record payload too small: the scan knows the exact count
      if (attempt >= 8) return (fail(GSDF_ERR_CAPACITY, "device record buffer capacity exceeded"));
      pool_give(p->device, (float*)m->d_recs, m->recs_cap36);
      m->d_recs = nullptr; m->recs_cap36 = 0;
      want_rec_n = hc.n_cut + hc.n_cut / 16 + 1024;
      *again = true;
      return GSDF_OK;
    }
    if (hc.overflow) {  // triangle buffer too small: the kernel kept counting, so the exact size is known
      if (opts.max_tris) return (fail(GSDF_ERR_CAPACITY, "device triangle buffer capacity exceeded"));
      if (attempt >= 8) return (fail(GSDF_ERR_CAPACITY, "device triangle buffer capacity exceeded"));
      release_tris(m);
      want = hc.n_tris + hc.n_tris / 16 + 1024;
      *again = true;
      return GSDF_OK;
    }
    *again = false;
    return GSDF_OK;
}

int gsdf_mesh_job::stats() {
  MeshCounters& hc = *hcp;
  float ms01 = 0, ms12 = 0, ms13 = 0;
  HIP_TRYM(hipEventElapsedTime(&ms01, ev0, ev1));
  HIP_TRYM(hipEventElapsedTime(&ms12, ev1, ev2));
  if (two_kernel) HIP_TRYM(hipEventElapsedTime(&ms13, ev1, ev3));  // the evaluating kernel alone
  uint64_t evals_prune = 0, pruned = 0;
  for (int level = levels; level >= lq; level--) {
    evals_prune += hc.n_items[level];
    // cubes that passed their test: counted apart only where "passed" and "kept by this rank" differ (prune_kernel)
    const bool dealt_here = opts.shard_count > 1 && level == ls;
    const uint64_t passed = (level <= chain_first && !dealt_here) ? hc.n_level[level] : hc.n_pass[level];
    if (hc.n_items[level]) pruned += (hc.n_items[level] - passed) << (3 * (level - 1));  // DecomposesTo(1) = 8^(level-1)
  }
  const uint64_t n_leaves = hc.n_level[lq] << (3 * (lq - 1));
  const uint64_t evals_leaf = used_brick || ((used_dz || used_dense) && two_kernel) ? hc.n_points  // distinct lattice points (share_corners = 1) / distinct z rows of the bricks (= 2) evaluated once each
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
  if (want_recs) { m->payload = GSDF_PAYLOAD_RECORDS; m->n_recs = hc.n_cut; p->last_recs = hc.n_cut; }
  return GSDF_OK;
}
#undef HIP_TRYM

static void job_release(gsdf_mesh_job* j, bool failed = true) {
  if (!j) return;
  // the way out of a failed start / wait: a partly enqueued chain may still be running -- nothing goes back to the pool, and the
  // workspace is not handed to the next job, under it
  if (failed && j->s) { (void)hipStreamSynchronize(j->s); (void)hipGetLastError(); }
  if (j->p) j->p->job_busy[j->slot] = false;
  if (j->m) gsdf_hip_mesh_destroy(j->m);
  delete j;
}

extern "C" int gsdf_hip_mesh_octree_start(gsdf_program* p, float res, const gsdf_mesh_opts* opts_in, gsdf_mesh_job** out) {
  if (!p || !out) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  *out = nullptr;
  spec_adopt(p);  // (a background build that has finished: its kernels from here on)
  if (p->prog.is2d) return fail(GSDF_ERR_DIMENSION, "program is 2D");
  if (!(res > 0) || std::isnan(res) || std::isinf(res)) return fail(GSDF_ERR_RESOLUTION, "invalid renderer cube resolution");
  gsdf_mesh_opts opts{};
  opts.prune = 1; opts.shard_rank = 0; opts.shard_count = 1; opts.share_corners = 0;
  if (opts_in) opts = *opts_in;
  if (opts.shard_count < 1 || opts.shard_rank < 0 || opts.shard_rank >= opts.shard_count) return fail(GSDF_ERR_BAD_ARGUMENT, "bad shard rank/count");
  if (opts.payload != GSDF_PAYLOAD_TRIANGLES && opts.payload != GSDF_PAYLOAD_RECORDS) return fail(GSDF_ERR_BAD_ARGUMENT, "bad payload kind");
  const bool want_recs = opts.payload == GSDF_PAYLOAD_RECORDS;
  if (opts.share_corners < 0 || opts.share_corners > 3) return fail(GSDF_ERR_BAD_ARGUMENT, "share_corners is 0 (every corner of every leaf, as the reference), 1 (distinct lattice points of a brick), 2 (distinct z rows of a brick) or 3 (1 or 2, whichever suits the tree)");
  // 3: by the program's share of work that depends on x and y alone (the column bricks of 0 and 2 compute it once per lane for all z
  // rows, the packed points of 1 give that up): hypot / atan2 sharing instructions weighted as for the one-body leaf kernel. bolt (0)
  // and knurled-cylinder (2) gain 9 % / 17 % from 1, npt-flange (7) loses 40 % with it and breaks even with 2.
  if (opts.share_corners == 3) opts.share_corners = gsdf_dev::spec_xy_shared_weight(p->prog) < 3 ? 1 : 2;
  if (want_recs && (opts.host_output || opts.max_tris || fused_leaf()))
    return fail(GSDF_ERR_BAD_ARGUMENT, "payload = records goes with the two-kernel leaf phase only (no host_output, max_tris, fused leaf kernel)");
  HIP_TRY(hipSetDevice(p->device));

  // Octree.Reset (octreerenderer.go:71-128) + makeICube (:222-235)
  float mn[3], mx[3];
  scale_centered(p->prog.bb, 1.01f, mn, mx);
  const float longAxis = fmaxf(mx[0] - mn[0], fmaxf(mx[1] - mn[1], mx[2] - mn[2]));
  const float l2 = gsdf::log2f32(longAxis / res);
  const int levels = (int)std::ceil(l2) + 1;
  if (levels <= 1) return fail(GSDF_ERR_RESOLUTION, "resolution not fine enough for marching cubes");
  if (levels > 17) return fail(GSDF_ERR_RESOLUTION, "resolution too fine: more than 17 octree levels");

  int slot = -1;
  for (int k = 0; k < gsdf_program::kJobs; k++) if (!p->job_busy[k]) { slot = k; break; }
  if (slot < 0) return fail(GSDF_ERR_BAD_ARGUMENT, "three meshes of this program are in flight already: wait for one (gsdf_hip_mesh_octree_wait)");
  // Each slot has a workspace and a stream of its own: two chains in flight run beside each other (a caller's stream takes both).
  // Measured at npt-flange resdiv 1600, per mesh: one blocking call at a time 0.59 ms; two chains back to back on ONE stream 0.58;
  // on TWO streams 0.50 -- the centre tests and the marching kernel of one mesh (latency- and HBM-bound) run under the evaluating
  // kernel of the other. Tried and dropped: one stream for the evaluating kernels of all meshes and side streams for the stages
  // around them, ordered by events (a three-stage pipeline): 0.54, and a blocking call 0.61 -- every cross-stream event wait is
  // ~10 us on this runtime.
  hipStream_t* own = slot == 1 ? &p->stream_b : (slot == 2 ? &p->stream_c : nullptr);
  if (own && !opts.stream && !*own && hipStreamCreateWithFlags(own, hipStreamNonBlocking) != hipSuccess) {
    (void)hipGetLastError();
    return fail(GSDF_ERR_HIP, "hipStreamCreate failed");
  }
  hipStream_t s = opts.stream ? (hipStream_t)opts.stream : (own ? *own : p->stream);
  gsdf_mesh_job* j = new (std::nothrow) gsdf_mesh_job(p, slot);
  gsdf_mesh* m = new (std::nothrow) gsdf_mesh();
  if (!j || !m) { delete j; delete m; return fail(GSDF_ERR_BAD_ARGUMENT, "out of memory"); }
  j->m = m; j->opts = opts; j->res = res; j->s = s; j->levels = levels; j->want_recs = want_recs;
  j->ox = mn[0]; j->oy = mn[1]; j->oz = mn[2];
  p->job_busy[slot] = true;
  p->job_stream[slot] = s;
  m->device = p->device;
  m->stream = s;
  m->num_cu = p->num_cu;
  m->st.levels = levels;
  m->st.res = res;
  m->st.origin[0] = j->ox; m->st.origin[1] = j->oy; m->st.origin[2] = j->oz;
  auto bail = [&](int code) { job_release(j); return code; };
  static_assert(sizeof(MeshCounters) <= 2048, "counters: a 2 KB slot of the pinned block each (slot 1: DCCounters of dual contouring)");
  if (!p->h_ctr && hipHostMalloc(&p->h_ctr, 16384, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return bail(fail(GSDF_ERR_HIP, "hipHostMalloc of the counter block failed")); }
  j->hcp = (MeshCounters*)((char*)p->h_ctr + (size_t)slot * 4096);
  hipEvent_t* evs = slot == 0 ? p->ev : (slot == 1 ? p->ev_b : p->ev_c);
  for (int k = 0; k < 5; k++)
    if (!evs[k] && hipEventCreate(&evs[k]) != hipSuccess) return bail(fail(GSDF_ERR_HIP, "hipEventCreate failed"));
  j->ev0 = evs[0]; j->ev1 = evs[1]; j->ev2 = evs[2]; j->ev3 = evs[3]; j->evr = evs[4];

  // ---- level-synchronous descent from the top cube to level lq = min(levels, 3). The whole chain
  // (memset, one prune launch per level, the leaf kernel) is enqueued without a host round trip: each
  // kernel reads its input count from the previous level's counter in device memory.
  j->lq = levels < 3 ? levels : 3;
  // multi-GPU: bricks of level ls are dealt to ranks by a hash of their coordinates (brick_owner).
  j->ls = levels < j->lq + 2 ? levels : j->lq + 2;
  j->prune_cols = p->prog.nslots + p->prog.lip_depth;  // interval mode: two points per lane + the interval stack
  j->lds_prune = (size_t)(j->prune_cols > 0 ? j->prune_cols : 1) * 2 * BLOCK * sizeof(float) + PRUNE_STAGE * sizeof(Cube) + 64;
  j->pmask = opts.prune & ~GSDF_PRUNE_ASSUME_SDF;  // levels to test: 0 none, 1 all, else bit L = Level L
  if ((opts.prune & GSDF_PRUNE_ASSUME_SDF) && j->pmask == 0) j->pmask = 1;
  j->ptest = (opts.prune & GSDF_PRUNE_ASSUME_SDF) ? 2 : 1;
  p->leaf_config(&j->lk, &j->lw, &j->lds_m);
  {
    static const size_t lds_pad = [] { const char* e = getenv("GSDF_HIP_LEAF_LDS_PAD"); return e ? (size_t)atoi(e) : (size_t)0; }();  // developer knob: LDS a leaf workgroup asks for beyond its need (occupancy experiments)
    j->lds_m += lds_pad;
  }
  j->qcap = j->w.q0.cap / sizeof(Cube);
  {
    // 1 M cubes (8 MB) per queue to start with; GSDF_HIP_QCAP_MIN lowers it so that tests can drive the
    // overflow -> grow -> rerun path (the arenas only ever grow, so a handle that already meshed keeps its size)
    const char* e = getenv("GSDF_HIP_QCAP_MIN");
    const uint64_t qmin = e ? (uint64_t)strtoull(e, nullptr, 10) : ((uint64_t)1 << 20);
    if (j->qcap < qmin) j->qcap = qmin;
    if (j->qcap < 64) j->qcap = 64;
  }
  j->want = opts.max_tris;
  *j->hcp = MeshCounters{};
  j->chain_first = levels;
  if (int rc = j->enqueue()) return bail(rc);
  *out = j;
  return GSDF_OK;
}

extern "C" int gsdf_hip_mesh_octree_wait(gsdf_mesh_job* j, gsdf_mesh** out) {
  if (!j || !out) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  *out = nullptr;
  (void)hipSetDevice(j->p->device);
  for (;;) {
    bool again = false;
    if (int rc = j->finish(&again)) { job_release(j); return rc; }
    if (!again) break;
    j->attempt++;
    if (int rc = j->enqueue()) { job_release(j); return rc; }
  }
  if (int rc = j->stats()) { job_release(j); return rc; }
  *out = j->m;
  j->m = nullptr;
  job_release(j, /*failed=*/false);
  return GSDF_OK;
}

extern "C" int gsdf_hip_mesh_octree(gsdf_program* p, float res, const gsdf_mesh_opts* opts_in, gsdf_mesh** out) {
  if (!out) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  *out = nullptr;
  gsdf_mesh_job* j = nullptr;
  if (int rc = gsdf_hip_mesh_octree_start(p, res, opts_in, &j)) return rc;
  return gsdf_hip_mesh_octree_wait(j, out);
}

// ---- packed cut-leaf records -> triangles (kernels_octree.h: march_dense_kernel) ------------------------------------------
static_assert(sizeof(gsdf_dense_part) == sizeof(DensePart) && kDenseMaxParts == DENSE_MAX_PARTS, "abi_host.h mirrors kernels_octree.h");
size_t dense_parts_bytes() { return sizeof(DenseParts) + 8; }  // (+ 8: the table starts at the next 8-byte boundary behind the triangles)
void* dense_parts_at(float* d_tris, uint64_t n_tris) {  // n_tris * 36 bytes is a multiple of 4 only: the table holds 64-bit fields
  return (void*)(((uintptr_t)(d_tris + n_tris * 9) + 7) & ~(uintptr_t)7);
}
int mesh_march_dense(const uint8_t* d_buf, const gsdf_dense_part* parts, int nparts, void* d_parts, float ox, float oy, float oz, float res,
                     float* d_tris, int num_cu, hipStream_t s) {
  if (nparts < 1 || nparts > DENSE_MAX_PARTS) return fail(GSDF_ERR_BAD_ARGUMENT, "march over packed records: 1 to 64 parts");
  DenseParts h{};
  h.n = nparts;
  uint64_t chunks = 0;
  for (int i = 0; i < nparts; i++) {
    h.p[i].off = parts[i].off; h.p[i].n_recs = parts[i].n_recs; h.p[i].tri0 = parts[i].tri0;
    chunks += (parts[i].n_recs + DENSE_CHUNK - 1) / DENSE_CHUNK;
    if (dense_bytes(parts[i].n_recs) != dense_payload_bytes(parts[i].n_recs)) return fail(GSDF_ERR_BAD_ARGUMENT, "internal: payload layouts differ");
  }
  if (chunks == 0) return GSDF_OK;
  // the table is small and the copy's source is this stack frame: hipMemcpyAsync from pageable memory stages it before returning
  HIP_TRY(hipMemcpyAsync(d_parts, &h, sizeof h, hipMemcpyHostToDevice, s));
  static const int march_bpc = [] { const char* e = getenv("GSDF_HIP_MARCH_BPC"); return e ? atoi(e) : 4; }();  // tuning knob (7 fit a CU; 4 leave LDS and wave slots to the other mesh in flight: 0.503 -> 0.495 ms per mesh, and a blocking mesh loses nothing)
  const uint64_t gmax = (uint64_t)num_cu * (uint64_t)(march_bpc > 0 ? march_bpc : 4);
  const size_t lds = MARCH_DENSE_LDS_BYTES;
  hipLaunchKernelGGL(march_dense_kernel, dim3((unsigned)(chunks < gmax ? chunks : gmax)), dim3(BLOCK), lds, s, d_buf, (const DenseParts*)d_parts, ox, oy, oz, res, d_tris);
  HIP_TRY(hipGetLastError());
  return GSDF_OK;
}

extern "C" int gsdf_hip_mesh_stage_ms(const gsdf_mesh* m, double* ms, const char** names, int cap, int* n) {
  if (!m || !n) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  *n = m->n_stages;
  for (int k = 0; k < m->n_stages && k < cap; k++) {
    if (ms) ms[k] = m->stage_ms[k];
    if (names) names[k] = m->stage_name[k];
  }
  return GSDF_OK;
}

extern "C" int gsdf_hip_mesh_payload(const gsdf_mesh* m, uint64_t* n_records, uint64_t* payload_bytes) {
  if (!m) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  const bool r = m->payload == GSDF_PAYLOAD_RECORDS;
  if (n_records) *n_records = r ? m->n_recs : 0;
  if (payload_bytes) *payload_bytes = r ? dense_bytes(m->n_recs) : 0;
  return m->payload;
}

extern "C" int gsdf_hip_mesh_march(gsdf_mesh* m) {
  if (!m) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  if (m->payload != GSDF_PAYLOAD_RECORDS) return GSDF_OK;
  if (m->inflight.load() > 0) return fail(GSDF_ERR_BAD_ARGUMENT, "the mesh is being gathered: wait for the gather first");
  HIP_TRY(hipSetDevice(m->device));
  const uint64_t n = m->st.n_tris;
  if (n) {
    const uint64_t units = n + (dense_parts_bytes() + 35) / 36;  // the parts table rides behind the triangles
    m->d_tris = pool_take(m->device, units, &m->cap);
    if (!m->d_tris) {
      if (hipMalloc((void**)&m->d_tris, units * 36) != hipSuccess) { (void)hipGetLastError(); return fail(GSDF_ERR_HIP, "hipMalloc of the triangle buffer failed"); }
      m->cap = units;
    }
    const gsdf_dense_part part{0, m->n_recs, 0};
    hipStream_t rs = mesh_stream(m);
    const int rc = mesh_march_dense(m->d_recs, &part, 1, dense_parts_at(m->d_tris, n), m->st.origin[0], m->st.origin[1], m->st.origin[2], m->st.res, m->d_tris,
                                    m->num_cu, rs);
    if (rc) { release_tris(m); return rc; }
    hipError_t e = hipStreamSynchronize(rs);
    if (e != hipSuccess) { release_tris(m); return fail(GSDF_ERR_HIP, std::string("march over the records: ") + hipGetErrorString(e)); }
  }
  pool_give(m->device, (float*)m->d_recs, m->recs_cap36);
  m->d_recs = nullptr; m->recs_cap36 = 0; m->n_recs = 0;
  m->payload = GSDF_PAYLOAD_TRIANGLES;
  return GSDF_OK;
}

// glrender.DualContourRenderer.Reset + RenderAll with DualContourLeastSquares on device.
extern "C" int gsdf_hip_mesh_dualcontour(gsdf_program* p, float res, int chiseled, int shard_rank, int shard_count, void* stream,
                                         gsdf_mesh** out) {
  if (!p || !out) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  *out = nullptr;
  spec_adopt(p);  // (a background build that has finished: its kernels from here on)
  if (p->prog.is2d) return fail(GSDF_ERR_DIMENSION, "program is 2D");
  if (!(res > 0) || std::isnan(res) || std::isinf(res)) return fail(GSDF_ERR_RESOLUTION, "invalid renderer cube resolution");
  if (shard_count < 1 || shard_rank < 0 || shard_rank >= shard_count) return fail(GSDF_ERR_BAD_ARGUMENT, "bad shard rank/count");
  if (p->mesh_in_flight()) return fail(GSDF_ERR_BAD_ARGUMENT, "an octree mesh of this program is in flight: wait for it first (shared workspace)");
  HIP_TRY(hipSetDevice(p->device));
  hipStream_t s = stream ? (hipStream_t)stream : p->stream;
  // Reset (dual_contour.go:26-41): bounds shifted by -res/2, makeICube
  const float sub = res / 2;
  float mn[3], mx[3];
  for (int a = 0; a < 3; a++) { mn[a] = p->prog.bb[a] + -sub; mx[a] = p->prog.bb[a + 3] + -sub; }
  const float longAxis = fmaxf(mx[0] - mn[0], fmaxf(mx[1] - mn[1], mx[2] - mn[2]));
  const int levels = (int)std::ceil(gsdf::log2f32(longAxis / res)) + 1;
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
  for (auto& e : p->ev_b)
    if (!e) HIP_TRYM(hipEventCreate(&e));
  // Workspace lives in the program handle (grow-only): the index grid alone is 4.3 GB at 1024^3 cells, and a
  // hipMalloc/hipFree pair of that size per mesh cost more wall time than the whole device pass.
  gsdf_program::Arena &grid = p->dc_grid, &d2 = p->dc_dist, &f2 = p->dc_fv, &n2 = p->dc_nrm, &e2 = p->dc_edge;
  HIP_TRYM(grid.ensure(ncell * sizeof(int)));
  HIP_TRYM(p->ctr.ensure(sizeof(MeshCounters) > sizeof(DCCounters) ? sizeof(MeshCounters) : sizeof(DCCounters)));
  DCCounters* d_ctr = (DCCounters*)p->ctr.p;
  const int lk = p->batch_k();
  // Kept cubes hug the surface: ~ c * n^2 of the n^3 lattice. Start from the previous pass on this handle, else from
  // 12 n^2 (an overflow repeats the full-lattice origin pass, so be generous: 84 B per cube).
  uint64_t ccap = p->last_dc_cubes ? p->last_dc_cubes + p->last_dc_cubes / 8 + 4096 : (uint64_t)12 << (2 * nshift);
  if (ccap < (1u << 20)) ccap = 1u << 20;
  DCCounters hc{};
  uint64_t n_cubes = 0, n_edges = 0, n_cube_runs = 0, n_edge_runs = 0, last_cap = 0;
  const float h = (chiseled ? (float)1e-4 : (float)2e-8) * 0.5f;  // NormalsCentralDiff: step *= 0.5
  const float sqrtLambda = chiseled ? (float)(std::sqrt(1e-5) * 1e-4) : (float)std::sqrt(1e-5);
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
  const uint64_t ntiles = (uint64_t)tn[0] * tn[1] * tn[2];  // of the origin sweep
  for (int attempt = 0;; attempt++) {
    // the lists are kept in DC_PARTS parts (kernels_dc.h), an equal share of the arrays each; a part cannot hold more than the cells
    // of the tiles that go to it
    const uint64_t cmax = DC_PARTS * ((ntiles + DC_PARTS - 1) / DC_PARTS) * (uint64_t)(256 * lk);
    if (ccap > cmax) ccap = cmax;
    ccap = (ccap + DC_PARTS - 1) / DC_PARTS * DC_PARTS;
    if (ccap < DC_PARTS) ccap = DC_PARTS;
    const uint64_t ecap = 3 * ccap, tcap = 2 * ecap;
    // run descriptors (kernels_dc.h: DC_DESC): the origin sweep's follow the cubes in q0, the edge stage's have an arena of their own
    const uint64_t crun_cap = DC_PARTS * dc_cube_run_seg(ntiles), erun_cap = DC_PARTS * dc_edge_run_seg(ccap, ntiles);
    HIP_TRYM(p->q0.ensure((ccap + crun_cap) * sizeof(Cube) + ccap * sizeof(float)));  // (+ the sweep's distance of every kept cube, behind the descriptors)
    HIP_TRYM(p->dc_erun.ensure(erun_cap * sizeof(unsigned long long)));
    HIP_TRYM(p->dc_flag.ensure(ccap));  // "this cube is placed", a byte per cube: set by the edge stage, read by the placement stage
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
    HIP_TRYM(hipMemsetAsync(p->dc_flag.p, 0, ccap, s));
    if (shard_count > 1) HIP_TRYM(hipMemsetAsync(grid.p, 0xff, ncell * sizeof(int), s));  // cells outside the slab read as empty
    HIP_TRYM(hipEventRecord(p->ev[0], s));
    static const int origin_grid = [] { const char* e = getenv("GSDF_HIP_DC_ORIGIN_GRID"); return e ? atoi(e) : 0; }();  // developer knob: workgroups of the origin sweep (1 = tiles strictly in order)
    unsigned g1 = origin_grid > 0 ? (unsigned)origin_grid : grid_for((nslab + lk - 1) / lk, p->num_cu, 32);
    g1 = (g1 + DC_PARTS - 1) / DC_PARTS * DC_PARTS;  // tile T goes to workgroup T % g1, hence to part T % DC_PARTS: what bounds a part's run descriptors (dc_cube_run_seg)
#define LAUNCH_O(KK, WW) hipLaunchKernelGGL((dc_origin_kernel<KK, WW>), dim3(g1), dim3(BLOCK), p->lds_bytes(KK) + 32, s, p->d_code, p->prog.nslots, nshift, ox, oy, oz, res, (int*)grid.p, (Cube*)p->q0.p, (unsigned long long)ccap, zlo, zhi, ub, eb[0], eb[1], eb[2], eb[3], eb[4], eb[5], t0[0], t0[1], t0[2], tn[0], tn[1], tn[2], d_ctr, d_keep)
    // Stage 0: the interval test of the sweep's blocks (kernels_dc.h: dc_block_test_kernel), one lane per block of 8 x 8 x K origins.
    // GSDF_HIP_NO_DC_BLOCK_TEST=1: every block is evaluated (developer knob: A/B timing, cross-check in the tests).
    static const bool block_test_off = [] { const char* e = getenv("GSDF_HIP_NO_DC_BLOCK_TEST"); return e && atoi(e) != 0; }();
    const uint32_t* d_keep = nullptr;
    {
      const uint64_t nblk = (uint64_t)tn[0] * tn[1] * tn[2] * 4ull;
      if (!block_test_off && nblk > 0 && p->dc_tile.ensure((size_t)nblk * sizeof(uint32_t)) == hipSuccess) {
        const int cols = p->prog.nslots + p->prog.lip_depth;
        const size_t lds_t = (size_t)(cols > 0 ? cols : 1) * 2 * BLOCK * sizeof(float);
        const unsigned gt = grid_for(nblk, p->num_cu, 8);
        if (p->f_dc_block_test) HIP_TRYM(launch_fn(p->f_dc_block_test, gt, BLOCK, lds_t, s, (const uint32_t*)p->d_code, (int)cols, (int)p->prog.nslots, (int)lk, ox, oy, oz, res,
                                                    (unsigned)zlo, t0[0], t0[1], t0[2], tn[0], tn[1], tn[2], (uint32_t*)p->dc_tile.p));
        else hipLaunchKernelGGL(dc_block_test_kernel, dim3(gt), dim3(BLOCK), lds_t, s, p->d_code, cols, p->prog.nslots, lk, ox, oy, oz, res, zlo, t0[0], t0[1], t0[2], tn[0], tn[1],
                                tn[2], (uint32_t*)p->dc_tile.p);
        HIP_TRYM(hipGetLastError());
        // ... and "no cube here" in the cleared tiles next to evaluated ones: the cells the later stages read and the sweep does not write
        hipLaunchKernelGGL(dc_grid_clear_kernel, dim3(grid_for((uint64_t)((tn[0] + 3) / 4) * ((tn[1] + 3) / 4) * ((tn[2] + 3) / 4) * BLOCK, p->num_cu, 16)), dim3(BLOCK), 0, s, (int*)grid.p, nshift, lk, zlo, zhi, t0[0], t0[1], t0[2], tn[0], tn[1],
                           tn[2], (const uint32_t*)p->dc_tile.p);
        HIP_TRYM(hipGetLastError());
        d_keep = (const uint32_t*)p->dc_tile.p;
      } else {
        (void)hipGetLastError();
      }
    }
    if (p->f_dc_origin) HIP_TRYM(launch_fn(p->f_dc_origin, g1, BLOCK, p->lds_bytes(lk) + 32, s, (const uint32_t*)p->d_code, (int)p->prog.nslots, (int)nshift, ox, oy, oz, res, (int*)grid.p, (Cube*)p->q0.p, (unsigned long long)ccap, (unsigned)zlo, (unsigned)zhi, ub, eb[0], eb[1], eb[2], eb[3], eb[4], eb[5], t0[0], t0[1], t0[2], tn[0], tn[1], tn[2], d_ctr, d_keep));
    else
    if (lk == 4) LAUNCH_O(4, 3);  // <4, 4> needs scratch: not built (see fn_scratch_bytes)
    else if (lk == 2) LAUNCH_O(2, 3);  // (<2, 4> sits at its register budget: with the block verdicts' pointer it spills four VGPRs, and a kernel with scratch is not shipped)
    else LAUNCH_O(1, 4);
#undef LAUNCH_O
    HIP_TRYM(hipGetLastError());
    HIP_TRYM(hipEventRecord(p->ev_b[0], s));  // origin sweep done
    if (p->lds_bytes(4) > 150 * 1024) return bail(fail(GSDF_ERR_BAD_TREE, "tree needs too much LDS scratch for the dual contouring edge pass"));
    // (a multiple of DC_PARTS like the sweep's grid: run r goes to workgroup r % grid, hence to the edge list's part r % DC_PARTS --
    // what bounds a part's edge-run descriptors, dc_edge_run_seg)
    const unsigned g_edges = (grid_for(ccap, p->num_cu, 8) + DC_PARTS - 1) / DC_PARTS * DC_PARTS;
    if (p->f_dc_edges) HIP_TRYM(launch_fn(p->f_dc_edges, g_edges, BLOCK, p->lds_bytes(3), s, (const uint32_t*)p->d_code, (const Cube*)p->q0.p,
                       (unsigned long long)ccap, ox, oy, oz, res, (float4*)d2.p, (float*)f2.p, (unsigned*)e2.p, (unsigned long long)ecap, (unsigned long long*)p->dc_erun.p, (unsigned long long)ntiles, (const int*)grid.p, (int)nshift, (unsigned char*)p->dc_flag.p, d_ctr));
    else
    hipLaunchKernelGGL(dc_edges_kernel, dim3(g_edges), dim3(BLOCK), p->lds_bytes(3), s, p->d_code, (const Cube*)p->q0.p,
                       (unsigned long long)ccap, ox, oy, oz, res, (float4*)d2.p, (float*)f2.p, (unsigned*)e2.p, (unsigned long long)ecap, (unsigned long long*)p->dc_erun.p, (unsigned long long)ntiles, (const int*)grid.p, (int)nshift, (unsigned char*)p->dc_flag.p, d_ctr);
    HIP_TRYM(hipGetLastError());
    HIP_TRYM(hipEventRecord(p->ev_b[1], s));  // edges done
    if (p->f_dc_normals) HIP_TRYM(launch_fn(p->f_dc_normals, grid_for(ecap, p->num_cu, 8), BLOCK, p->lds_bytes(2), s, (const uint32_t*)p->d_code, (const Cube*)p->q0.p,
                       (const float4*)d2.p, (const unsigned*)e2.p, (const unsigned long long*)p->dc_erun.p, (unsigned long long)ccap, (unsigned long long)ntiles, ox, oy, oz, res, h, (float*)n2.p, d_ctr));
    else
    hipLaunchKernelGGL(dc_normals_kernel, dim3(grid_for(ecap, p->num_cu, 8)), dim3(BLOCK), p->lds_bytes(2), s, p->d_code, (const Cube*)p->q0.p,
                       (const float4*)d2.p, (const unsigned*)e2.p, (const unsigned long long*)p->dc_erun.p, (unsigned long long)ccap, (unsigned long long)ntiles, ox, oy, oz, res, h, (float*)n2.p, d_ctr);
    HIP_TRYM(hipGetLastError());
    HIP_TRYM(hipEventRecord(p->ev_b[2], s));  // normals done
    hipLaunchKernelGGL(dc_place_kernel, dim3(grid_for(ccap * 4, p->num_cu, 16)), dim3(DC_BLOCK), 0, s, (const Cube*)p->q0.p,
                       (unsigned long long)ccap, (const float4*)d2.p, (const int*)grid.p, (const float*)n2.p, nshift, ox, oy, oz, res,
                       sqrtLambda, (float*)f2.p, zown_hi, (const unsigned char*)p->dc_flag.p, d_ctr);
    HIP_TRYM(hipGetLastError());
    HIP_TRYM(hipEventRecord(p->ev_b[3], s));  // placement done
    hipLaunchKernelGGL(dc_quads_kernel, dim3(grid_for(ecap, p->num_cu, 8)), dim3(BLOCK), 0, s, (const Cube*)p->q0.p, (const float4*)d2.p,
                       (const unsigned*)e2.p, (unsigned long long)ecap, (const int*)grid.p, (const float*)f2.p, nshift, zown_lo, zown_hi,
                       m->d_tris, (unsigned long long)m->cap, d_ctr);
    HIP_TRYM(hipGetLastError());
    HIP_TRYM(hipEventRecord(p->ev[1], s));
    HIP_TRYM(hipMemcpyAsync(&hc, d_ctr, sizeof(hc), hipMemcpyDeviceToHost, s));
    HIP_TRYM(hipStreamSynchronize(s));
    n_cubes = n_edges = n_cube_runs = n_edge_runs = 0;
    uint64_t max_part = 0;  // the fullest part of the cubes' list: what the capacity has to hold DC_PARTS times
    for (int k = 0; k < DC_PARTS; k++) {
      const uint64_t c = DC_W_COUNT(hc.cubes_w[k * DC_WORD_STRIDE]);
      n_cubes += c; n_edges += DC_W_COUNT(hc.edges_w[k * DC_WORD_STRIDE]);
      n_cube_runs += DC_W_RUNS(hc.cubes_w[k * DC_WORD_STRIDE]); n_edge_runs += DC_W_RUNS(hc.edges_w[k * DC_WORD_STRIDE]);
      if (c > max_part) max_part = c;
    }
    if (hc.q_overflow || hc.t_overflow || max_part > ccap / DC_PARTS) {
      if (attempt >= 8 || ccap >= cmax) return bail(fail(GSDF_ERR_CAPACITY, "dual contouring queue capacity exceeded"));
      // the origin pass keeps counting past the capacity, so the exact number of kept cubes is known (part by part)
      ccap = max_part > ccap / DC_PARTS ? DC_PARTS * (max_part + max_part / 16 + 512) : ccap * 2;
      continue;
    }
    last_cap = DC_PARTS * max_part;
    break;
  }
  float ms = 0;
  HIP_TRYM(hipEventElapsedTime(&ms, p->ev[0], p->ev[1]));
  {  // per stage (gsdf_hip_mesh_stage_ms): origin sweep, edges, normals, placement, quads
    hipEvent_t cut[6] = {p->ev[0], p->ev_b[0], p->ev_b[1], p->ev_b[2], p->ev_b[3], p->ev[1]};
    static const char* const names[5] = {"dc_origin", "dc_edges", "dc_normals", "dc_place", "dc_quads"};
    m->n_stages = 5;
    for (int k = 0; k < 5; k++) {
      float t = 0;
      HIP_TRYM(hipEventElapsedTime(&t, cut[k], cut[k + 1]));
      m->stage_ms[k] = t;
      m->stage_name[k] = names[k];
    }
  }
  {
    static const bool dbg = [] { const char* e = getenv("GSDF_HIP_DC_DEBUG"); return e && atoi(e) != 0; }();  // developer knob
    if (dbg) fprintf(stderr, "dc: cubes %llu in %llu runs, edges %llu in %llu runs\n", (unsigned long long)n_cubes, (unsigned long long)n_cube_runs,
                     (unsigned long long)n_edges, (unsigned long long)n_edge_runs);
  }
  p->last_dc_cubes = last_cap;  // (DC_PARTS times the fullest part: what the next mesh of this handle sizes its arrays by)
  m->st.n_tris = 2 * hc.n_tris;  // quads -> 2 triangles
  for (int k = 0; k < 64; k++) hc.n_origin_evals += hc.n_origin_part[k * 8];
  m->st.evals = hc.n_origin_evals + 3 * n_cubes + 6 * n_edges;  // (performed: a kept cube's origin is the sweep's evaluation, not repeated by the edge stage)
  m->st.evals_prune = hc.n_origin_evals;  // of the nslab lattice cells; the rest lay outside the exact box by > 2 res
  m->st.evals_leaf = 3 * n_cubes + 6 * n_edges;
  m->st.pruned_leaves = nslab - n_cubes;
  m->st.leaf_cubes = n_cubes;
  m->st.active_leaves = n_edges;
  m->st.ms_total = ms;
  p->evals += m->st.evals;
  *out = m;
  return GSDF_OK;
#undef HIP_TRYM
}

// glrender.FlatRenderer (flatrenderer.go:36-256) on device: Reset's lattice, evalGrid into a dense grid in HBM,
// ReadTriangles as one marching-cubes pass over every cube. Multi-GPU: z-slabs of cubes like the reference's goroutines
// (:120-122); a rank evaluates the planes its cubes touch (one shared plane per boundary is recomputed, nothing exchanged).
// minecraftRender (glrender/dual_contour.go:297-403): the axis-aligned faces between cubes whose origins lie on different sides of
// the surface. Unexported in the reference and used by one test; here for completeness of row a20. Three steps on the caller's
// stream: the four positions of every level-1 cube to HBM, the program's ordinary Evaluate over them (gsdf_hip_eval3_dev: whatever
// kernels the handle runs), one pass that counts the faces and one that writes them.
extern "C" int gsdf_hip_mesh_minecraft(gsdf_program* p, float res, void* stream, gsdf_mesh** out) {
  if (!p || !out) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  *out = nullptr;
  if (p->prog.is2d) return fail(GSDF_ERR_DIMENSION, "program is 2D");
  if (!(res > 0) || std::isnan(res) || std::isinf(res)) return fail(GSDF_ERR_RESOLUTION, "invalid renderer cube resolution");
  if (p->mesh_in_flight()) return fail(GSDF_ERR_BAD_ARGUMENT, "an octree mesh of this program is in flight: wait for it first (shared workspace)");
  HIP_TRY(hipSetDevice(p->device));
  hipStream_t s = stream ? (hipStream_t)stream : p->stream;
  // makeICube over the SDF's own bounds (dual_contour.go:298-299; glrender.go:222-235)
  const float* bb = p->prog.bb;
  const float longAxis = fmaxf(bb[3] - bb[0], fmaxf(bb[4] - bb[1], bb[5] - bb[2]));
  const int levels = (int)std::ceil(gsdf::log2f32(longAxis / res)) + 1;
  if (levels <= 1) return fail(GSDF_ERR_RESOLUTION, "resolution not fine enough for marching cubes");
  if (levels > 9) return fail(GSDF_ERR_RESOLUTION, "minecraft render lattice too large: more than 9 octree levels (every cube of the lattice is evaluated)");
  const int nshift = levels - 1;
  const uint64_t n_cubes = (uint64_t)1 << (3 * nshift);
  gsdf_mesh* m = new (std::nothrow) gsdf_mesh();
  if (!m) return fail(GSDF_ERR_BAD_ARGUMENT, "out of memory");
  m->device = p->device; m->stream = s;
  m->st.levels = levels; m->st.res = res;
  m->st.origin[0] = bb[0]; m->st.origin[1] = bb[1]; m->st.origin[2] = bb[2];
  float* d_pos = nullptr;
  float* d_dist = nullptr;
  unsigned long long* d_ctr = nullptr;
  auto bail = [&](int code) {
    if (d_pos) (void)hipFree(d_pos);
    if (d_dist) (void)hipFree(d_dist);
    if (d_ctr) (void)hipFree(d_ctr);
    gsdf_hip_mesh_destroy(m);
    return code;
  };
#define HIP_TRYM(expr)                                                                                          \
  do {                                                                                                          \
    hipError_t _e = (expr);                                                                                     \
    if (_e != hipSuccess) return bail(fail(GSDF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e))); \
  } while (0)
  HIP_TRYM(hipMalloc((void**)&d_pos, n_cubes * 48));
  HIP_TRYM(hipMalloc((void**)&d_dist, n_cubes * 16));
  HIP_TRYM(hipMalloc((void**)&d_ctr, 8));
  const unsigned grid = grid_for(n_cubes, p->num_cu, 16);
  hipLaunchKernelGGL(mcr_positions_kernel, dim3(grid), dim3(BLOCK), 0, s, bb[0], bb[1], bb[2], res, nshift, n_cubes, d_pos);
  HIP_TRYM(hipGetLastError());
  if (int rc = gsdf_hip_eval3_dev(p, d_pos, 12, d_dist, (size_t)(4 * n_cubes), (void*)s)) return bail(rc);
  unsigned long long n_tris = 0;
  for (int pass = 0; pass < 2; pass++) {  // count, then emit into a buffer of exactly that size
    HIP_TRYM(hipMemsetAsync(d_ctr, 0, 8, s));
    hipLaunchKernelGGL(mcr_faces_kernel, dim3(grid), dim3(BLOCK), 0, s, (const float*)d_dist, bb[0], bb[1], bb[2], res, nshift, n_cubes,
                       pass == 0 ? (float*)nullptr : m->d_tris, (unsigned long long)m->cap, d_ctr);
    HIP_TRYM(hipGetLastError());
    HIP_TRYM(hipMemcpyAsync(&n_tris, d_ctr, 8, hipMemcpyDeviceToHost, s));
    HIP_TRYM(hipStreamSynchronize(s));
    if (pass == 0) {
      if (n_tris == 0) break;
      m->d_tris = pool_take(p->device, n_tris, &m->cap);
      if (!m->d_tris) { HIP_TRYM(hipMalloc((void**)&m->d_tris, n_tris * 36)); m->cap = n_tris; }
    }
  }
  m->st.n_tris = n_tris;
  m->st.evals = 4 * n_cubes;
  m->st.leaf_cubes = n_cubes;
  (void)hipFree(d_pos); (void)hipFree(d_dist); (void)hipFree(d_ctr);
#undef HIP_TRYM
  *out = m;
  return GSDF_OK;
}

extern "C" int gsdf_hip_mesh_flat(gsdf_program* p, float res, int shard_rank, int shard_count, void* stream, gsdf_mesh** out) {
  if (!p || !out) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  *out = nullptr;
  spec_adopt(p);  // (a background build that has finished: its kernels from here on)
  if (p->prog.is2d) return fail(GSDF_ERR_DIMENSION, "program is 2D");
  if (!(res > 0) || std::isnan(res) || std::isinf(res)) return fail(GSDF_ERR_RESOLUTION, "invalid renderer cube resolution");
  if (shard_count < 1 || shard_rank < 0 || shard_rank >= shard_count) return fail(GSDF_ERR_BAD_ARGUMENT, "bad shard rank/count");
  if (p->mesh_in_flight()) return fail(GSDF_ERR_BAD_ARGUMENT, "an octree mesh of this program is in flight: wait for it first (shared workspace)");
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

static const char* const kRecordsMsg = "the mesh holds cut-leaf records, not triangles yet: gsdf_hip_mesh_march it, or gather it";
extern "C" int gsdf_hip_mesh_stats_get(const gsdf_mesh* m, gsdf_mesh_stats* st) {
  if (!m || !st) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  *st = m->st;
  return GSDF_OK;
}
extern "C" int gsdf_hip_mesh_read(const gsdf_mesh* m, uint64_t first, uint64_t count, float* dst) {
  if (!m || (!dst && count)) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  if (m->payload == GSDF_PAYLOAD_RECORDS) return fail(GSDF_ERR_BAD_ARGUMENT, kRecordsMsg);
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

extern "C" int gsdf_hip_mesh_host_tris(gsdf_mesh* m, const float** tris) {
  if (!m || !tris) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  *tris = nullptr;
  if (m->payload == GSDF_PAYLOAD_RECORDS) return fail(GSDF_ERR_BAD_ARGUMENT, kRecordsMsg);
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
  if (m->payload == GSDF_PAYLOAD_RECORDS) return fail(GSDF_ERR_BAD_ARGUMENT, kRecordsMsg);
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

static void mesh_free(gsdf_mesh* m) {
  release_tris(m);
  pool_give(m->device, (float*)m->d_recs, m->recs_cap36);
  hpool_give(m->h_tris, m->h_tris_cap);
  hpool_give(m->h_stl, m->h_stl_cap);
  if (m->rstream) (void)hipStreamDestroy(m->rstream);
  delete m;
}
extern "C" void gsdf_hip_mesh_destroy(gsdf_mesh* m) {
  if (!m) return;
  // a gather may still be reading the buffers (gsdf_hip_mesh_gatherv_start .. _wait): the last reference frees them
  if (m->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) mesh_free(m);
}
void mesh_inflight_done(gsdf_mesh* m) {
  m->inflight.fetch_sub(1);
  if (m->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) mesh_free(m);
}
