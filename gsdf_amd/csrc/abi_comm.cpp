// abi_comm.cpp -- C ABI (include/gsdf_hip.h), multi-GPU side: the communicator and the variable-length gather of the ranks'
// results. Host code only; the kernel that runs on gathered records is launched through abi_mesh.hip.
//
// One process per GPU. The meshers shard with no data-path collective (brick_owner / z-slabs); the one exchange is the final
// variable-length gather, done here so that a Go (or C) caller of this ABI has the multi-GPU path without any Python:
//   1. an all-gather of four numbers per rank (payload kind, triangles, records, payload bytes);
//   2. gsdf_hip_gather_plan -- a pure function of those counts -- lists what this rank copies, sends and receives: every rank's
//      payload lands at offset sum(bytes of the ranks before it), no padding, no staging copies;
//   3. the plan runs as ONE group of point-to-point transfers on the communicator's own stream. xGMI is point to point (one link
//      per peer), so in mode ALL a rank puts its shard on each of its world-1 links at once and receives the others' the same way:
//      the pattern the wire is built for (a ring broadcast per rank would walk every shard round the ring);
//   4. a payload of cut-leaf records (gsdf_mesh_opts.payload) is marched into triangles on the ranks that received it.
// Transports: RCCL (librccl, loaded at first use by soname -- the library has no link-time dependency on it and single-GPU users
// never load it), and an in-process loopback (GSDF_HIP_COMM=loopback: the ranks are threads of one process on one GPU, a transfer
// is a device-to-device copy ordered by events) with which the whole path -- counts, plan, transfers, marching -- runs at world
// sizes 2..64 on a one-GPU box (tests/test_gpu_gather.py); only librccl's own send/recv is not exercised by it.
#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <thread>

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <vector>

#include "abi_host.h"

// ---------------------------------------------------------------------------------------------
// the plan (pure; tests/test_gather_gloo.py executes it with gloo send / recv on CPU)
// ---------------------------------------------------------------------------------------------
extern "C" int gsdf_hip_gather_plan(const uint64_t* bytes_per_rank, int world, int rank, int mode, int root, gsdf_gather_op* ops, size_t ops_cap,
                                    size_t* n_ops, uint64_t* total_bytes) {
  if (n_ops) *n_ops = 0;
  if (total_bytes) *total_bytes = 0;
  if (!bytes_per_rank || world < 1 || rank < 0 || rank >= world) return fail(GSDF_ERR_BAD_ARGUMENT, "bad rank / world size");
  if (mode != GSDF_GATHER_ALL && mode != GSDF_GATHER_ROOT && mode != GSDF_GATHER_NONE) return fail(GSDF_ERR_BAD_ARGUMENT, "bad gather mode");
  if (mode == GSDF_GATHER_ROOT && (root < 0 || root >= world)) return fail(GSDF_ERR_BAD_ARGUMENT, "bad root rank");
  std::vector<uint64_t> off((size_t)world + 1, 0);
  for (int r = 0; r < world; r++) off[(size_t)r + 1] = off[(size_t)r] + bytes_per_rank[r];
  const bool receives = mode == GSDF_GATHER_ALL || (mode == GSDF_GATHER_ROOT && rank == root);
  if (total_bytes) *total_bytes = receives ? off[(size_t)world] : 0;
  size_t n = 0;
  bool shorted = false;
  auto put = [&](int kind, int peer, uint64_t src_off, uint64_t dst_off, uint64_t bytes) {
    if (bytes == 0) return;  // a rank without surface takes no part
    if (ops && n < ops_cap) { ops[n].kind = kind; ops[n].peer = peer; ops[n].src_off = src_off; ops[n].dst_off = dst_off; ops[n].bytes = bytes; }
    else if (ops) shorted = true;
    n++;
  };
  const uint64_t mine = bytes_per_rank[rank];
  if (mode == GSDF_GATHER_ALL) {
    put(GSDF_GOP_COPY, rank, 0, off[(size_t)rank], mine);
    // peers in rotated order (rank+1, rank+2, ...): at every step of the list the world's sends go to distinct receivers
    for (int d = 1; d < world; d++) {
      const int to = (rank + d) % world, from = (rank - d + world) % world;
      put(GSDF_GOP_SEND, to, 0, 0, mine);
      put(GSDF_GOP_RECV, from, 0, off[(size_t)from], bytes_per_rank[from]);
    }
  } else if (mode == GSDF_GATHER_ROOT) {
    if (rank == root) {
      put(GSDF_GOP_COPY, rank, 0, off[(size_t)rank], mine);
      for (int d = 1; d < world; d++) {
        const int from = (rank - d + world) % world;
        put(GSDF_GOP_RECV, from, 0, off[(size_t)from], bytes_per_rank[from]);
      }
    } else {
      put(GSDF_GOP_SEND, root, 0, 0, mine);
    }
  }
  if (n_ops) *n_ops = n;
  if (shorted) return fail(GSDF_ERR_SHORT_BUFFER, "short buffer");
  return GSDF_OK;
}

// ---------------------------------------------------------------------------------------------
// transports
// ---------------------------------------------------------------------------------------------
namespace {
struct Transport {
  virtual ~Transport() {}
  // every rank contributes n u64 at d_send; d_recv receives world * n, rank-major (device memory, on stream s)
  virtual int all_gather_u64(const unsigned long long* d_send, unsigned long long* d_recv, size_t n, hipStream_t s) = 0;
  virtual int all_reduce_sum_u64(unsigned long long* d_buf, size_t n, hipStream_t s) = 0;
  virtual int group_start() = 0;
  virtual int send(const void* d_src, size_t bytes, int peer, hipStream_t s) = 0;
  virtual int recv(void* d_dst, size_t bytes, int peer, hipStream_t s) = 0;
  virtual int group_end(hipStream_t s) = 0;
  virtual const char* name() const = 0;
};

// ---- RCCL ---------------------------------------------------------------------------------------
struct RcclApi {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
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
    GSDF_RCCL_SYM(AllGather, "ncclAllGather") GSDF_RCCL_SYM(AllReduce, "ncclAllReduce")
    GSDF_RCCL_SYM(Send, "ncclSend") GSDF_RCCL_SYM(Recv, "ncclRecv")
    GSDF_RCCL_SYM(GroupStart, "ncclGroupStart") GSDF_RCCL_SYM(GroupEnd, "ncclGroupEnd") GSDF_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef GSDF_RCCL_SYM
  });
  return &api;
}
struct RcclTransport final : Transport {
  RcclApi* R;
  ncclComm_t comm = nullptr;
  explicit RcclTransport(RcclApi* r) : R(r) {}
  ~RcclTransport() override { if (comm && R->CommDestroy) (void)R->CommDestroy(comm); }
  int chk(ncclResult_t r, const char* what) { return r == ncclSuccess ? GSDF_OK : fail(GSDF_ERR_HIP, std::string(what) + ": " + (R->GetErrorString ? R->GetErrorString(r) : "rccl error")); }
  int all_gather_u64(const unsigned long long* d_send, unsigned long long* d_recv, size_t n, hipStream_t s) override {
    return chk(R->AllGather(d_send, d_recv, n, ncclUint64, comm, s), "ncclAllGather");
  }
  int all_reduce_sum_u64(unsigned long long* d_buf, size_t n, hipStream_t s) override {
    return chk(R->AllReduce(d_buf, d_buf, n, ncclUint64, ncclSum, comm, s), "ncclAllReduce");
  }
  int group_start() override { return chk(R->GroupStart(), "ncclGroupStart"); }
  int send(const void* d, size_t bytes, int peer, hipStream_t s) override { return chk(R->Send(d, bytes, ncclUint8, peer, comm, s), "ncclSend"); }
  int recv(void* d, size_t bytes, int peer, hipStream_t s) override { return chk(R->Recv(d, bytes, ncclUint8, peer, comm, s), "ncclRecv"); }
  int group_end(hipStream_t) override { return chk(R->GroupEnd(), "ncclGroupEnd"); }
  const char* name() const override { return "rccl"; }
};

// ---- loopback: the ranks are threads of this process, on one device --------------------------------------------------
// A send posts (source pointer, an event recorded on the sender's stream); the matching recv -- same (source, destination)
// pair, FIFO -- makes the receiver's stream wait for that event, enqueues a device-to-device copy and records an event the
// sender's stream then waits on, so the source stays untouched until the copy has run: the ordering ncclSend / ncclRecv give.
// Within a group the sends are posted first and the receives matched after, so no two ranks wait on each other.
struct LoopWorld {
  int world = 0;
  std::mutex mu;
  std::condition_variable cv;
  struct Msg { const void* src; size_t bytes; hipEvent_t ready; hipEvent_t done; bool taken; int rc; };
  std::map<std::pair<int, int>, std::deque<std::shared_ptr<Msg>>> box;
  // barrier + exchange board for the small collectives
  int arrived = 0;
  unsigned long long gen = 0;
  std::vector<std::vector<unsigned long long>> board;
  int joined = 0;
};
std::mutex g_loop_mu;
std::vector<std::shared_ptr<LoopWorld>> g_loop_worlds;
const char kLoopMagic[8] = {'G', 'S', 'D', 'F', 'L', 'O', 'O', 'P'};

struct LoopTransport final : Transport {
  std::shared_ptr<LoopWorld> w;
  int rank = 0;
  struct PendingRecv { void* dst; size_t bytes; int peer; };
  std::vector<std::shared_ptr<LoopWorld::Msg>> sent;
  std::vector<PendingRecv> recvs;
  // every rank deposits `mine` on the board; returns once all have, with a copy of the whole board
  std::vector<std::vector<unsigned long long>> exchange(const std::vector<unsigned long long>& mine) {
    std::unique_lock<std::mutex> lk(w->mu);
    const unsigned long long g = w->gen;
    if (w->board.size() != (size_t)w->world) w->board.assign((size_t)w->world, {});
    w->board[(size_t)rank] = mine;
    if (++w->arrived == w->world) {
      w->arrived = 0;
      w->gen++;
      w->cv.notify_all();
    } else {
      w->cv.wait(lk, [&] { return w->gen != g; });
    }
    // (the board is rewritten only after every rank has passed the NEXT exchange's entry, i.e. taken its copy: each rank
    // enters the next exchange only after returning from this one)
    auto out = w->board;
    lk.unlock();
    barrier_exit();
    return out;
  }
  void barrier_exit() {  // second phase: nobody rewrites the board before everybody has copied it
    std::unique_lock<std::mutex> lk(w->mu);
    const unsigned long long g = w->gen;
    if (++w->arrived == w->world) { w->arrived = 0; w->gen++; w->cv.notify_all(); }
    else w->cv.wait(lk, [&] { return w->gen != g; });
  }
  int all_gather_u64(const unsigned long long* d_send, unsigned long long* d_recv, size_t n, hipStream_t s) override {
    std::vector<unsigned long long> mine(n);
    HIP_TRY(hipMemcpyAsync(mine.data(), d_send, n * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const auto all = exchange(mine);
    std::vector<unsigned long long> flat;
    for (const auto& v : all) flat.insert(flat.end(), v.begin(), v.end());
    HIP_TRY(hipMemcpyAsync(d_recv, flat.data(), flat.size() * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));  // (the source is this stack frame)
    return GSDF_OK;
  }
  int all_reduce_sum_u64(unsigned long long* d_buf, size_t n, hipStream_t s) override {
    std::vector<unsigned long long> mine(n);
    HIP_TRY(hipMemcpyAsync(mine.data(), d_buf, n * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const auto all = exchange(mine);
    std::vector<unsigned long long> sum(n, 0);
    for (const auto& v : all) for (size_t i = 0; i < n && i < v.size(); i++) sum[i] += v[i];
    HIP_TRY(hipMemcpyAsync(d_buf, sum.data(), n * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    return GSDF_OK;
  }
  int group_start() override { sent.clear(); recvs.clear(); return GSDF_OK; }
  int send(const void* d, size_t bytes, int peer, hipStream_t s) override {
    auto m = std::make_shared<LoopWorld::Msg>();
    m->src = d; m->bytes = bytes; m->taken = false; m->rc = GSDF_OK; m->ready = nullptr; m->done = nullptr;
    HIP_TRY(hipEventCreateWithFlags(&m->ready, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&m->done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(m->ready, s));
    {
      std::lock_guard<std::mutex> lk(w->mu);
      w->box[{rank, peer}].push_back(m);
    }
    w->cv.notify_all();
    sent.push_back(m);
    return GSDF_OK;
  }
  int recv(void* d, size_t bytes, int peer, hipStream_t) override { recvs.push_back({d, bytes, peer}); return GSDF_OK; }
  int group_end(hipStream_t s) override {
    int rc = GSDF_OK;
    for (const PendingRecv& r : recvs) {
      std::shared_ptr<LoopWorld::Msg> m;
      {
        std::unique_lock<std::mutex> lk(w->mu);
        auto& q = w->box[{r.peer, rank}];
        w->cv.wait(lk, [&] { return !q.empty(); });
        m = q.front();
        q.pop_front();
      }
      int mrc = GSDF_OK;
      if (m->bytes != r.bytes) mrc = fail(GSDF_ERR_BAD_ARGUMENT, "loopback: a receive's size differs from the matching send's");
      else if (hipStreamWaitEvent(s, m->ready, 0) != hipSuccess || hipMemcpyAsync(r.dst, m->src, r.bytes, hipMemcpyDeviceToDevice, s) != hipSuccess ||
               hipEventRecord(m->done, s) != hipSuccess) mrc = fail(GSDF_ERR_HIP, "loopback: device-to-device transfer failed");
      {
        std::lock_guard<std::mutex> lk(w->mu);
        m->taken = true;
        m->rc = mrc;
      }
      w->cv.notify_all();
      if (mrc && !rc) rc = mrc;
    }
    for (auto& m : sent) {  // the source buffers stay untouched until the receivers' copies have run
      {
        std::unique_lock<std::mutex> lk(w->mu);
        w->cv.wait(lk, [&] { return m->taken; });
      }
      if (m->rc == GSDF_OK && hipStreamWaitEvent(s, m->done, 0) != hipSuccess && !rc) rc = fail(GSDF_ERR_HIP, "loopback: hipStreamWaitEvent failed");
      if (m->rc && !rc) rc = fail(GSDF_ERR_HIP, "loopback: the receiving rank failed");
      // (the events are destroyed with the last reference; hipEventDestroy of an event with pending work is deferred by the runtime)
      (void)hipEventDestroy(m->ready);
      (void)hipEventDestroy(m->done);
    }
    sent.clear();
    recvs.clear();
    return rc;
  }
  const char* name() const override { return "loopback"; }
};

bool loopback_requested() {
  const char* e = getenv("GSDF_HIP_COMM");
  return e && !strcmp(e, "loopback");
}

// ---- ipc: the ranks are PROCESSES of one node that may share a device ------------------------------------------------------------
// RCCL refuses two ranks on one GPU ("duplicate GPU"), so on a one-GPU box the N > 1 path of a multi-process caller -- bench.py under
// torch.distributed.run: a rendezvous of the 128-byte id, one HIP context per process, barriers, the counts, the plan, marching what
// arrived -- could not run at all. GSDF_HIP_COMM=ipc gives it a transport: a POSIX shared-memory segment named by the id holds a
// barrier, a board for the small collectives and one mailbox per (sender, receiver); a send posts the hipIpcMemHandle of its source
// allocation (+ offset, size) once the sender's stream has produced the data, the matching receive maps it (hipIpcOpenMemHandle,
// cached), copies device to device on its own stream and marks the mailbox taken, and the sender's group_end returns when its
// mailboxes are: ncclSend / ncclRecv ordering with the host in the loop (no interprocess events: a transfer has completed when
// group_end returns, so a _start overlaps nothing -- a test transport for what one GPU can show, not a fast one). Works between
// devices of one node as well. Every wait gives up after GSDF_HIP_IPC_TIMEOUT_S (default 120) seconds with an error.
struct IpcShm {
  static constexpr int kMaxWorld = 64;
  std::atomic<uint32_t> magic;      // set last by the creator
  std::atomic<uint32_t> world;
  std::atomic<uint32_t> bar_count;  // sense-reversing barrier
  std::atomic<uint32_t> bar_gen;
  std::atomic<uint32_t> failed;     // a rank gave up: everybody stops waiting
  uint32_t pad[11];
  unsigned long long board[kMaxWorld][16];
  struct Box {
    std::atomic<uint32_t> state;    // 0 empty, 1 posted, 2 taken
    int32_t rc;
    uint64_t off, bytes;
    hipIpcMemHandle_t h;
  } box[kMaxWorld][kMaxWorld];      // [sender][receiver]
};
const char kIpcMagic[8] = {'G', 'S', 'D', 'F', 'I', 'P', 'C', '1'};
bool ipc_requested() {
  const char* e = getenv("GSDF_HIP_COMM");
  return e && !strcmp(e, "ipc");
}
double ipc_timeout_s() {
  const char* e = getenv("GSDF_HIP_IPC_TIMEOUT_S");
  const double v = e ? atof(e) : 0.0;
  return v > 0 ? v : 120.0;
}

struct IpcTransport final : Transport {
  IpcShm* shm = nullptr;
  std::string shm_name;
  bool creator = false;
  int rank = 0, world = 0;
  struct PendingRecv { void* dst; size_t bytes; int peer; };
  std::vector<int> sent_to;
  std::vector<PendingRecv> recvs;
  IpcTransport() { pool_exporter_opened(); }
  ~IpcTransport() override {
    pool_exporter_closed();
    for (Mapped& m : mapped) (void)hipIpcCloseMemHandle(m.p);
    if (shm) munmap(shm, sizeof(IpcShm));
    if (creator) shm_unlink(shm_name.c_str());
  }
  template <typename F>
  int wait_until(F done, const char* what) {
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (!done()) {
      if (shm->failed.load(std::memory_order_acquire)) return fail(GSDF_ERR_HIP, std::string("ipc: another rank gave up while this one waited for ") + what);
      if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
      if ((spins & 1023u) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > ipc_timeout_s()) {
        shm->failed.store(1, std::memory_order_release);
        return fail(GSDF_ERR_HIP, std::string("ipc: timed out waiting for ") + what);
      }
    }
    return GSDF_OK;
  }
  int barrier() {
    const uint32_t gen = shm->bar_gen.load(std::memory_order_acquire);
    if (shm->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)world) {
      shm->bar_count.store(0, std::memory_order_relaxed);
      shm->bar_gen.fetch_add(1, std::memory_order_acq_rel);
      return GSDF_OK;
    }
    return wait_until([&] { return shm->bar_gen.load(std::memory_order_acquire) != gen; }, "the other ranks at a barrier");
  }
  int exchange(const std::vector<unsigned long long>& mine, std::vector<std::vector<unsigned long long>>& all) {
    if (mine.size() > 16) return fail(GSDF_ERR_BAD_ARGUMENT, "ipc: a small collective carries at most 16 words per rank");
    for (size_t i = 0; i < mine.size(); i++) shm->board[rank][i] = mine[i];
    if (int rc = barrier()) return rc;
    all.assign((size_t)world, std::vector<unsigned long long>(mine.size()));
    for (int r = 0; r < world; r++) for (size_t i = 0; i < mine.size(); i++) all[(size_t)r][i] = shm->board[r][i];
    return barrier();  // nobody rewrites the board before everybody has read it
  }
  int all_gather_u64(const unsigned long long* d_send, unsigned long long* d_recv, size_t n, hipStream_t s) override {
    std::vector<unsigned long long> mine(n);
    HIP_TRY(hipMemcpyAsync(mine.data(), d_send, n * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    std::vector<std::vector<unsigned long long>> all;
    if (int rc = exchange(mine, all)) return rc;
    std::vector<unsigned long long> flat;
    for (const auto& v : all) flat.insert(flat.end(), v.begin(), v.end());
    HIP_TRY(hipMemcpyAsync(d_recv, flat.data(), flat.size() * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    return GSDF_OK;
  }
  int all_reduce_sum_u64(unsigned long long* d_buf, size_t n, hipStream_t s) override {
    std::vector<unsigned long long> mine(n);
    HIP_TRY(hipMemcpyAsync(mine.data(), d_buf, n * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    std::vector<unsigned long long> sum(n, 0);
    for (size_t o = 0; o < n; o += 16) {  // in pieces of the board's 16 words per rank
      const size_t k = n - o < 16 ? n - o : 16;
      std::vector<std::vector<unsigned long long>> all;
      if (int rc = exchange(std::vector<unsigned long long>(mine.begin() + (long)o, mine.begin() + (long)(o + k)), all)) return rc;
      for (const auto& v : all) for (size_t i = 0; i < k; i++) sum[o + i] += v[i];
    }
    HIP_TRY(hipMemcpyAsync(d_buf, sum.data(), n * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    return GSDF_OK;
  }
  int group_start() override { sent_to.clear(); recvs.clear(); return GSDF_OK; }
  int send(const void* d, size_t bytes, int peer, hipStream_t s) override {
    if (peer < 0 || peer >= world || peer == rank) return fail(GSDF_ERR_BAD_ARGUMENT, "ipc: bad peer");
    HIP_TRY(hipStreamSynchronize(s));  // the payload is complete before its handle leaves the process
    void* base = nullptr;
    size_t size = 0;
    HIP_TRY(hipMemGetAddressRange((hipDeviceptr_t*)&base, &size, (hipDeviceptr_t)d));
    IpcShm::Box& b = shm->box[rank][peer];
    if (int rc = wait_until([&] { return b.state.load(std::memory_order_acquire) == 0u; }, "a mailbox to empty")) return rc;
    hipIpcMemHandle_t h;
    HIP_TRY(hipIpcGetMemHandle(&h, base));
    pool_note_exported(base);  // the peer will cache its mapping: the pool keeps this allocation alive while a transport like this one exists
    b.h = h; b.off = (uint64_t)((const char*)d - (const char*)base); b.bytes = bytes; b.rc = GSDF_OK;
    b.state.store(1u, std::memory_order_release);
    sent_to.push_back(peer);
    return GSDF_OK;
  }
  int recv(void* d, size_t bytes, int peer, hipStream_t) override { recvs.push_back({d, bytes, peer}); return GSDF_OK; }
  // Mappings of peers' allocations, by handle bytes, at most 32 (oldest closed first), all closed with the communicator. (Round 6 tried
  // "map for one copy, unmap again" -- the advisor's point: a pool buffer the peer has returned stays pinned here, and a recycled
  // address with the same handle bytes would be read through the stale mapping. With it hipIpcGetMemHandle on the SENDING side
  // began to fail with "invalid argument" after a dozen gathers at npt-flange@1600 (profiles/r6g: tools/gpu_evidence.sh dist):
  // re-exporting an allocation whose last import was closed is not something this runtime does reliably. The cache stays; the
  // transport is for tests on one GPU and never a default. GSDF_HIP_IPC_CLOSE=1 selects the other behaviour for experiments.
  // The stale-mapping case was then seen for real: with the triangle pool at four idle buffers and eight circulating per rank (three
  // meshes in flight + a gather), every step freed a buffer, a recycled address reached a peer with the old handle bytes, and the copy
  // faulted ("Memory access fault by GPU"). The pool keeps sixteen now (abi_host.cpp: pool_max) and nothing is freed in a steady loop,
  // and it never frees an allocation whose handle a send of this transport has posted (pool_note_exported) while such a transport lives.)
  struct Mapped { hipIpcMemHandle_t h; void* p; };
  std::vector<Mapped> mapped;
  static bool close_after_copy() { static const bool v = [] { const char* e = getenv("GSDF_HIP_IPC_CLOSE"); return e && atoi(e) != 0; }(); return v; }
  void* map(const hipIpcMemHandle_t& h) {
    if (!close_after_copy()) {
      for (const Mapped& m : mapped) if (!std::memcmp(&m.h, &h, sizeof h)) return m.p;
      if (mapped.size() >= 32) { (void)hipIpcCloseMemHandle(mapped.front().p); mapped.erase(mapped.begin()); }  // (pool buffers come and go)
    }
    void* p = nullptr;
    if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (!close_after_copy()) mapped.push_back({h, p});
    return p;
  }
  int group_end(hipStream_t s) override {
    int rc = GSDF_OK;
    for (const PendingRecv& r : recvs) {
      IpcShm::Box& b = shm->box[r.peer][rank];
      int mrc = wait_until([&] { return b.state.load(std::memory_order_acquire) == 1u; }, "a peer's send");
      if (mrc) { if (!rc) rc = mrc; break; }
      if (b.bytes != r.bytes) mrc = fail(GSDF_ERR_BAD_ARGUMENT, "ipc: a receive's size differs from the matching send's");
      else {
        void* src = map(b.h);
        if (!src) mrc = fail(GSDF_ERR_HIP, "ipc: hipIpcOpenMemHandle failed (is HSA_ENABLE_IPC_MODE_LEGACY=0 set?)");
        else {
          if (hipMemcpyAsync(r.dst, (const char*)src + b.off, r.bytes, hipMemcpyDeviceToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
            mrc = fail(GSDF_ERR_HIP, "ipc: device-to-device copy from a peer's allocation failed");
          if (close_after_copy()) (void)hipIpcCloseMemHandle(src);
        }
      }
      b.rc = mrc;
      b.state.store(2u, std::memory_order_release);
      if (mrc && !rc) rc = mrc;
    }
    for (int peer : sent_to) {  // the source stays untouched until the receiver has copied it
      IpcShm::Box& b = shm->box[rank][peer];
      const int wrc = wait_until([&] { return b.state.load(std::memory_order_acquire) == 2u; }, "a peer to take a send");
      if (wrc) { if (!rc) rc = wrc; continue; }
      if (b.rc && !rc) rc = fail(GSDF_ERR_HIP, "ipc: the receiving rank failed");
      b.state.store(0u, std::memory_order_release);
    }
    sent_to.clear();
    recvs.clear();
    return rc;
  }
  const char* name() const override { return "ipc"; }
  // rank 0 creates the segment, the others wait for it
  int attach(const uint8_t* id, int rank_, int world_) {
    rank = rank_; world = world_;
    uint64_t token = 0;
    std::memcpy(&token, id + 8, 8);
    char nm[64];
    snprintf(nm, sizeof nm, "/gsdf_ipc_%016llx", (unsigned long long)token);
    shm_name = nm;
    int fd = -1;
    if (rank == 0) {
      (void)shm_unlink(nm);
      fd = shm_open(nm, O_CREAT | O_EXCL | O_RDWR, 0600);
      if (fd < 0 || ftruncate(fd, (off_t)sizeof(IpcShm)) != 0) { if (fd >= 0) close(fd); return fail(GSDF_ERR_HIP, std::string("ipc: cannot create the shared-memory segment ") + nm); }
      creator = true;
    } else {
      const auto t0 = std::chrono::steady_clock::now();
      for (;;) {
        fd = shm_open(nm, O_RDWR, 0600);
        struct stat st;
        if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= sizeof(IpcShm)) break;
        if (fd >= 0) { close(fd); fd = -1; }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > ipc_timeout_s()) return fail(GSDF_ERR_HIP, std::string("ipc: rank 0 never created ") + nm);
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
      }
    }
    void* p = mmap(nullptr, sizeof(IpcShm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return fail(GSDF_ERR_HIP, "ipc: mmap of the shared-memory segment failed");
    shm = (IpcShm*)p;  // (a fresh segment is zero-filled: counters 0, mailboxes empty)
    if (rank == 0) { shm->world.store((uint32_t)world); shm->magic.store(0x43504947u, std::memory_order_release); }
    else {
      if (int rc = wait_until([&] { return shm->magic.load(std::memory_order_acquire) == 0x43504947u; }, "rank 0 to publish the segment")) return rc;
      if (shm->world.load() != (uint32_t)world) return fail(GSDF_ERR_BAD_ARGUMENT, "ipc: ranks disagree about the world size");
    }
    const int brc = barrier();  // everybody is attached: the name can go now -- the segment lives on in the mappings, and a rank 0
    if (creator) { shm_unlink(shm_name.c_str()); creator = false; }  // that dies later leaves nothing behind in /dev/shm
    return brc;
  }
};
}  // namespace

struct gsdf_comm {
  std::unique_ptr<Transport> t;
  int rank = 0, world = 1, device = 0, num_cu = 256;
  hipStream_t stream = nullptr;
  static constexpr int kCountWords = 4;    // per rank: payload kind, triangles, records, payload bytes
  unsigned long long* d_counts = nullptr;  // [world * kCountWords] gathered, then this rank's own kCountWords
  unsigned long long* h_counts = nullptr;  // pinned mirror
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};  // around the counts exchange and the payload, on `stream`
};

// A gather whose payload is on its way (gsdf_hip_mesh_gatherv_start): the result mesh, the counts, and the events that time it.
struct gsdf_gather {
  gsdf_comm* c = nullptr;
  gsdf_mesh* src = nullptr;  // the mesh being gathered: its buffers are read until the payload has moved (inflight)
  gsdf_mesh* g = nullptr;    // result (NULL on the ranks that receive nothing)
  uint8_t* d_recv = nullptr; // records payload: the receive buffer (triangle pool), released at wait
  uint64_t recv_cap36 = 0;
  std::vector<uint64_t> counts;  // triangles per rank
  gsdf_gather_stats st{};
  hipEvent_t ev_payload0 = nullptr, ev_payload1 = nullptr, ev_march1 = nullptr;
  float ms_counts = 0;
};

extern "C" int gsdf_hip_comm_unique_id(uint8_t id[GSDF_COMM_ID_BYTES]) {
  if (!id) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  std::memset(id, 0, GSDF_COMM_ID_BYTES);
  if (loopback_requested()) {
    std::lock_guard<std::mutex> lk(g_loop_mu);
    g_loop_worlds.push_back(std::make_shared<LoopWorld>());
    const uint64_t idx = g_loop_worlds.size() - 1;
    std::memcpy(id, kLoopMagic, 8);
    std::memcpy(id + 8, &idx, 8);
    return GSDF_OK;
  }
  if (ipc_requested()) {  // a random token names the shared-memory segment
    uint64_t token = 0;
    int fd = open("/dev/urandom", O_RDONLY);
    if (fd >= 0) { if (read(fd, &token, 8) != 8) token = 0; close(fd); }
    if (!token) token = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() ^ ((uint64_t)getpid() << 32);
    std::memcpy(id, kIpcMagic, 8);
    std::memcpy(id + 8, &token, 8);
    return GSDF_OK;
  }
  RcclApi* R = rccl();
  if (!R->err.empty()) return fail(GSDF_ERR_HIP, R->err);
  static_assert(sizeof(ncclUniqueId) <= GSDF_COMM_ID_BYTES, "ncclUniqueId larger than GSDF_COMM_ID_BYTES");
  ncclUniqueId u;
  ncclResult_t r = R->GetUniqueId(&u);
  if (r != ncclSuccess) return fail(GSDF_ERR_HIP, std::string("ncclGetUniqueId: ") + R->GetErrorString(r));
  std::memcpy(id, &u, sizeof u);
  return GSDF_OK;
}

extern "C" void gsdf_hip_comm_destroy(gsdf_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  c->t.reset();
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
  if (world > kDenseMaxParts) return fail(GSDF_ERR_BAD_ARGUMENT, "world size above 64");
  const bool loop = std::memcmp(id, kLoopMagic, 8) == 0, ipc = std::memcmp(id, kIpcMagic, 8) == 0;
  RcclApi* R = (loop || ipc) ? nullptr : rccl();
  if (R && !R->err.empty()) return fail(GSDF_ERR_HIP, R->err);
  gsdf_comm* c = new (std::nothrow) gsdf_comm();
  if (!c) return fail(GSDF_ERR_BAD_ARGUMENT, "out of memory");
  c->rank = rank; c->world = world;
  auto bail = [&](int code) { gsdf_hip_comm_destroy(c); return code; };
  if (hipGetDevice(&c->device) != hipSuccess) return bail(fail(GSDF_ERR_HIP, "hipGetDevice failed"));
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, c->device) == hipSuccess) c->num_cu = prop.multiProcessorCount;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(GSDF_ERR_HIP, "hipStreamCreate failed"));
  const size_t words = (size_t)(world + 1) * gsdf_comm::kCountWords;
  if (hipMalloc((void**)&c->d_counts, sizeof(unsigned long long) * words) != hipSuccess) return bail(fail(GSDF_ERR_HIP, "hipMalloc(counts) failed"));
  if (hipHostMalloc((void**)&c->h_counts, sizeof(unsigned long long) * words, hipHostMallocDefault) != hipSuccess) return bail(fail(GSDF_ERR_HIP, "hipHostMalloc(counts) failed"));
  if (loop) {
    uint64_t idx = 0;
    std::memcpy(&idx, id + 8, 8);
    std::shared_ptr<LoopWorld> w;
    {
      std::lock_guard<std::mutex> lk(g_loop_mu);
      if (idx >= g_loop_worlds.size()) return bail(fail(GSDF_ERR_BAD_ARGUMENT, "loopback: unknown communicator id (ranks must be threads of one process)"));
      w = g_loop_worlds[(size_t)idx];
    }
    {
      std::lock_guard<std::mutex> lk(w->mu);
      if (w->world == 0) w->world = world;
      if (w->world != world) return bail(fail(GSDF_ERR_BAD_ARGUMENT, "loopback: ranks disagree about the world size"));
      w->joined++;
    }
    auto t = std::make_unique<LoopTransport>();
    t->w = w; t->rank = rank;
    c->t = std::move(t);
  } else if (ipc) {
    auto t = std::make_unique<IpcTransport>();
    if (int rc = t->attach(id, rank, world)) return bail(rc);
    c->t = std::move(t);
  } else {
    auto t = std::make_unique<RcclTransport>(R);
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    ncclResult_t r = R->CommInitRank(&t->comm, world, u, rank);
    if (r != ncclSuccess) { t->comm = nullptr; return bail(fail(GSDF_ERR_HIP, std::string("ncclCommInitRank: ") + R->GetErrorString(r))); }
    c->t = std::move(t);
  }
  *out = c;
  return GSDF_OK;
}
extern "C" int gsdf_hip_comm_rank(const gsdf_comm* c) { return c ? c->rank : -1; }
extern "C" int gsdf_hip_comm_world(const gsdf_comm* c) { return c ? c->world : 0; }
extern "C" const char* gsdf_hip_comm_transport(const gsdf_comm* c) { return c && c->t ? c->t->name() : ""; }

// Sum of `n` host uint64 values over all ranks, in place (Evaluations(), TotalPruned(), triangle totals).
extern "C" int gsdf_hip_comm_allreduce_sum_u64(gsdf_comm* c, uint64_t* vals, size_t n) {
  if (!c || !vals) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  if (n == 0) return GSDF_OK;
  HIP_TRY(hipSetDevice(c->device));
  unsigned long long* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, n * 8));
  int rc = GSDF_OK;
  do {
    if (hipMemcpyAsync(d, vals, n * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess) { rc = fail(GSDF_ERR_HIP, "H2D copy failed"); break; }
    if ((rc = c->t->all_reduce_sum_u64(d, n, c->stream)) != GSDF_OK) break;
    if (hipMemcpyAsync(vals, d, n * 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { rc = fail(GSDF_ERR_HIP, "D2H copy failed"); break; }
  } while (0);
  (void)hipFree(d);
  return rc;
}

// Gather of the ranks' meshes. mode ALL: every rank ends up with the triangles of all ranks in rank order; ROOT: only `root` does
// (a rank's link then carries its own payload only, 1/world of what ALL puts on it); NONE: the counts only -- every rank keeps its
// shard (a caller that writes per-rank files, or consumes the shards where they are). What moves is the mesh's payload: triangles
// (36 B each), or packed cut-leaf records (gsdf_mesh_opts.payload: 40 B per cut leaf = 20 B per triangle), which the receiving ranks
// march into triangles behind the transfer, on the communicator's stream. _start returns once the counts are known and everything
// is enqueued: the caller may mesh the next part while it moves; _wait returns the result, a mesh like any other (read / host
// views / STL / destroy as usual), NULL on ranks that received nothing. The source mesh may be destroyed after _start: its buffers
// are kept until the payload has moved.
extern "C" int gsdf_hip_mesh_gatherv_start(const gsdf_mesh* m_in, gsdf_comm* c, int mode, int root, gsdf_gather** out) {
  gsdf_mesh* m = const_cast<gsdf_mesh*>(m_in);
  if (!m || !c || !out) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  *out = nullptr;
  if (mode != GSDF_GATHER_ALL && mode != GSDF_GATHER_ROOT && mode != GSDF_GATHER_NONE) return fail(GSDF_ERR_BAD_ARGUMENT, "bad gather mode");
  if (mode == GSDF_GATHER_ROOT && (root < 0 || root >= c->world)) return fail(GSDF_ERR_BAD_ARGUMENT, "bad root rank");
  if (m->host_out) return fail(GSDF_ERR_BAD_ARGUMENT, "gatherv needs device-resident triangles (host_output meshes live in host memory)");
  if (m->device != c->device) return fail(GSDF_ERR_BAD_ARGUMENT, "mesh and communicator are on different devices");
  HIP_TRY(hipSetDevice(c->device));
  for (auto& e : c->ev) if (!e) HIP_TRY(hipEventCreate(&e));
  const int W = c->world, K = gsdf_comm::kCountWords;
  const bool recs = m->payload == GSDF_PAYLOAD_RECORDS;
  // 1. counts
  HIP_TRY(hipEventRecord(c->ev[0], c->stream));
  unsigned long long* mine_h = c->h_counts + (size_t)W * K;
  mine_h[0] = (unsigned long long)m->payload;
  mine_h[1] = m->st.n_tris;
  mine_h[2] = recs ? m->n_recs : 0;
  mine_h[3] = recs ? dense_bytes(m->n_recs) : m->st.n_tris * 36ull;
  if (recs && m->n_recs == 0) mine_h[3] = 0;
  HIP_TRY(hipMemcpyAsync(c->d_counts + (size_t)W * K, mine_h, 8 * (size_t)K, hipMemcpyHostToDevice, c->stream));
  if (int rc = c->t->all_gather_u64(c->d_counts + (size_t)W * K, c->d_counts, (size_t)K, c->stream)) return rc;
  HIP_TRY(hipMemcpyAsync(c->h_counts, c->d_counts, 8 * (size_t)W * K, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipEventRecord(c->ev[1], c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  std::vector<uint64_t> bytes((size_t)W), ntri((size_t)W), nrec((size_t)W);
  for (int r = 0; r < W; r++) {
    const unsigned long long* h = c->h_counts + (size_t)r * K;
    if ((int)h[0] != m->payload) return fail(GSDF_ERR_BAD_ARGUMENT, "gatherv: the ranks' meshes hold different payloads (triangles on some, records on others)");
    ntri[(size_t)r] = h[1]; nrec[(size_t)r] = h[2]; bytes[(size_t)r] = h[3];
  }
  gsdf_gather* p = new (std::nothrow) gsdf_gather();
  if (!p) return fail(GSDF_ERR_BAD_ARGUMENT, "out of memory");
  p->c = c;
  p->counts = ntri;
  (void)hipEventElapsedTime(&p->ms_counts, c->ev[0], c->ev[1]);
  auto bail = [&](int code) {
    (void)hipStreamSynchronize(c->stream);  // transfers or a marching kernel may already be enqueued: nothing goes back to the pool under them
    (void)hipGetLastError();
    if (p->g) gsdf_hip_mesh_destroy(p->g);
    if (p->d_recv) pool_give(c->device, (float*)p->d_recv, p->recv_cap36);
    for (hipEvent_t e : {p->ev_payload0, p->ev_payload1, p->ev_march1}) if (e) (void)hipEventDestroy(e);
    if (p->src) mesh_inflight_done(p->src);
    delete p;
    return code;
  };
  // 2. the plan
  std::vector<gsdf_gather_op> ops((size_t)2 * W + 1);
  size_t n_ops = 0;
  uint64_t total_bytes = 0;
  if (int rc = gsdf_hip_gather_plan(bytes.data(), W, c->rank, mode, root, ops.data(), ops.size(), &n_ops, &total_bytes)) return bail(rc);
  uint64_t total_tris = 0;
  for (int r = 0; r < W; r++) total_tris += ntri[(size_t)r];
  const bool receives = mode == GSDF_GATHER_ALL || (mode == GSDF_GATHER_ROOT && c->rank == root);
  uint8_t* dst_base = nullptr;
  if (receives) {
    gsdf_mesh* g = new (std::nothrow) gsdf_mesh();
    if (!g) return bail(fail(GSDF_ERR_BAD_ARGUMENT, "out of memory"));
    p->g = g;
    g->device = c->device;
    g->num_cu = c->num_cu;
    g->st = m->st;  // resolution, origin, levels; per-rank counters stay per-rank (sum them with gsdf_hip_comm_allreduce_sum_u64)
    g->st.n_tris = total_tris;
    if (total_tris) {
      const uint64_t units = total_tris + (recs ? (dense_parts_bytes() + 35) / 36 : 0);  // records: the parts table rides behind the triangles
      g->d_tris = pool_take(c->device, units, &g->cap);
      if (!g->d_tris) {
        if (hipMalloc((void**)&g->d_tris, units * 36) != hipSuccess) { (void)hipGetLastError(); return bail(fail(GSDF_ERR_HIP, "hipMalloc of the gathered triangle buffer failed")); }
        g->cap = units;
      }
    }
    if (recs && total_bytes) {
      const uint64_t units = (total_bytes + 35) / 36;
      p->d_recv = (uint8_t*)pool_take(c->device, units, &p->recv_cap36);
      if (!p->d_recv) {
        if (hipMalloc((void**)&p->d_recv, units * 36) != hipSuccess) { (void)hipGetLastError(); return bail(fail(GSDF_ERR_HIP, "hipMalloc of the gathered record buffer failed")); }
        p->recv_cap36 = units;
      }
    }
    dst_base = recs ? p->d_recv : (uint8_t*)g->d_tris;
    p->st.bytes_received = total_bytes - bytes[(size_t)c->rank];
  }
  const uint8_t* src_base = recs ? m->d_recs : (const uint8_t*)m->d_tris;
  for (hipEvent_t* e : {&p->ev_payload0, &p->ev_payload1, &p->ev_march1})
    if (hipEventCreate(e) != hipSuccess) return bail(fail(GSDF_ERR_HIP, "hipEventCreate failed"));
  m->inflight.fetch_add(1);
  m->refs.fetch_add(1);
  p->src = m;
  if (hipEventRecord(p->ev_payload0, c->stream) != hipSuccess) return bail(fail(GSDF_ERR_HIP, "hipEventRecord failed"));
  // 3. the payload: one grouped launch of the plan
  if (n_ops) {
    if (int rc = c->t->group_start()) return bail(rc);
    int rb = GSDF_OK;
    for (size_t i = 0; i < n_ops && rb == GSDF_OK; i++) {
      const gsdf_gather_op& o = ops[i];
      if (o.kind == GSDF_GOP_COPY) {
        if (hipMemcpyAsync(dst_base + o.dst_off, src_base + o.src_off, o.bytes, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) rb = fail(GSDF_ERR_HIP, "device copy of this rank's own payload failed");
      } else if (o.kind == GSDF_GOP_SEND) {
        rb = c->t->send(src_base + o.src_off, o.bytes, o.peer, c->stream);
        p->st.bytes_sent += o.bytes;
      } else {
        rb = c->t->recv(dst_base + o.dst_off, o.bytes, o.peer, c->stream);
      }
    }
    const std::string first_err = rb ? std::string(gsdf_hip_last_error()) : std::string();
    const int r1 = c->t->group_end(c->stream);
    if (rb) return bail(fail(rb, "gather payload: " + first_err));
    if (r1) return bail(r1);
  }
  if (hipEventRecord(p->ev_payload1, c->stream) != hipSuccess) return bail(fail(GSDF_ERR_HIP, "hipEventRecord failed"));
  // 4. records: marching cubes over everybody's, behind the transfer on the same stream
  if (receives && recs && total_tris) {
    std::vector<gsdf_dense_part> parts;
    uint64_t off = 0, t0 = 0;
    for (int r = 0; r < W; r++) {
      if (nrec[(size_t)r]) parts.push_back(gsdf_dense_part{off, nrec[(size_t)r], t0});
      off += bytes[(size_t)r];
      t0 += ntri[(size_t)r];
    }
    if (int rc = mesh_march_dense(p->d_recv, parts.data(), (int)parts.size(), dense_parts_at(p->g->d_tris, total_tris), m->st.origin[0], m->st.origin[1], m->st.origin[2],
                                  m->st.res, p->g->d_tris, c->num_cu, c->stream)) return bail(rc);
  }
  if (hipEventRecord(p->ev_march1, c->stream) != hipSuccess) return bail(fail(GSDF_ERR_HIP, "hipEventRecord failed"));
  *out = p;
  return GSDF_OK;
}

extern "C" int gsdf_hip_mesh_gatherv_wait(gsdf_gather* p, gsdf_mesh** out, uint64_t* counts, gsdf_gather_stats* st) {
  if (!p) return fail(GSDF_ERR_BAD_ARGUMENT, "null argument");
  if (out) *out = nullptr;
  (void)hipSetDevice(p->c->device);
  hipError_t e = hipEventSynchronize(p->ev_march1);
  int rc = GSDF_OK;
  if (e != hipSuccess) rc = fail(GSDF_ERR_HIP, std::string("gatherv: ") + hipGetErrorString(e));
  float ms = 0, ms2 = 0;
  if (rc == GSDF_OK) { (void)hipEventElapsedTime(&ms, p->ev_payload0, p->ev_payload1); (void)hipEventElapsedTime(&ms2, p->ev_payload1, p->ev_march1); }
  p->st.ms_counts = p->ms_counts;
  p->st.ms_payload = ms;
  p->st.ms_march = ms2;
  if (counts) for (size_t r = 0; r < p->counts.size(); r++) counts[r] = p->counts[r];
  if (st) *st = p->st;
  if (rc == GSDF_OK && out) { *out = p->g; p->g = nullptr; }
  if (p->g) gsdf_hip_mesh_destroy(p->g);
  if (p->d_recv) pool_give(p->c->device, (float*)p->d_recv, p->recv_cap36);
  if (p->src) mesh_inflight_done(p->src);
  (void)hipEventDestroy(p->ev_payload0);
  (void)hipEventDestroy(p->ev_payload1);
  (void)hipEventDestroy(p->ev_march1);
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
