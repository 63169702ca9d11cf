// abi_host.h -- what the translation units of the C ABI share on the host side: error plumbing, the mesh handle, the buffer
// pools. No device code (abi_comm.cpp and abi_host.cpp are plain C++).
//   abi_host.cpp  error text, triangle / pinned-buffer pools, result copies
//   abi_eval.hip  programs, specialisation, gleval.SDF3 / SDF2 Evaluate, normals, image renderer, block cache
//   abi_mesh.hip  octree / flat / dual-contouring meshers and the mesh accessors
//   abi_comm.cpp  multi-GPU: communicator (RCCL or the in-process loopback transport), gather plan, gatherv
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <string>

#pragma GCC visibility push(default)
#include "../../include/gsdf_hip.h"
#pragma GCC visibility pop

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
int fail(int code, const std::string& msg);  // sets the calling thread's gsdf_hip_last_error text, returns code
#define HIP_TRY(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t _e = (expr);                                                                             \
    if (_e != hipSuccess) return fail(GSDF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

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
  // GSDF_PAYLOAD_RECORDS: the packed cut-leaf records (kernels_octree.h: dense_payload_bytes) instead of triangles, until
  // gsdf_hip_mesh_march / a gather turns them into triangles. d_recs comes from the triangle pool (capacity in 36-byte units).
  int payload = 0;
  uint8_t* d_recs = nullptr;
  uint64_t recs_cap36 = 0;
  uint64_t n_recs = 0;
  int num_cu = 256;
  // device time per stage, where a mesher records it (dual contouring: five stages; gsdf_hip_mesh_stage_ms)
  int n_stages = 0;
  double stage_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const char* stage_name[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  // A gather in flight reads this mesh's buffers on the communicator's stream (gsdf_hip_mesh_gatherv_start .. _wait): a
  // destroy in between is deferred to the gather's end instead of handing the buffers to the next mesh under it.
  // ONE count of references: the owner's (dropped by gsdf_hip_mesh_destroy) + one per gather in flight (dropped by
  // mesh_inflight_done); whoever drops the last one frees the mesh -- destroy and a gather's end may run on different threads, and
  // two separate flags let both of them free it.
  std::atomic<int> refs{1};
  std::atomic<int> inflight{0};  // gathers in flight (what gsdf_hip_mesh_march asks about); the lifetime is decided by refs
};
void mesh_inflight_done(gsdf_mesh* m);  // abi_mesh.hip: one gather fewer; destroys a mesh whose owner already let go

// Packed records of several ranks side by side in one device buffer (abi_mesh.hip; layout: kernels_octree.h DensePart).
struct gsdf_dense_part { uint64_t off, n_recs, tri0; };
constexpr int kDenseMaxParts = 64;
inline uint64_t dense_bytes(uint64_t n_recs) { return n_recs * 40ull + ((((n_recs + 255ull) / 256ull) * 4ull + 7ull) & ~7ull); }
// Marching cubes over them into d_tris (room for the parts' triangles: the last part's tri0 + its own), on stream s.
// d_parts: device scratch of at least dense_parts_bytes() that stays valid until the work on s has run.
size_t dense_parts_bytes();
void* dense_parts_at(float* d_tris, uint64_t n_tris);  // where the table goes behind n_tris triangles (8-byte aligned)
int mesh_march_dense(const uint8_t* d_buf, const gsdf_dense_part* parts, int nparts, void* d_parts, float ox, float oy, float oz, float res,
                     float* d_tris, int num_cu, hipStream_t s);

// Triangle buffers and pinned host buffers are recycled through small per-process pools (abi_host.cpp).
float* pool_take(int device, uint64_t need, uint64_t* cap_out);
void pool_give(int device, float* p, uint64_t cap);
// the ipc test transport: a transport of that kind exists / is gone; the allocation at `base` has been exported to a peer process
void pool_exporter_opened();
void pool_exporter_closed();
void pool_note_exported(void* base);
void* hpool_take(size_t need, size_t* cap_out);
void hpool_give(void* p, size_t cap);
void big_memcpy(void* dst, const void* src, size_t n);
int host_buf(void** buf, size_t* cap, size_t need);
hipStream_t mesh_stream(gsdf_mesh* m);
void release_tris(gsdf_mesh* m);
