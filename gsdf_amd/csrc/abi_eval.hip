// abi_eval.hip -- C ABI (include/gsdf_hip.h), evaluator side: program handles and their run-time specialisation, the
// gleval.SDF3 / SDF2 Evaluate drop-ins (host buffers, pipelined tickets, registered memory, device-resident), central-difference
// normals, the 2-D image renderer, the block cache. Kernels: kernels_eval.h over the interpreter of interp.h.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <unordered_map>

#include "kernels_common.h"
#include "kernels_eval.h"
#include "abi_program.h"

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

// Test hook: dm::atan2_fast (float32 of math.Atan2 without the reference's float64 operation sequence, accepted only where the
// rounding is decided) against dm::atan2_ref, over 2^log2n pairs of `mode` (kernels_eval.h: atan2_selftest_kernel).
extern "C" int gsdf_hip_selftest_atan2(int mode, int log2n, uint64_t* mismatches, uint64_t* fast_path_points) {
  if (mismatches) *mismatches = 0;
  if (fast_path_points) *fast_path_points = 0;
  if (mode < 0 || mode > 2 || log2n < 0 || log2n > 40) return fail(GSDF_ERR_BAD_ARGUMENT, "mode in 0..2, log2n in 0..40");
  unsigned long long* d_c = nullptr;
  HIP_TRY(hipMalloc((void**)&d_c, 16));
  int rc = GSDF_OK;
  do {
    if (hipMemset(d_c, 0, 16) != hipSuccess) { rc = fail(GSDF_ERR_HIP, "memset failed"); break; }
    hipLaunchKernelGGL(atan2_selftest_kernel, dim3(4096), dim3(BLOCK), 0, nullptr, mode, log2n, d_c, d_c + 1);
    unsigned long long h[2] = {0, 0};
    if (hipMemcpy(h, d_c, 16, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(GSDF_ERR_HIP, "selftest kernel failed"); break; }
    if (mismatches) *mismatches = h[0];
    if (fast_path_points) *fast_path_points = h[1];
  } while (0);
  (void)hipFree(d_c);
  return rc;
}

// Test hook: dm::cossin_fast (float32 of math.Cos / math.Sin by FMAs, accepted only where the rounding is decided) against dm::cossinf_,
// over every float32 argument.
extern "C" int gsdf_hip_selftest_cossin(uint64_t* mismatches, uint64_t* fast_path_arguments) {
  if (mismatches) *mismatches = 0;
  if (fast_path_arguments) *fast_path_arguments = 0;
  unsigned long long* d_c = nullptr;
  HIP_TRY(hipMalloc((void**)&d_c, 16));
  int rc = GSDF_OK;
  do {
    if (hipMemset(d_c, 0, 16) != hipSuccess) { rc = fail(GSDF_ERR_HIP, "memset failed"); break; }
    hipLaunchKernelGGL(cossin_selftest_kernel, dim3(4096), dim3(BLOCK), 0, nullptr, d_c, d_c + 1);
    unsigned long long h[2] = {0, 0};
    if (hipMemcpy(h, d_c, 16, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(GSDF_ERR_HIP, "selftest kernel failed"); break; }
    if (mismatches) *mismatches = h[0];
    if (fast_path_arguments) *fast_path_arguments = h[1];
  } while (0);
  (void)hipFree(d_c);
  return rc;
}

// Test hook: dm::sqrt_1to2 against sqrtf for all 8,388,609 floats in [1, 2]; dm::sqrt_core against sqrtf for every float >= 2^-96.
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
void spec_aux(gsdf_program* p) {
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
    if (spec_build(p, {"dc_origin_kernel<" + kw + ">", "dc_edges_kernel", "dc_normals_kernel", "normals_kernel", "flat_grid_kernel<" + kw + ">", "dc_block_test_kernel"},
                   &p->spec_mod2, f, &p->spec_compile_s) == GSDF_OK) {
      const char* nm[6] = {"dc_origin_kernel", "dc_edges_kernel", "dc_normals_kernel", "normals_kernel", "flat_grid_kernel", "dc_block_test_kernel"};
      hipFunction_t* dst[6] = {&p->f_dc_origin, &p->f_dc_edges, &p->f_dc_normals, &p->f_normals, &p->f_flat_grid, &p->f_dc_block_test};
      for (int i = 0; i < 6; i++) {
        const bool ok = fn_scratch_bytes(f[(size_t)i]) == 0;
        spec_report("specialised", nm[i], f[(size_t)i], ok);
        *dst[i] = ok ? f[(size_t)i] : nullptr;
      }
    }
  }
}

// leaf_eval_kernel with distinct z rows (kernels_octree.h: DZ) for a specialised handle whose leaf phase runs four points per
// lane: the one-body form at the occupancy the handle's evaluating kernel has, then at four workgroups per CU, then the two-site
// form; the first that builds without scratch is taken. If none does (knurled-cylinder with hipcc 7.2: three variants of the
// second pass in one body want more than 128 registers; at three workgroups per CU the option costs more than it saves), a
// specialised handle keeps evaluating every row with its specialised kernel.
void spec_leaf_dz(gsdf_program* p) {
  if (!p->spec_mod || p->spec_dz_tried || p->prog.is2d) return;
  p->spec_dz_tried = true;
  int lk, lw;
  size_t lds_m;
  p->leaf_config(&lk, &lw, &lds_m);
  if (fused_leaf() || lk != 4 || !p->f_leaf || p->spec_leaf_k != 4) return;
  const std::string nt = p->leaf_nt_in_lds() ? "true" : "false";
  std::vector<std::string> names;
  std::vector<int> ws;
  std::vector<bool> both;
  auto add = [&](int w, bool b) {
    for (size_t i = 0; i < ws.size(); i++) if (ws[i] == w && both[i] == b) return;
    names.push_back("leaf_eval_kernel<4, " + std::to_string(w) + ", true, " + nt + (b ? ", true, true>" : ", false, true>"));
    ws.push_back(w); both.push_back(b);
  };
  if (p->spec_leaf_both) { add(p->spec_leaf_w, true); add(lw, true); }
  add(p->spec_leaf_w, false); add(lw, false);
  std::vector<hipFunction_t> f;
  if (spec_build(p, names, &p->spec_mod_dz, f, &p->spec_compile_s) != GSDF_OK) return;
  for (size_t i = 0; i < names.size(); i++) {
    const bool ok = fn_scratch_bytes(f[i]) == 0;
    spec_report("specialised", names[i], f[i], ok && !p->f_leaf_dz);
    if (ok && !p->f_leaf_dz) { p->f_leaf_dz = f[i]; p->spec_leaf_dz_w = ws[i]; p->spec_leaf_dz_both = both[i]; }
  }
}

// eval_kernel<D, 1, 4> for a specialised handle: one point per lane, what host-mapped Evaluate calls run (eval_dev). Built at the
// first such call, in a module of its own (the first group's build key -- the handle's code identity -- stays what it is).
void spec_eval_k1(gsdf_program* p) {
  static std::mutex mu;  // host-buffer calls may come from several threads
  std::lock_guard<std::mutex> lk(mu);
  if (!p->spec_mod || p->spec_k1_tried || p->batch_k() == 1) return;
  p->spec_k1_tried = true;
  std::vector<hipFunction_t> f;
  const std::string name = std::string("eval_kernel<") + (p->prog.is2d ? "2" : "3") + ", 1, 4>";
  if (spec_build(p, {name}, &p->spec_mod_k1, f, &p->spec_compile_s) != GSDF_OK) return;
  const bool ok = fn_scratch_bytes(f[0]) == 0;
  spec_report("specialised", name, f[0], ok);
  if (ok) p->f_eval_k1 = f[0];
}

// leaf_dense_kernel (share_corners = 1: every bitwise-distinct lattice point of a brick once) for a specialised handle, at the most
// workgroups per CU its LDS allows for which the compiler needs no scratch. Failure leaves the interpreter's kernel in use.
void spec_leaf_dense(gsdf_program* p) {
  if (!p->spec_mod || p->spec_dense_tried || p->prog.is2d || fused_leaf()) return;
  p->spec_dense_tried = true;
  std::vector<std::string> names;
  std::vector<int> ws;
  for (int w = 5; w >= 3; w--)
    if ((size_t)w * p->lds_dense() <= (size_t)160 * 1024) { names.push_back("leaf_dense_kernel<" + std::to_string(w) + ", true, true>"); ws.push_back(w); }
  if (names.empty()) return;
  std::vector<hipFunction_t> f;
  if (spec_build(p, names, &p->spec_mod_dense, f, &p->spec_compile_s) != GSDF_OK) return;
  for (size_t i = 0; i < names.size(); i++) {
    const bool ok = fn_scratch_bytes(f[i]) == 0;
    spec_report("specialised", names[i], f[i], ok && !p->f_leaf_dense);
    if (ok && !p->f_leaf_dense) { p->f_leaf_dense = f[i]; p->spec_leaf_dense_w = ws[i]; }
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
  int both_at = -1, both5_at = -1;
  const int ew = p->sweep_waves(ek);
  names.push_back(std::string("eval_kernel<") + (p->prog.is2d ? "2" : "3") + ", " + std::to_string(ek) + ", " + std::to_string(ew) + ">");
  if (!p->prog.is2d) {
    names.push_back("prune_kernel");
    names.push_back("prune_spec_kernel");
    names.push_back(std::string(fused_leaf() ? "leaf_kernel<" : "leaf_eval_kernel<") + std::to_string(lk) + ", " + std::to_string(lw) + (fused_leaf() ? ">" : (p->leaf_nt_in_lds() ? ", true, true, false, false>" : ", true, false, false, false>")));
    // a fifth workgroup per CU where the LDS has room for it (npt-flange's 7 slots): the 96-register build is taken if the
    // compiler reaches it without scratch (-3 % on the evaluating kernel); built beside the 128-register one, same process
    if (!fused_leaf() && lk == 4 && lw == 4 && 5 * lds_m <= (size_t)160 * 1024) names.push_back("leaf_eval_kernel<4, 5, true, true, false, false>");
    // both passes of a column brick in one body -- taken, ahead of the others, if the compiler reaches it without scratch: -7 %
    // where much of the program depends on x and y alone (an atan2, several hypots: npt-flange), -1..2 % elsewhere
    static const bool both_off = [] { const char* e = getenv("GSDF_HIP_NO_BOTH_PASSES"); return e && atoi(e) != 0; }();  // developer knob (A/B timing)
    static const int both_min = [] { const char* e = getenv("GSDF_HIP_BOTH_MIN_WEIGHT"); return e ? atoi(e) : 0; }();  // developer knob. 0: always -- programs without x,y-only work gain 1-2 % too (bolt 1.029 -> 1.007 ms, knurled-cylinder 3.19 -> 3.16: one body of eight points schedules a little better than two of four)
    if (!fused_leaf() && !both_off && lk == 4 && gsdf_dev::spec_xy_shared_weight(p->prog) >= both_min) {
      both_at = (int)names.size();
      names.push_back(std::string("leaf_eval_kernel<4, ") + std::to_string(lw) + (p->leaf_nt_in_lds() ? ", true, true, true, false>" : ", true, false, true, false>"));
      // ... and at five workgroups per CU (96 registers) where the LDS has room: taken if it builds without scratch
      static const bool both5_off = [] { const char* e = getenv("GSDF_HIP_NO_BOTH5"); return e && atoi(e) != 0; }();  // developer knob (A/B timing)
      if (!both5_off && lw == 4 && 5 * lds_m <= (size_t)160 * 1024) {
        both5_at = (int)names.size();
        names.push_back(std::string("leaf_eval_kernel<4, 5") + (p->leaf_nt_in_lds() ? ", true, true, true, false>" : ", true, false, true, false>"));
      }
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
      if (both5_at >= 0) {
        const bool ok5b = fn_scratch_bytes(f[(size_t)both5_at]) == 0;
        spec_report("specialised", names[(size_t)both5_at], f[(size_t)both5_at], ok5b);
        if (ok5b) { p->f_leaf = f[(size_t)both5_at]; p->spec_leaf_w = 5; p->spec_leaf_both = true; okl = true; }
      }
    }
    // the leaf kernel is where the time goes: before giving it up, trade occupancy for registers (W = workgroups per CU
    // the register budget is sized for; the launch is the same)
    for (int w2 = lw - 1; !okl && w2 >= 2; w2--) {
      std::vector<hipFunction_t> fl;
      hipModule_t m2 = nullptr;
      const std::string nl = std::string(fused_leaf() ? "leaf_kernel<" : "leaf_eval_kernel<") + std::to_string(lk) + ", " + std::to_string(w2) + (fused_leaf() ? ">" : (p->leaf_nt_in_lds() ? ", true, true, false, false>" : ", true, false, false, false>"));
      if (spec_build(p, {nl}, &m2, fl, &p->spec_compile_s) != GSDF_OK) break;
      okl = fn_scratch_bytes(fl[0]) == 0;
      spec_report("specialised", nl, fl[0], okl);
      if (okl) { p->f_leaf = fl[0]; p->spec_leaf_w = w2; p->spec_mod4 = m2; }
      else (void)hipModuleUnload(m2);
    }
  }
  return GSDF_OK;
}
// ---- specialisation in the background -----------------------------------------------------------------------------------------
// Every example of the reference is one tree -> one mesh -> one file (examples/npt-flange/flange.go:61-98): a caller that waits for
// a compiler before its only mesh has gained nothing. _async starts the build (and the module load) on a thread of its own and
// returns; the handle keeps meshing and evaluating through the interpreter kernels and switches to the specialised ones at the
// first entry-point call after they are ready -- same bits either way (same statements, same flags), so a mesh may even change
// kernels between its attempts. With GSDF_HIP_CACHE_DIR set and the tree seen before, "ready" is a file read away.
void spec_adopt_slow(gsdf_program* p) {
  std::lock_guard<std::mutex> lk(p->spec_async_mu);
  if (p->spec_async.load(std::memory_order_acquire) != 2) return;  // another thread of the caller was faster
  if (p->spec_thread.joinable()) p->spec_thread.join();
  gsdf_program* q = p->spec_shadow;
  p->spec_shadow = nullptr;
  if (q && !p->spec_mod) {
    p->spec_compile_s += q->spec_compile_s; p->spec_compiler = q->spec_compiler; p->spec_key = q->spec_key;
    p->spec_eval_k = q->spec_eval_k; p->spec_eval_w = q->spec_eval_w; p->spec_leaf_k = q->spec_leaf_k; p->spec_leaf_w = q->spec_leaf_w;
    p->spec_leaf_both = q->spec_leaf_both;
    p->spec_mod3 = q->spec_mod3; p->spec_mod4 = q->spec_mod4;
    std::atomic_thread_fence(std::memory_order_release);  // (the configuration before the functions: a reader that sees a function sees what it was built for)
    p->f_prune = q->f_prune; p->f_prune_spec = q->f_prune_spec; p->f_leaf = q->f_leaf; p->f_eval = q->f_eval;
    p->spec_mod = q->spec_mod;
    q->spec_mod = q->spec_mod3 = q->spec_mod4 = nullptr;
  }
  if (q) gsdf_hip_program_destroy(q);
  p->spec_async.store(0, std::memory_order_release);
}

extern "C" int gsdf_hip_program_specialize_async(gsdf_program* p) {
  if (!p) return fail(GSDF_ERR_BAD_ARGUMENT, "null program");
  std::lock_guard<std::mutex> lk(p->spec_async_mu);
  if (p->spec_mod || p->spec_async.load() != 0) return GSDF_OK;  // specialised already, or a build is under way / waiting to be adopted
  if (gsdf_dev::spec_instruction_count(p->prog) > 4000)
    return fail(GSDF_ERR_BAD_TREE, "program too large to specialise (more than 4000 instructions): the interpreter kernels stay in use");
  gsdf_program* q = new (std::nothrow) gsdf_program;
  if (!q) return fail(GSDF_ERR_CAPACITY, "out of memory");
  q->prog = p->prog; q->device = p->device; q->num_cu = p->num_cu;  // all a build reads (no streams, no device memory)
  p->spec_shadow = q;
  p->spec_async_err.clear();
  p->spec_async.store(1, std::memory_order_release);
  try {
    p->spec_thread = std::thread([p, q] {
      const int rc = gsdf_hip_program_specialize(q);
      if (rc != GSDF_OK) p->spec_async_err = gsdf_hip_last_error();  // (the error text is the building thread's)
      p->spec_async.store(rc == GSDF_OK ? 2 : 3, std::memory_order_release);
    });
  } catch (...) {
    p->spec_shadow = nullptr;
    gsdf_hip_program_destroy(q);
    p->spec_async.store(0);
    return fail(GSDF_ERR_CAPACITY, "cannot start the build thread");
  }
  return GSDF_OK;
}

/* 1: the handle runs specialised kernels now; 0: the build is still under way (wait = 0) or none was started; a negative status if
 * the build failed (the interpreter kernels stay in use; gsdf_hip_last_error has the compiler's words). wait != 0 blocks until the
 * build has finished. */
extern "C" int gsdf_hip_program_specialize_poll(gsdf_program* p, int wait) {
  if (!p) return fail(GSDF_ERR_BAD_ARGUMENT, "null program");
  if (wait) {
    std::unique_lock<std::mutex> lk(p->spec_async_mu);
    if (p->spec_async.load() == 1 && p->spec_thread.joinable()) p->spec_thread.join();
  }
  spec_adopt(p);
  if (p->spec_async.load(std::memory_order_acquire) == 3) {
    std::lock_guard<std::mutex> lk(p->spec_async_mu);
    if (p->spec_thread.joinable()) p->spec_thread.join();
    if (p->spec_shadow) { gsdf_hip_program_destroy(p->spec_shadow); p->spec_shadow = nullptr; }
    p->spec_async.store(0);
    return fail(GSDF_ERR_HIP, "background specialisation failed: " + p->spec_async_err);
  }
  return p->spec_mod ? 1 : 0;
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
  char buf[512];
  if (p->prog.is2d)
    snprintf(buf, sizeof buf, "eval=eval_kernel<2,%d,%d>:%s", ek, ew, se ? "specialised" : "interpreter");
  else
    snprintf(buf, sizeof buf, "eval=eval_kernel<3,%d,%d>:%s leaf=%s<%d,%d%s>:%s prune=prune_kernel:%s", ek, ew, se ? "specialised" : "interpreter",
             fused_leaf() ? "leaf_kernel" : "leaf_eval_kernel", lk, sl ? p->spec_leaf_w : aw, sl && p->spec_leaf_both ? ",both" : "", sl ? "specialised" : "interpreter",
             p->f_prune ? "specialised" : "interpreter");
  if (p->f_leaf_dense && strlen(buf) + 64 < sizeof buf)  // the evaluating kernel of share_corners = 1, once it has been built
    snprintf(buf + strlen(buf), sizeof buf - strlen(buf), " leaf_dense=leaf_dense_kernel<%d>:specialised", p->spec_leaf_dense_w);
  if (p->f_leaf_dz && strlen(buf) + 64 < sizeof buf)  // the evaluating kernel of share_corners = 2, once it has been built
    snprintf(buf + strlen(buf), sizeof buf - strlen(buf), " leaf_rows=leaf_eval_kernel<4,%d%s,rows>:specialised", p->spec_leaf_dz_w, p->spec_leaf_dz_both ? ",both" : "");
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
                                                   : std::vector<std::string>{"eval_kernel<3, 4, 4>", "prune_kernel", "prune_spec_kernel", "leaf_eval_kernel<4, 4, true, true, false, false>", "leaf_eval_kernel<4, 4, true, true, true, true>", "leaf_dense_kernel<4, true, true>", "leaf_kernel<4, 4>", "flat_grid_kernel<4, 4>"};
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
  if (p->spec_thread.joinable()) p->spec_thread.join();  // a build under way finishes first (its thread writes into this handle)
  if (p->spec_shadow) { gsdf_hip_program_destroy(p->spec_shadow); p->spec_shadow = nullptr; }
  if (p->d_code) (void)hipFree(p->d_code);
  if (p->d_pos) (void)hipFree(p->d_pos);
  if (p->d_dist) (void)hipFree(p->d_dist);
  if (p->h_pos) (void)hipHostFree(p->h_pos);
  if (p->h_dist) (void)hipHostFree(p->h_dist);
  for (auto& sl : p->slot) {
    if (sl.h_pos) (void)hipHostFree(sl.h_pos);
    if (sl.h_dist) (void)hipHostFree(sl.h_dist);
    if (sl.h_flag) (void)hipHostFree(sl.h_flag);
    if (sl.s) (void)hipStreamDestroy(sl.s);
  }
  if (p->spec_mod) (void)hipModuleUnload(p->spec_mod);
  if (p->spec_mod2) (void)hipModuleUnload(p->spec_mod2);
  if (p->spec_mod3) (void)hipModuleUnload(p->spec_mod3);
  if (p->spec_mod4) (void)hipModuleUnload(p->spec_mod4);
  if (p->spec_mod_dz) (void)hipModuleUnload(p->spec_mod_dz);
  if (p->spec_mod_k1) (void)hipModuleUnload(p->spec_mod_k1);
  if (p->spec_mod_dense) (void)hipModuleUnload(p->spec_mod_dense);
  p->q0.release(); p->q1.release(); p->ctr.release();
  p->rec.release(); p->hdr.release(); p->grp.release();
  p->b_q0.release(); p->b_q1.release(); p->b_ctr.release(); p->b_spec_pass.release(); p->b_rec.release(); p->b_hdr.release(); p->b_grp.release();
  p->c_q0.release(); p->c_q1.release(); p->c_ctr.release(); p->c_spec_pass.release(); p->c_rec.release(); p->c_hdr.release(); p->c_grp.release();
  if (p->stream_b) (void)hipStreamDestroy(p->stream_b);
  if (p->stream_c) (void)hipStreamDestroy(p->stream_c);
  p->flat_grid.release(); p->flat_bits.release(); p->flat_list.release(); p->dc_tile.release(); p->dc_grid.release(); p->dc_dist.release(); p->dc_fv.release(); p->dc_nrm.release(); p->dc_edge.release(); p->dc_erun.release(); p->dc_flag.release();
  for (auto e : p->ev) if (e) (void)hipEventDestroy(e);
  for (auto e : p->ev_b) if (e) (void)hipEventDestroy(e);
  for (auto e : p->ev_c) if (e) (void)hipEventDestroy(e);
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

// host_mapped: the buffers are pinned host memory the kernel reaches across PCIe (the host-buffer API's small and registered calls).
// Such a call is bound by the reads' latency and bandwidth, not by the evaluation: ONE point per lane puts four times as many
// workgroups on the bus at once (32 768 points: 128 workgroups instead of 32) -- 24.6 -> 19.6 us per blocking call from registered
// buffers, 31.4 -> 26.1 us from pageable ones (tools/gpu_dropin_k.sh).
static int eval_dev(gsdf_program* p, int dim, const void* d_pos, size_t stride_bytes, float* d_dist, size_t n, hipStream_t s, bool count = true, bool host_mapped = false) {
  if (stride_bytes % 4 != 0 || stride_bytes < (size_t)dim * 4) return fail(GSDF_ERR_BAD_ARGUMENT, "bad position stride");
  spec_adopt(p);  // (a background build that has finished: its kernels from here on)
  static const bool k1_off = [] { const char* e = getenv("GSDF_HIP_NO_EVAL_K1"); return e && atoi(e) != 0; }();  // developer knob (A/B timing)
  // This entry runs on several host threads at once (the host-buffer calls): the kernel it launches and what that kernel was built
  // for are read as ONE snapshot under the mutex an adopting thread holds while it swaps them (spec_adopt_slow).
  hipFunction_t f_eval = nullptr, f_eval_k1 = nullptr;
  int spec_eval_k = 0;
  {
    std::lock_guard<std::mutex> lk(p->spec_async_mu);
    f_eval = p->f_eval; f_eval_k1 = p->f_eval_k1; spec_eval_k = p->spec_eval_k;
  }
  const bool latency = host_mapped && !k1_off && (f_eval_k1 != nullptr || f_eval == nullptr);  // (a specialised handle without a K = 1 build keeps its specialised kernel)
  const int k = latency ? 1 : p->batch_k();
  static const int eval_bpc = [] { const char* e = getenv("GSDF_HIP_EVAL_BPC"); return e ? atoi(e) : 64; }();  // tuning knob: finer grids drain evenly (8 -> 64 per CU: +10 % on npt-flange)
  const unsigned grid = grid_for((n + k - 1) / k, p->num_cu, eval_bpc);
  const uint32_t sf = (uint32_t)(stride_bytes / 4);
  const float* q = (const float*)d_pos;
  const uint64_t nn = (uint64_t)n;
  const int w = p->sweep_waves(k);
#define LAUNCH_EVAL(D, KK, WW) hipLaunchKernelGGL((eval_kernel<D, KK, WW>), dim3(grid), dim3(BLOCK), p->lds_bytes(KK), s, p->d_code, q, sf, d_dist, nn)
  if (latency && f_eval_k1) {
    HIP_TRY(launch_fn(f_eval_k1, grid, BLOCK, p->lds_bytes(1), s, (const uint32_t*)p->d_code, q, sf, d_dist, nn));
  } else if (f_eval && spec_eval_k == k) {  // whichever W the specialised kernel was built for: same launch
    HIP_TRY(launch_fn(f_eval, grid, BLOCK, p->lds_bytes(k), s, (const uint32_t*)p->d_code, q, sf, d_dist, nn));
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
  if (p->spec_mod && !p->spec_k1_tried) spec_eval_k1(p);
  rc = eval_dev(p, dim, dp, stride, (float*)dd, n_pos, sl.s, /*count=*/false, /*host_mapped=*/true);
  if (rc) return bail(rc);
  // completion flag behind the kernel (eval_wait polls it; hipStreamSynchronize remains the fallback)
  // (hipStreamWriteValue32 instead of the one-thread kernel was tried: slower, 38.7 vs 29.9 us per blocking pageable call)
  static const bool use_flag = [] { const char* e = getenv("GSDF_HIP_EVAL_FLAG"); return !e || atoi(e) != 0; }();  // developer knob (A/B timing)
  sl.flagged = false;
  if (use_flag) {
    if (!sl.h_flag) {
      void* hf = nullptr; void* df = nullptr;
      if (hipHostMalloc(&hf, 64, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess && hipHostGetDevicePointer(&df, hf, 0) == hipSuccess) {
        sl.h_flag = (unsigned*)hf; sl.d_flag = (unsigned*)df; *sl.h_flag = 0u;
      } else { (void)hipGetLastError(); if (hf) (void)hipHostFree(hf); }
    }
    if (sl.h_flag) {
      sl.flag_val = sl.flag_val + 1u ? sl.flag_val + 1u : 1u;
      hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(1), 0, sl.s, sl.d_flag, sl.flag_val);
      if (hipGetLastError() == hipSuccess) sl.flagged = true;
    }
  }
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
  hipError_t e = hipSuccess;
  bool done = false;
  if (sl.flagged) {  // poll the flag the stream writes behind the kernel: ~100 us of spinning at most, then the blocking wait
    const volatile unsigned* f = sl.h_flag;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; spin++) {
      if (*f == sl.flag_val) { done = true; break; }
      __builtin_ia32_pause();
      if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) break;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  }
  if (!done) e = hipStreamSynchronize(sl.s);
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
  spec_adopt(p);  // (a background build that has finished: its kernels from here on)
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

// glrender.ImageRendererSDF2.Render for a 2D program: w x h pixels over Bounds(); host outputs (either may be NULL).
extern "C" int gsdf_hip_image2(gsdf_program* p, int w, int h, float* dist_out, uint8_t* rgba_out) {
  if (!p) return fail(GSDF_ERR_BAD_ARGUMENT, "null program");
  if (!p->prog.is2d) return fail(GSDF_ERR_DIMENSION, "program is 3D, image2 called");
  if (w <= 0 || h <= 0) return fail(GSDF_ERR_BAD_ARGUMENT, "bad image size");
  HIP_TRY(hipSetDevice(p->device));
  spec_adopt(p);  // (a background build that has finished: its kernels from here on)
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
