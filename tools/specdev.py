#!/usr/bin/env python3
"""Developer tool (host only, no GPU): build the run-time specialised kernels of a scene from the CURRENT interp.h /
kernels.h and report registers / scratch per kernel; optionally dump the ISA.

  tools/specdev.py npt-flange "leaf_kernel<4, 4>" [--isa out.s] [--keep dir]
  tools/specdev.py fuzz2d:708:1 "eval_kernel<2, 4, 4>"
"""
import argparse, ctypes as C, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
CSRC = os.path.join(ROOT, "gsdf_amd", "csrc")
DEV = os.path.join(ROOT, "tools", "specdev")
LLVM = "/opt/rocm/lib/llvm/bin"


def preload_torch_rocm():
    """Bind the shim to PyTorch's bundled ROCm (hiprtc / comgr 7.0.x) instead of /opt/rocm's: bench.py and the GPU tests
    import torch before libgsdfhip.so, so THAT compiler builds the specialised kernels on the GPU box -- and its register
    allocation differs from the system compiler's."""
    import glob
    import importlib.util
    lib = os.path.join(os.path.dirname(importlib.util.find_spec("torch").origin), "lib")
    for n in ("libamd_comgr.so", "libamdhip64.so", "libhiprtc.so"):
        C.CDLL(os.path.join(lib, n), mode=C.RTLD_GLOBAL)


def build_shim():
    subprocess.check_call([sys.executable, "gen_embedded.py", "embedded_src.inc"], cwd=CSRC)
    so = os.path.join(DEV, "libspecdev.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           "-I" + os.path.join(ROOT, "include"), os.path.join(DEV, "specdev.cpp"), os.path.join(CSRC, "compile.cpp"),
                           os.path.join(CSRC, "specialize.cpp"), "-L/opt/rocm/lib", "-lhiprtc", "-Wl,-rpath,/opt/rocm/lib", "-o", so])
    return C.CDLL(so)


def scene(name):
    from scaffold.builder import Builder
    if name.startswith("fuzzlong:"):  # the trees of tools/gpu_fuzz_long.sh: fuzzlong:<seed>:<k>
        import fuzz_trees
        _, seed, idx = name.split(":")
        return fuzz_trees.random_shapes(int(seed), 10, depth=4)[1][int(idx)]
    if name.startswith("fuzz2d:") or name.startswith("fuzz3d:"):
        import fuzz_trees
        _, seed, idx = name.split(":")
        f = fuzz_trees.random_shapes2d if name.startswith("fuzz2d") else fuzz_trees.random_shapes
        _, shapes = f(int(seed), int(idx) + 5, depth=3)
        return shapes[int(idx)]
    return Builder().Scene(name)


def resources(elf):
    out = subprocess.check_output([LLVM + "/llvm-readelf", "--notes", elf], text=True)
    rows, d = [], {}
    for l in out.splitlines():
        m = re.search(r"\.(name|private_segment_fixed_size|vgpr_count|vgpr_spill_count|sgpr_spill_count|sgpr_count):\s+(\S+)", l)
        if m:
            d[m.group(1)] = m.group(2)
        if ".wavefront_size" in l:
            rows.append(d); d = {}
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene"); ap.add_argument("names")
    ap.add_argument("--isa"); ap.add_argument("--src")
    ap.add_argument("--system-rocm", action="store_true", help="compile with /opt/rocm's hiprtc instead of PyTorch's bundled one (default: PyTorch's, like the GPU box)")
    a = ap.parse_args()
    if not a.system_rocm:
        preload_torch_rocm()
    lib = build_shim()
    sh = scene(a.scene)
    t = sh.tree()
    tmp = tempfile.mkdtemp(prefix="specdev_")
    elf = os.path.join(tmp, "k.elf")
    if a.src:
        lib.specdev_source(C.byref(t), a.src.encode())
    log = C.create_string_buffer(1 << 20)
    os.environ.pop("GSDF_HIP_CACHE_DIR", None)
    rc = lib.specdev_build(C.byref(t), a.names.encode(), elf.encode(), log, len(log))
    if rc:
        print(log.value.decode()[-6000:]); sys.exit(rc)
    want = [n.split("<")[0].strip() for n in a.names.split(";")]
    for r in resources(elf):
        nm = subprocess.check_output(["c++filt", r["name"]], text=True).strip().split("(")[0]
        if any(w in nm for w in want):
            print(f"{nm}: vgpr {r.get('vgpr_count')} sgpr {r.get('sgpr_count')} scratch {r.get('private_segment_fixed_size')} "
                  f"vspill {r.get('vgpr_spill_count')} sspill {r.get('sgpr_spill_count')}")
    if a.isa:
        with open(a.isa, "w") as f:
            subprocess.check_call([LLVM + "/llvm-objdump", "-d", elf], stdout=f)
        print("isa ->", a.isa)
    print("elf ->", elf)


if __name__ == "__main__":
    main()
