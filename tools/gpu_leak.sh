#!/bin/bash
# leak check: device memory and process RSS over a few hundred meshes through every result path
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python - <<'PY'
import ctypes as C, os, resource
import numpy as np
from scaffold.builder import Builder
from gsdf_amd import hip
hip.init(0)
rt = C.CDLL("libamdhip64.so")
def free_mb():
    f, t = C.c_size_t(), C.c_size_t()
    rt.hipMemGetInfo(C.byref(f), C.byref(t))
    return f.value / 1e6
def rss_mb():
    return int(open("/proc/self/statm").read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 1e6
b = Builder()
s = b.Scene("npt-flange")
sdf = hip.SDF3HIP(s); sdf.specialize()
res = np.float32(float(s.Diagonal()) / 800)
def round_():
    oc = hip.OctreeHIP(sdf, res); v = oc.triangles_view(); st = oc.stl_view(); n = int(v.shape[0]) + len(st); del v, st, oc
    ho = hip.OctreeHIP(sdf, res, host_output=True); v = ho.triangles_view(); n += v.shape[0]; del v, ho
    fl = hip.FlatHIP(sdf, res); a = fl.RenderAll(); n += a.shape[0]; del a, fl
    dc = hip.DualContourHIP(sdf, np.float32(res * 2)); buf = np.empty((4096, 3, 3), np.float32)
    while True:
        k, eof = dc.ReadTriangles(buf); n += k
        if eof: break
    del dc
    pos = np.random.default_rng(0).random((32768, 3), np.float32)
    sdf.Evaluate(pos)
    return n
for _ in range(10): round_()
f0, r0 = free_mb(), rss_mb()
for i in range(150): round_()
f1, r1 = free_mb(), rss_mb()
print(f"device free {f0:.0f} -> {f1:.0f} MB (delta {f0 - f1:+.0f}), RSS {r0:.0f} -> {r1:.0f} MB (delta {r1 - r0:+.0f})")
assert abs(f0 - f1) < 64 and r1 - r0 < 256, "memory grows"
print("no growth")
PY
