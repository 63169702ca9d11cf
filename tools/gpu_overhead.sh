#!/bin/bash
# host-side overhead of one mesh call: wall time of gsdf_hip_mesh_octree vs its device time (HIP events)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 300 python - <<'PY'
import time, ctypes as C
import numpy as np
from scaffold.builder import Builder
from gsdf_amd import hip
hip.init(0)
b = Builder()
s = b.Scene("npt-flange")
sdf = hip.SDF3HIP(s)
sdf.specialize()
res = np.float32(float(s.Diagonal()) / 1600)
L = hip.lib()
opts = hip.MeshOpts(1, 0, 1, 0, None, 0)
for _ in range(3):
    hip.OctreeHIP(sdf, res)
N = 20
t_call = t_obj = dev = 0.0
for _ in range(N):
    m = C.c_void_p()
    t0 = time.perf_counter()
    rc = L.gsdf_hip_mesh_octree(sdf._h, res, C.byref(opts), C.byref(m))
    t1 = time.perf_counter()
    st = hip.MeshStats()
    L.gsdf_hip_mesh_stats_get(m, C.byref(st))
    L.gsdf_hip_mesh_destroy(m)
    t_call += t1 - t0
    dev += st.ms_total
t0 = time.perf_counter()
for _ in range(N):
    oc = hip.OctreeHIP(sdf, res)
t_obj = time.perf_counter() - t0
print(f"C call wall {t_call / N * 1e3:.3f} ms, device {dev / N:.3f} ms, python object loop {t_obj / N * 1e3:.3f} ms")
PY
