#!/bin/bash
# end-to-end cost of the STL path at resdiv 1600: device record build + transfer to the host
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 300 python - <<'PY'
import time
import numpy as np
from scaffold.builder import Builder
from gsdf_amd import hip
hip.init(0)
b = Builder()
s = b.Scene("npt-flange")
sdf = hip.SDF3HIP(s)
sdf.specialize()
res = np.float32(float(s.Diagonal()) / 1600)
oc = hip.OctreeHIP(sdf, res)
n = oc.n_tris()
for name, fn in (("WriteBinarySTL", oc.WriteBinarySTL), ("RenderAll", oc.RenderAll), ("stl_view (fresh mesh)", lambda: hip.OctreeHIP(sdf, res).stl_view()), ("triangles_view (fresh mesh)", lambda: hip.OctreeHIP(sdf, res).triangles_view()), ("triangles_view (fresh mesh, host_output)", lambda: hip.OctreeHIP(sdf, res, host_output=True).triangles_view())):
    out = fn()
    del out
    t0 = time.perf_counter()
    for _ in range(5):
        out = fn()
        nbytes = len(out) if isinstance(out, bytes) else out.nbytes
        del out  # one mesh alive at a time: its pinned host memory goes back to the pool for the next one
    dt = (time.perf_counter() - t0) / 5
    print(f"{name}: {dt * 1e3:.2f} ms for {n} triangles, {nbytes / dt / 1e9:.1f} GB/s")
a = hip.OctreeHIP(sdf, res)
h = hip.OctreeHIP(sdf, res, host_output=True)
def srt(t):
    t = np.ascontiguousarray(t, np.float32).reshape(-1, 9); return t[np.lexsort(t.view(np.uint32).T[::-1])]
print("host_output mesh identical:", bool((srt(a.triangles_view()).view(np.uint32) == srt(h.triangles_view()).view(np.uint32)).all()),
      "device ms", a.stats.ms_total, "host_output ms", h.stats.ms_total)
PY
timeout 300 python - <<'PY'
import time
import numpy as np
from scaffold.builder import Builder
from gsdf_amd import hip
hip.init(0)
b = Builder()
s = b.Scene("npt-flange")
sdf = hip.SDF3HIP(s)
sdf.specialize()
res = np.float32(float(s.Diagonal()) / 1600)
buf = np.empty((4096, 3, 3), np.float32)
for rep in range(3):
    t0 = time.perf_counter()
    oc = hip.OctreeHIP(sdf, res)
    total = 0
    while True:
        n, eof = oc.ReadTriangles(buf)
        total += n
        if eof:
            break
    dt = time.perf_counter() - t0
    print(f"mesh + ReadTriangles loop (4096 per call): {dt * 1e3:.2f} ms, {total} triangles")
    del oc
PY
