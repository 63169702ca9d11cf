#!/bin/bash
# throughput of T host threads meshing npt-flange resdiv 1600 concurrently (one handle and stream per thread)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python - <<'PY'
import threading, time
import numpy as np
from scaffold.builder import Builder
from gsdf_amd import hip
hip.init(0)
b = Builder()
s = b.Scene("npt-flange")
res = np.float32(float(s.Diagonal()) / 1600)
for T in (1, 2, 3, 4):
    sdfs = []
    for _ in range(T):
        sdf = hip.SDF3HIP(s)
        sdf.specialize()
        for _ in range(20):
            hip.OctreeHIP(sdf, res)
        sdfs.append(sdf)
    N = 60
    evals = [0] * T
    def work(i):
        for _ in range(N):
            oc = hip.OctreeHIP(sdfs[i], res)
            evals[i] += oc.stats.evals
    th = [threading.Thread(target=work, args=(i,)) for i in range(T)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print(f"{T} thread(s): {T * N / dt:.0f} meshes/s, {sum(evals) / dt / 1e9:.1f} G evals/s, {dt / (T * N) * 1e3:.3f} ms per mesh")
PY
