#!/bin/bash
# latency of the host-buffer Evaluate drop-in for reference-sized batches
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for mode in default; do
GSDF_HIP_SMALL_MODE=$mode timeout 300 python - <<'PY'
import os, time
import numpy as np
from gsdf_amd.builder import Builder
from gsdf_amd import hip
from oracle.oracle import OracleSDF
hip.init(0)
b = Builder()
s = b.Scene("npt-flange")
sdf = hip.SDF3HIP(s)
sdf.specialize()
rng = np.random.default_rng(0)
bb = s.Bounds()
for n in (4096, 32768, 65536):
    pos = (bb[:3] + rng.random((n, 3), np.float32) * (bb[3:] - bb[:3])).astype(np.float32)
    dist = np.empty(n, np.float32)
    for _ in range(50):
        sdf.Evaluate(pos, dist)
    t0 = time.perf_counter()
    for _ in range(400):
        sdf.Evaluate(pos, dist)
    dt = (time.perf_counter() - t0) / 400
    ok = (dist.view(np.uint32) == OracleSDF(s.tree()).Evaluate(pos).view(np.uint32)).all() if n == 4096 else True
    print(f"mode {os.environ['GSDF_HIP_SMALL_MODE']} n {n}: {dt * 1e6:.1f} us per call, {n / dt / 1e9:.3f} G evals/s, exact {bool(ok)}")
PY
done
