#!/bin/bash
# flat renderer: parity tests, then the marching pass over the bit planes (default) against the float-stream pass (GSDF_HIP_FLAT_STREAM=1)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_flat.py -m gpu -x -q 2>&1 | tail -5
for st in 0 1; do
GSDF_HIP_FLAT_STREAM=$st timeout 600 python - <<'PY'
import os
import numpy as np
from scaffold.builder import Builder
from gsdf_amd import hip
hip.init(0)
b = Builder()
for scene, resdiv in (("npt-flange", 400), ("npt-flange", 1600), ("bolt", 1000), ("knurled-cylinder", 800)):
    s = b.Scene(scene)
    sdf = hip.SDF3HIP(s)
    sdf.specialize()
    res = np.float32(float(s.Diagonal()) / resdiv)
    best = None
    for _ in range(6):
        f = hip.FlatHIP(sdf, res)
        st = f.stats
        if best is None or st.ms_march < best[2]:
            best = (st.ms_total, st.ms_leaf, st.ms_march, st.evals, st.n_tris, st.active_leaves)
        del f
    ms, mg, mm, ev, nt, na = best
    print(f"stream={os.environ['GSDF_HIP_FLAT_STREAM']} flat {scene} {resdiv}: grid {mg:.3f} ms ({ev / mg / 1e6:.1f} G evals/s) "
          f"march {mm:.3f} ms ({(4 * ev + 36 * nt) / mm / 1e6:.0f} GB/s algorithmic) tris {nt} active {na}", flush=True)
PY
done
