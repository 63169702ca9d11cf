#!/bin/bash
# flat renderer on device: parity tests + timings at resdiv 400 / 1600 (interpreter and specialised kernels)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 1500 python -m pytest tests/test_gpu_flat.py -m gpu -x -q 2>&1 | tail -5
timeout 600 python - <<'PY'
import numpy as np
from scaffold.builder import Builder
from gsdf_amd import hip
hip.init(0)
b = Builder()
s = b.Scene("npt-flange")
for spec in (False, True):
    sdf = hip.SDF3HIP(s)
    if spec:
        sdf.specialize()
    for resdiv in (400, 1600):
        res = np.float32(float(s.Diagonal()) / resdiv)
        best = None
        for _ in range(4):
            f = hip.FlatHIP(sdf, res)
            st = f.stats
            if best is None or st.ms_total < best[0]:
                best = (st.ms_total, st.ms_leaf, st.ms_march, st.evals, st.n_tris, st.active_leaves)
            del f
        ms, mg, mm, ev, nt, na = best
        print(f"flat npt-flange {resdiv} {'spec' if spec else 'interp'}: total {ms:.3f} ms grid {mg:.3f} ms ({ev / mg / 1e6:.1f} G evals/s) "
              f"march {mm:.3f} ms ({(4 * ev + 36 * nt) / mm / 1e6:.0f} GB/s) tris {nt} active {na}", flush=True)
PY
