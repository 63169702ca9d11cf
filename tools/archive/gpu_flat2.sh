#!/bin/bash
# flat renderer marching pass: grid-size sweep at resdiv 1600
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for bpc in 4 6 8 16 32; do
GSDF_HIP_FLAT_BPC=$bpc timeout 300 python - <<'PY'
import os
import numpy as np
from scaffold.builder import Builder
from gsdf_amd import hip
hip.init(0)
b = Builder()
s = b.Scene("npt-flange")
sdf = hip.SDF3HIP(s)
sdf.specialize()
res = np.float32(float(s.Diagonal()) / 1600)
best = None
for _ in range(4):
    f = hip.FlatHIP(sdf, res)
    st = f.stats
    if best is None or st.ms_march < best[0]:
        best = (st.ms_march, st.ms_leaf, st.evals, st.n_tris)
    del f
mm, mg, ev, nt = best
print(f"bpc {os.environ['GSDF_HIP_FLAT_BPC']}: grid {mg:.3f} ms march {mm:.3f} ms ({(4 * ev + 36 * nt) / mm / 1e6:.0f} GB/s) tris {nt}", flush=True)
PY
done
