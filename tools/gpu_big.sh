#!/bin/bash
# One-off: npt-flange at resdiv 3200 (13 levels, ~27 M triangles): GPU (interpreter + specialised) vs the oracle.
cd $GRAFT_REPO_ROOT
timeout 1500 python - <<'PY' 2>&1 | tail -8
import hashlib, time, numpy as np
from scaffold.builder import Builder
from gsdf_amd import hip
from oracle.oracle import OracleSDF
hip.init(0)
s = Builder().Scene("npt-flange")
res = np.float32(float(s.Diagonal()) / 3200)
def dig(t):
    t = np.ascontiguousarray(t, np.float32).reshape(-1, 9)
    return hashlib.sha256(t[np.lexsort(t.view(np.uint32).T[::-1])].tobytes()).hexdigest()[:16]
out = []
for spec in (False, True):
    sdf = hip.SDF3HIP(s)
    if spec: sdf.specialize()
    oc = hip.OctreeHIP(sdf, res); oc.Reset(sdf, res)
    st = oc.stats
    d = dig(oc.RenderAll())
    print("spec" if spec else "interp", "levels", st.levels, "tris", st.n_tris, "evals", st.evals, "ms %.2f" % st.ms_total, d)
    out.append((st.n_tris, d))
    del oc
t0 = time.time()
m = OracleSDF(s.tree()).render_octree(res, 4096, True)
print("oracle tris", m.n_tris, dig(m.tris), "in %.0f s" % (time.time() - t0))
assert out[0] == out[1] == (m.n_tris, dig(m.tris))
print("OK identical")
PY
