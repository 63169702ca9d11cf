cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
timeout 600 python - <<'PY' 2>&1 | tail -20
import numpy as np, time
from gsdf_amd.builder import Builder
from gsdf_amd import hip
hip.init(0)
b = Builder(); s = b.Scene("npt-flange"); sdf = hip.SDF3HIP(s)
print(sdf.info())
for rd in (400, 1600):
    res = np.float32(float(s.Diagonal())/rd)
    for it in range(4):
        t=time.perf_counter(); oc = hip.OctreeHIP(sdf, res); dt=time.perf_counter()-t
        st=oc.stats
        print(rd, "tris", st.n_tris, "evals", st.evals, "leaf", st.leaf_cubes, "active", st.active_leaves, "ms total/prune/leaf", round(st.ms_total,3), round(st.ms_prune,3), round(st.ms_leaf,3), "wall", round(dt*1e3,2), "Gevals/s dev", round(st.evals/st.ms_total/1e6,2))
PY
