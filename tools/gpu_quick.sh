cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
for K in 1 2 4; do
GSDF_HIP_BATCH_K=$K timeout 600 python - <<'PY' 2>&1 | tail -20
import numpy as np, time, os, torch
from gsdf_amd.builder import Builder
from gsdf_amd import hip
from oracle.oracle import OracleSDF
hip.init(0)
b = Builder(); s = b.Scene("npt-flange"); sdf = hip.SDF3HIP(s)
print("K", os.environ["GSDF_HIP_BATCH_K"], sdf.info())
rng = np.random.default_rng(1); bb = s.Bounds()
pos = (bb[:3] + rng.random((100003, 3), np.float32) * (bb[3:] - bb[:3])).astype(np.float32)
dg = sdf.Evaluate(pos); dc = OracleSDF(s.tree()).Evaluate(pos)
print("eval mismatches", int((dg.view(np.uint32) != dc.view(np.uint32)).sum()))
# resident eval throughput
n = 1 << 24
tp = torch.rand((n, 3), device="cuda") * 60 - 30
td = torch.empty(n, device="cuda")
for it in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    sdf.evaluate_dev(tp.data_ptr(), 12, td.data_ptr(), n); torch.cuda.synchronize()
    dt = time.perf_counter() - t
print("eval_dev 16M points: %.2f ms  %.2f Gevals/s" % (dt * 1e3, n / dt / 1e9))
for rd in (400, 1600):
    res = np.float32(float(s.Diagonal())/rd)
    for it in range(3):
        t=time.perf_counter(); oc = hip.OctreeHIP(sdf, res); dt=time.perf_counter()-t
        st=oc.stats
    print(rd, "tris", st.n_tris, "evals", st.evals, "leaf", st.leaf_cubes, "active", st.active_leaves, "ms total/prune/leaf", round(st.ms_total,3), round(st.ms_prune,3), round(st.ms_leaf,3), "wall", round(dt*1e3,2), "Gevals/s dev", round(st.evals/st.ms_total/1e6,2))
PY
done
