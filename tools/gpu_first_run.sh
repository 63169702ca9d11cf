set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python - <<'PY' 2>&1 | tail -20
import numpy as np, time
from gsdf_amd.builder import Builder
from gsdf_amd import hip
hip.init(0)
b = Builder(); s = b.Scene("npt-flange"); sdf = hip.SDF3HIP(s)
print(sdf.info())
for rd in (400, 1600):
    res = np.float32(float(s.Diagonal())/rd)
    for it in range(3):
        t=time.perf_counter(); oc = hip.OctreeHIP(sdf, res); dt=time.perf_counter()-t
        st=oc.stats
        print(rd, "tris", st.n_tris, "evals", st.evals, "leaf", st.leaf_cubes, "active", st.active_leaves, "levels", st.levels, "ms total/prune/leaf/march", round(st.ms_total,2), round(st.ms_prune,2), round(st.ms_leaf,2), round(st.ms_march,2), "wall", round(dt*1e3,2))
PY
timeout 900 python bench.py --steps 5 --warmup 1 > gpurun_out/bench_r1_first.json 2> gpurun_out/bench_r1_first.err; tail -3 gpurun_out/bench_r1_first.err; cat gpurun_out/bench_r1_first.json
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof1.log 2>&1; tail -2 $GRAFT_REPO_ROOT/gpurun_out/prof1.log
find $GRAFT_REPO_ROOT/gpurun_out/prof1 -name "*stats*" | head
