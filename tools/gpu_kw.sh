cd $GRAFT_REPO_ROOT
for cfg in "0 0" "0 4" "2 4" "2 3"; do
set -- $cfg
echo "BATCH_K=$1 LEAF_WAVES=$2"
GSDF_HIP_BATCH_K=$1 GSDF_HIP_LEAF_WAVES=$2 timeout 600 python - <<'PY' 2>&1 | tail -3
import numpy as np
from gsdf_amd.builder import Builder
from gsdf_amd import hip
hip.init(0)
b = Builder()
for name, rd in (("npt-flange", 1600), ("bolt", 2000), ("knurled-cylinder", 2000)):
    sh = b.Scene(name)
    res = np.float32(float(sh.Diagonal()) / rd)
    oc = hip.OctreeHIP(hip.SDFHIP(sh), res)
    best = 1e9
    for _ in range(4):
        oc.Reset(oc.sdf, res); best = min(best, oc.stats.ms_leaf)
    print(name, rd, oc.stats.n_tris, "leaf ms %.3f" % best)
PY
done
