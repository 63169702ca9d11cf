cd $GRAFT_REPO_ROOT
timeout 900 python - <<'PY' 2>&1 | tail -8
import sys, os, numpy as np
sys.path.insert(0, "tests")
from scaffold.builder import Builder
from gsdf_amd import hip
import test_gpu_mesh as T
hip.init(0)
b = Builder()
s = T._text_plate(b, "soypat/gsdf on MI355X")
print("program", hip.lower(s)[1], "slots", len(hip.lower(s)[0]), "words")
for spec in (False, True):
    sdf = hip.SDF3HIP(s)
    if spec: sdf.specialize()
    for rd in (800,):
        res = np.float32(float(s.Diagonal()) / rd)
        oc = hip.OctreeHIP(sdf, res); oc.Reset(sdf, res)
        st = oc.stats
        print("spec" if spec else "interp", "octree", rd, "tris", st.n_tris, "evals", st.evals, "ms %.3f (prune %.3f leaf %.3f)" % (st.ms_total, st.ms_prune, st.ms_leaf))
        dc = hip.DualContourHIP(sdf, res); dc.Reset(sdf, res)
        print("spec" if spec else "interp", "dualcontour", rd, "tris", dc.stats.n_tris, "evals", dc.stats.evals, "ms %.3f" % dc.stats.ms_total, sdf.info()["specialize_s"])
PY
