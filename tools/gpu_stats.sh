cd $GRAFT_REPO_ROOT
timeout 600 python - <<'PY' 2>&1 | tail -14
import numpy as np
from scaffold.builder import Builder
from gsdf_amd import hip
hip.init(0)
b = Builder()
for name, rd in (("npt-flange", 1600), ("bolt", 2000), ("knurled-cylinder", 2000)):
    sh = b.Scene(name)
    res = np.float32(float(sh.Diagonal()) / rd)
    oc = hip.OctreeHIP(hip.SDFHIP(sh), res)
    oc.Reset(oc.sdf, res)
    s = oc.stats
    print(name, rd, dict(n_tris=s.n_tris, evals=s.evals, leaf_cubes=s.leaf_cubes, active=s.active_leaves, pruned=s.pruned_leaves,
          evals_prune=s.evals_prune, evals_leaf=s.evals_leaf, ms=(round(s.ms_total,3), round(s.ms_prune,3), round(s.ms_leaf,3))),
          "active/leaf=%.3f tris/active=%.2f" % (s.active_leaves / max(1, s.leaf_cubes), s.n_tris / max(1, s.active_leaves)))
PY
