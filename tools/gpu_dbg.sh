cd $GRAFT_REPO_ROOT
timeout 600 python - <<'PY' 2>&1 | tail -30
import sys; sys.path.insert(0,'tests')
import numpy as np, corpus
from gsdf_amd import hip
from oracle.oracle import OracleSDF
hip.init(0)
for fn in (corpus.shapes3d, corpus.shapes2d):
    _, shapes = fn()
    for name, sh in shapes:
        pos = corpus.sample_points(sh)
        dg = hip.SDFHIP(sh).Evaluate(pos); dc = OracleSDF(sh.tree()).Evaluate(pos)
        bad = np.where(dg.view(np.uint32) != dc.view(np.uint32))[0]
        if len(bad): print(name, len(bad), "of", len(pos), "e.g.", pos[bad[0]], dg[bad[0]], dc[bad[0]])
print("done")
PY
