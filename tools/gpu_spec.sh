cd $GRAFT_REPO_ROOT
timeout 900 python - <<'PY' 2>&1 | tail -12
import numpy as np, hashlib, time
from scaffold.builder import Builder
from gsdf_amd import hip
hip.init(0)
b = Builder()
for name, rd in (("npt-flange", 1600), ("bolt", 2000), ("knurled-cylinder", 2000), ("glyph-plate", 400)):
    sh = b.Scene(name)
    res = np.float32(float(sh.Diagonal()) / rd)
    out = {}
    for spec in (False, True):
        sdf = hip.SDFHIP(sh)
        if spec: sdf.specialize()
        oc = hip.OctreeHIP(sdf, res)
        best = 1e9
        for _ in range(4):
            oc.Reset(sdf, res); best = min(best, oc.stats.ms_leaf)
        tr = oc.RenderAll().reshape(-1, 9)
        o = np.lexsort(tr.T[::-1])
        dig = hashlib.sha256(tr[o].tobytes()).hexdigest()[:16]
        rng = np.random.default_rng(1); bb = sh.Bounds()
        pos = (bb[:3] + rng.random((200000, 3), np.float32) * (bb[3:] - bb[:3])).astype(np.float32)
        d = sdf.Evaluate(pos)
        out[spec] = (oc.stats.n_tris, dig, hashlib.sha256(d.tobytes()).hexdigest()[:16])
        print(name, rd, "spec" if spec else "interp", "leaf ms %.3f prune %.3f" % (best, oc.stats.ms_prune), out[spec], sdf.info())
    assert out[False] == out[True], name
PY
