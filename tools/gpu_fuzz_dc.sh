#!/bin/bash
# One-off extended fuzzing of dual contouring + normals + 2-D evaluation on random trees.
cd $GRAFT_REPO_ROOT
timeout ${2:-1500} python - "$1" <<'PY' 2>&1 | tail -12
import sys, numpy as np
sys.path.insert(0, "tests")
import fuzz_trees
from gsdf_amd import hip
from oracle.oracle import OracleSDF
hip.init(0)
lo, hi = [int(x) for x in sys.argv[1].split(":")]
def mism(a, b): return int(((a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))).sum())
def srt(t):
    t = np.ascontiguousarray(t, np.float32).reshape(-1, 9); return t[np.lexsort(t.view(np.uint32).T[::-1])]
bad = n = n2 = 0
for seed in range(lo, hi):
    _, shapes = fuzz_trees.random_shapes(seed, 6, depth=3)
    rng = np.random.default_rng(seed)
    for k, sh in enumerate(shapes):
        try:
            ref = OracleSDF(sh.tree()); sdf = hip.SDF3HIP(sh)
            if (seed + k) % 2: sdf.specialize()
            res = np.float32(float(sh.Diagonal()) / 36)
            m = ref.render_dualcontour(res, bool(k % 2))
            a = hip.DualContourHIP(sdf, res, chiseled=bool(k % 2)).RenderAll()
            ok = a.shape[0] == m.n_tris and (m.n_tris == 0 or (srt(a).view(np.uint32) == srt(m.tris).view(np.uint32)).all())
            n += 1
            if not ok: bad += 1; print("FAIL dc seed", seed, "k", k, a.shape[0], m.n_tris)
        except Exception as ex:
            print("skip seed", seed, "k", k, repr(ex)[:120])
    _, s2 = fuzz_trees.random_shapes2d(seed, 6, depth=3)
    for k, sh in enumerate(s2):
        bb = np.asarray(sh.Bounds(), np.float32); lo2, hi2 = bb[[0, 1]], bb[[3, 4]]
        pos = ((lo2 + hi2) / 2 + (rng.random((3000, 2), np.float32) * 2 - 1) * (hi2 - lo2) * np.float32(0.6)).astype(np.float32)
        sdf = hip.SDF2HIP(sh)
        if (seed + k) % 2: sdf.specialize()
        e = mism(sdf.Evaluate(pos), OracleSDF(sh.tree()).Evaluate(pos)); n2 += 1
        if e: bad += 1; print("FAIL 2d seed", seed, "k", k, e)
print("dc trees", n, "2d trees", n2, "failures", bad)
PY
