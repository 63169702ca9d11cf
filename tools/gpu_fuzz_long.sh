#!/bin/bash
# One-off extended fuzzing (not part of the suite): many more seeds of tests/test_gpu_fuzz.py's random trees.
cd $GRAFT_REPO_ROOT
timeout ${2:-1500} python - "$1" <<'PY' 2>&1 | tail -15
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
bad = 0; n = 0
for seed in range(lo, hi):
    _, shapes = fuzz_trees.random_shapes(seed, 10, depth=4)
    rng = np.random.default_rng(seed)
    for k, sh in enumerate(shapes):
        try:
            ref = OracleSDF(sh.tree()); sdf = hip.SDF3HIP(sh)
            bb = sh.Bounds().astype(np.float32); c, h = (bb[:3] + bb[3:]) / 2, (bb[3:] - bb[:3]) / 2 * np.float32(1.1)
            pos = (c + (rng.random((4000, 3), np.float32) * 2 - 1) * h).astype(np.float32)
            d = ref.Evaluate(pos)
            e1 = mism(sdf.Evaluate(pos), d)
            res = np.float32(float(sh.Diagonal()) / 40)
            m = ref.render_octree(res, 4096, True)
            if (seed + k) % 2 == 0: sdf.specialize()
            oc = hip.OctreeHIP(sdf, res)
            e2 = (oc.n_tris() != m.n_tris) or (m.n_tris and not (srt(oc.RenderAll()).view(np.uint32) == srt(m.tris).view(np.uint32)).all())
            e3 = mism(sdf.Evaluate(pos), d)
            # the share_corners options: same set, and exactly the evaluations the oracle counts for the surviving bricks
            kern = sdf.info()["kernels"]; k4 = sdf.info()["leaf_k"] == 4 and m.levels >= 3
            for sc in (1, 2):
                o2 = hip.OctreeHIP(sdf, res, share_corners=sc)
                e2 = e2 or (o2.n_tris() != m.n_tris) or (m.n_tris and not (srt(o2.RenderAll()).view(np.uint32) == srt(m.tris).view(np.uint32)).all())
                kern = sdf.info()["kernels"]
                if k4 and sc == 2 and (kern.get("leaf_rows") or not sdf.info()["specialized"]): e2 = e2 or int(o2.stats.evals_leaf) != m.evals_rows
                if k4 and sc == 1: e2 = e2 or int(o2.stats.evals_leaf) != (m.evals_points_tails if kern.get("leaf_dense") else m.evals_points_256)
            fl = hip.FlatHIP(sdf, res)   # flat renderer: COLUMN-mode lattice pass + balanced marching pass
            mf = ref.render_flat(res, 4096, 2)
            e4 = (fl.n_tris() != mf.n_tris) or (fl.Evaluations() != mf.evals) or (mf.n_tris and not (srt(fl.RenderAll()).view(np.uint32) == srt(mf.tris).view(np.uint32)).all())
            n += 1
            if e1 or e2 or e3 or e4:
                bad += 1; print("FAIL seed", seed, "k", k, e1, bool(e2), e3, bool(e4))
        except Exception as ex:
            bad += 1; print("EXC seed", seed, "k", k, repr(ex)[:200])
print("trees", n, "failures", bad)
PY
