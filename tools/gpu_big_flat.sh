#!/bin/bash
# One-off: device FlatRenderer on a lattice beyond 2^31 corners (npt-flange resdiv 3200: 3.36 G corners, 13.4 GB grid)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python - <<'PY'
import numpy as np
from scaffold.builder import Builder
from gsdf_amd import hip
hip.init(0)
b = Builder()
s = b.Scene("npt-flange")
sdf = hip.SDF3HIP(s)
sdf.specialize()
res = np.float32(float(s.Diagonal()) / 3200)
fl = hip.FlatHIP(sdf, res)
st = fl.stats
print(f"flat 3200: evals {st.evals} cubes {st.leaf_cubes} tris {st.n_tris} grid {st.ms_leaf:.2f} ms march {st.ms_march:.2f} ms")
oc = hip.OctreeHIP(sdf, res)
print(f"octree 3200: tris {oc.n_tris()} in {oc.stats.ms_total:.2f} ms; difference {int(st.n_tris) - oc.n_tris()}")
# thirds of the lattice as z-slabs: union must be the whole
parts = [hip.FlatHIP(sdf, res, shard_rank=r, shard_count=3) for r in range(3)]
print("slabs:", [p.n_tris() for p in parts], "sum", sum(p.n_tris() for p in parts))
def srt(t):
    t = np.ascontiguousarray(t, np.float32).reshape(-1, 9); return t[np.lexsort(t.view(np.uint32).T[::-1])]
w = srt(fl.RenderAll())
u = srt(np.concatenate([p.RenderAll().reshape(-1, 9) for p in parts]))
print("union of slabs identical to whole:", w.shape == u.shape and bool((w.view(np.uint32) == u.view(np.uint32)).all()))
PY
