#!/bin/bash
# One-off fuzz of the polygon edge culling: random concave polygons of 6-48 vertices, extruded / revolved / combined,
# meshed finely enough for the culling to act, against the oracle (interpreter and specialised kernels).
cd $GRAFT_REPO_ROOT
timeout ${2:-1500} python - "$1" <<'PY' 2>&1 | tail -8
import sys, math, numpy as np
from scaffold.builder import Builder
from gsdf_amd import hip
from oracle.oracle import OracleSDF
hip.init(0)
lo, hi = [int(x) for x in sys.argv[1].split(":")]
def srt(t):
    t = np.ascontiguousarray(t, np.float32).reshape(-1, 9); return t[np.lexsort(t.view(np.uint32).T[::-1])]
bad = n = 0
for seed in range(lo, hi):
    r = np.random.default_rng(seed)
    b = Builder()
    nv = int(r.integers(6, 49))
    ang = np.sort(r.uniform(0, 2 * math.pi, nv))
    rad = r.uniform(0.25, 1.2, nv) * (1.0 if seed % 3 else r.uniform(0.9, 1.1))
    off = (0.0, 0.0) if seed % 4 else (float(r.uniform(-40, 40)), float(r.uniform(-40, 40)))
    poly = b.NewPolygon([(off[0] + float(c * math.cos(a)), off[1] + float(c * math.sin(a))) for a, c in zip(ang, rad)])
    kind = seed % 5
    if kind == 0:
        part = b.Extrude(poly, float(r.uniform(0.2, 1.0)))
    elif kind == 1:
        part = b.Union(b.Extrude(poly, 0.4), b.Translate(b.NewSphere(0.5), off[0] + 0.3, off[1], 0.2))
    elif kind == 2:
        part = b.Difference(b.Extrude(poly, 0.5), b.Translate(b.NewBox(0.4, 0.4, 2.0, 0.0), off[0], off[1], 0.0))
    elif kind == 3:
        part = b.Rotate(b.Extrude(poly, 0.3), float(r.uniform(0, 3)), (0.3, 0.5, 0.8))
    else:
        part = b.SmoothUnion(0.1, b.Extrude(poly, 0.3), b.Translate(b.NewCylinder(0.4, 0.8, 0.0), off[0], off[1], 0.0))
    try:
        ref = OracleSDF(part.tree())
        res = np.float32(float(part.Diagonal()) / int(r.integers(90, 220)))
        want = srt(ref.render_octree(res, 4096, True).tris)
        sdf = hip.SDF3HIP(part)
        for spec in (False, True):
            if spec: sdf.specialize()
            got = srt(hip.OctreeHIP(sdf, res).RenderAll())
            if got.shape != want.shape or not (got.view(np.uint32) == want.view(np.uint32)).all():
                bad += 1; print("FAIL seed", seed, "nv", nv, "kind", kind, "spec", spec, got.shape, want.shape)
        n += 1
    except Exception as ex:
        bad += 1; print("EXC seed", seed, repr(ex)[:200])
print("polygon trees", n, "failures", bad)
PY
