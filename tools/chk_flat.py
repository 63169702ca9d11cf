import sys, numpy as np
sys.path.insert(0, '.')
from scaffold.builder import Builder
from gsdf_amd import hip
hip.init(0)
b = Builder()
for scene, divs in (("bolt", (300, 500, 700, 900)), ("knurled-cylinder", (300, 600)), ("npt-flange", (500, 800))):
    s = b.Scene(scene); sdf = hip.SDF3HIP(s); sdf.specialize()
    for d in divs:
        res = np.float32(float(s.Diagonal()) / d)
        a = hip.OctreeHIP(sdf, res).n_tris()
        n = hip.OctreeHIP(sdf, res, prune=False).n_tris()
        f = hip.FlatHIP(sdf, res).n_tris()
        r = hip.OctreeHIP(sdf, res, assume_sdf=True).n_tris()
        print(scene, d, "default", a, "noprune", n, "flat", f, "refpred", r, flush=True)
