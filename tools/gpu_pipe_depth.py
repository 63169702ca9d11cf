"""How deep a pipeline of meshes pays: one handle keeps two meshes in flight (its two workspaces); H handles of the same tree keep 2 H.
Round 6 experiment: is the step (0.37 ms at depth 2 against 0.28 ms of evaluating kernel) limited by what overlaps the kernel?"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from scaffold.builder import Builder
from gsdf_amd import hip
hip.init(0)
scene, rd = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("npt-flange", 1600)
sh = Builder().Scene(scene)
res = np.float32(float(sh.Diagonal()) / rd)
N = 80
handles = []
for _ in range(3):
    s = hip.SDF3HIP(sh); s.specialize(); handles.append(s)

def run(n, hs, per):
    """round-robin over the handles, `per` meshes in flight on each"""
    q = []
    k = 0
    last = None
    for i in range(n):
        h = hs[i % len(hs)]
        q.append(hip.OctreeHIP.start(h, res))
        if len(q) >= per * len(hs):
            last = q.pop(0).wait()
    while q:
        last = q.pop(0).wait()
    return last

for name, hs, per in (("1 handle, 1 in flight", handles[:1], 1), ("1 handle, 2 in flight", handles[:1], 2), ("2 handles x 1", handles[:2], 1), ("2 handles x 2", handles[:2], 2), ("3 handles x 1", handles[:3], 1), ("3 handles x 2", handles[:3], 2), ("1 handle, 2 in flight", handles[:1], 2)):
    run(30, hs, per)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); oc = run(N, hs, per); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name:24s} {dt / N * 1e3:.4f} ms/mesh", flush=True)
