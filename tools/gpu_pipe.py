"""Meshes per second of one program: blocking calls, start/wait on one stream, start/wait on the handle's two streams, two handles."""
import sys, time, threading
import numpy as np
import torch
sys.path.insert(0, ".")
from scaffold.builder import Builder
from gsdf_amd import hip
hip.init(0)
sh = Builder().Scene("npt-flange")
res = np.float32(float(sh.Diagonal()) / 1600)
sdf = hip.SDF3HIP(sh); sdf.specialize()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60

def blocking(n):
    for _ in range(n): oc = hip.OctreeHIP(sdf, res)
    return oc
def piped(n, stream=None):
    pend = hip.OctreeHIP.start(sdf, res, stream=stream)
    for k in range(n - 1):
        nxt = hip.OctreeHIP.start(sdf, res, stream=stream)
        oc = pend.wait(); pend = nxt
    return pend.wait()
ts = torch.cuda.Stream()
for name, fn in (("blocking", blocking), ("start/wait, two in flight", piped), ("start/wait, caller stream", lambda n: piped(n, ts.cuda_stream)), ("start/wait, two in flight", piped)):
    fn(30)
    t0 = time.perf_counter(); oc = fn(N); dt = time.perf_counter() - t0
    st = oc.stats
    print(f"{name:28s} {dt / N * 1e3:.4f} ms/mesh   device {st.ms_total:.4f} prune {st.ms_prune:.4f} eval {st.ms_march:.4f} march {st.ms_emit:.4f}", flush=True)
