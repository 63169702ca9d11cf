cd $GRAFT_REPO_ROOT
timeout 600 python - <<'PY' 2>&1 | tail -12
import numpy as np, time
from gsdf_amd.builder import Builder
from gsdf_amd import hip
hip.init(0)
b = Builder(); s = b.Scene("npt-flange"); sdf = hip.SDF3HIP(s)
bb = s.Bounds(); rng = np.random.default_rng(0)
for n in (4096, 32768, 1<<20, 1<<24):
    pos = (bb[:3] + rng.random((n, 3), np.float32) * (bb[3:] - bb[:3])).astype(np.float32)
    dist = np.empty(n, np.float32)
    for _ in range(3): sdf.Evaluate(pos, dist)
    reps = max(3, min(200, (1<<24)//n))
    t = time.perf_counter()
    for _ in range(reps): sdf.Evaluate(pos, dist)
    dt = (time.perf_counter() - t) / reps
    print(f"host Evaluate n={n}: {dt*1e6:.1f} us/call, {n/dt/1e9:.3f} Gevals/s, {n*16/dt/1e9:.2f} GB/s over PCIe")
PY
