cd $GRAFT_REPO_ROOT
timeout 600 python - <<'PY' 2>&1 | tail -14
import numpy as np, time, torch
from scaffold.builder import Builder, NutCircular
from gsdf_amd import hip
hip.init(0)
b = Builder()
cases = {
 "sphere": b.NewSphere(1.0),
 "cylinder r0": b.NewCylinder(1.0, 2.0, 0),
 "cylinder round": b.NewCylinder(1.0, 2.0, 0.1),
 "3 cylinders union": b.Union(b.NewCylinder(1.0, 2.0, 0), b.Translate(b.NewCylinder(1.2, 1.0, 0.1), 0, 0, 0.5), b.NewCylinder(0.3, 4, 0)),
 "iso thread poly (2D, 12 v)": b.ISOThread(0.84, 1/14., False),
 "screw(npt)": b.ScrewNPT(0.5, 1.0),
 "nut npt": b.NutNPT(0.5, NutCircular),
 "npt-flange": b.Scene("npt-flange"),
}
n = 1 << 24
ts = torch.cuda.Stream()
for name, sh in cases.items():
    sdf = hip.SDFHIP(sh)
    dim = 2 if sdf.is2d else 3
    tp = (torch.rand((n, dim), device="cuda") * 2 - 1).contiguous()
    td = torch.empty(n, device="cuda")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(ts):
        for _ in range(3): sdf.evaluate_dev(tp.data_ptr(), 4*dim, td.data_ptr(), n, ts.cuda_stream)
        e0.record(ts)
        for _ in range(10): sdf.evaluate_dev(tp.data_ptr(), 4*dim, td.data_ptr(), n, ts.cuda_stream)
        e1.record(ts); ts.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name:28s} {ms:7.3f} ms  {n/ms/1e6:8.1f} Gevals/s   ~{2.38e9*1024*32/ (n/ms*1e3):7.0f} VALU-lane-cycles/eval", sdf.info())
PY
