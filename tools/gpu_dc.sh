cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/dc_tmp; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/dc.py <<'PY'
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from scaffold.builder import Builder
from gsdf_amd import hip
hip.init(0)
b = Builder()
for name, rd in (("glyph-plate", 800), ("npt-flange", 800)):
    sh = b.Scene(name)
    res = np.float32(float(sh.Diagonal()) / rd)
    sdf = hip.SDFHIP(sh)
    if len(sys.argv) > 1: sdf.specialize()
    dc = hip.DualContourHIP(sdf, res)
    t0 = time.perf_counter()
    for _ in range(3): dc.Reset(sdf, res)
    dt = (time.perf_counter() - t0) / 3
    s = dc.stats
    print(name, rd, "tris", s.n_tris, "evals", s.evals, "ms_total %.3f wall %.3f ms" % (s.ms_total, dt * 1e3), sdf.info())
PY
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python /tmp/dc.py $1 > $OUT/log.txt 2>&1
grep "tris" $OUT/log.txt
cat $OUT/*/*kernel_stats.csv | cut -c1-60,150-260 | head -14
rm -rf $OUT
