#!/bin/bash
# rocprofv3 kernel statistics of the STL path (stl_kernel: 36 B in + 50 B out per triangle, HBM-bound)
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-stl}; rm -rf $OUT; mkdir -p $OUT
cat > /tmp/stl.py <<'PY'
import sys
import numpy as np
sys.path.insert(0, "/root/repo")
from scaffold.builder import Builder
from gsdf_amd import hip
hip.init(0)
b = Builder()
s = b.Scene("npt-flange")
sdf = hip.SDF3HIP(s)
sdf.specialize()
res = np.float32(float(s.Diagonal()) / 1600)
for _ in range(8):
    oc = hip.OctreeHIP(sdf, res)
    v = oc.stl_view()
    n = oc.n_tris()
    del v, oc
print("triangles", n)
PY
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python /tmp/stl.py > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt
cat $OUT/*/*kernel_stats.csv | cut -c1-40,100-220 | head -6
find $OUT -name "*kernel_trace.csv" -delete
