#!/bin/bash
# per-kernel times of the device FlatRenderer (npt-flange resdiv 1600, specialised lattice kernel) under rocprofv3
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-flat4}
set --
mkdir -p $OUT
cat > /tmp/flat_one.py <<'PY'
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from scaffold.builder import Builder
from gsdf_amd import hip
hip.init(0)
b = Builder()
s = b.Scene("npt-flange")
sdf = hip.SDF3HIP(s)
sdf.specialize()
res = np.float32(float(s.Diagonal()) / 1600)
for _ in range(8):
    f = hip.FlatHIP(sdf, res)
    st = f.stats
    print(f"grid {st.ms_leaf:.3f} march {st.ms_march:.3f} tris {st.n_tris}", flush=True)
    del f
PY
for knobs in "4 8" "8 8" "2 8" "4 4" "4 16" "16 8"; do set -- $knobs; echo "scan_bpc $1 list_bpc $2: $(GSDF_HIP_FLAT_SCAN_BPC=$1 GSDF_HIP_FLAT_LIST_BPC=$2 timeout 300 python /tmp/flat_one.py | sort -k4 -n | head -1)"; done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python /tmp/flat_one.py > $OUT/trace.log 2>&1
tail -3 $OUT/trace.log
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
head -8 "$f"
cp "$f" $OUT/kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -size +4M -delete
