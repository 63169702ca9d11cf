#!/bin/bash
# developer experiment: flat_cut_scan_kernel built with -DGSDF_EXP_SCAN_* (tools/exp/lib_*.so, timing only): where its time goes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
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
for _ in range(6):
    f = hip.FlatHIP(sdf, res)
    st = f.stats
    print(f"grid {st.ms_leaf:.3f} march {st.ms_march:.3f} tris {st.n_tris}", flush=True)
    del f
PY
cp gsdf_amd/csrc/libgsdfhip.so /tmp/lib_base.so
for v in base "$@"; do
  [ $v = base ] && cp /tmp/lib_base.so gsdf_amd/csrc/libgsdfhip.so || cp tools/exp/lib_$v.so gsdf_amd/csrc/libgsdfhip.so
  (cd /tmp; rm -rf /tmp/tr_$v; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$v -- python /tmp/flat_one.py > /tmp/tr_$v.log 2>&1)
  f=$(find /tmp/tr_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v: $(grep -E 'flat_cut_scan|flat_march_list' "$f" | awk -F, '{gsub(/"/,"",$1); printf "%s avg %.1f us min %.1f us; ", substr($1,1,22), $(NF-4)/1000, $(NF-2)/1000}')"
done
cp /tmp/lib_base.so gsdf_amd/csrc/libgsdfhip.so
