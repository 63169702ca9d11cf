#!/bin/bash
# developer experiment: grid size of flat_cut_scan_kernel (workgroups per CU), default build and tools/exp/lib_LB8.so (8 waves per SIMD)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cat > /tmp/flat_one.py <<'PY'
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from scaffold.builder import Builder
from gsdf_amd import hip
hip.init(0)
b = Builder()
for scene, resdiv in (("npt-flange", 1600), ("npt-flange", 400), ("bolt", 1000)):
    s = b.Scene(scene)
    sdf = hip.SDF3HIP(s)
    sdf.specialize()
    res = np.float32(float(s.Diagonal()) / resdiv)
    best = 1e9
    for _ in range(8):
        f = hip.FlatHIP(sdf, res)
        best = min(best, f.stats.ms_march)
        del f
    print(f"{scene}@{resdiv} march {best:.3f}", end="; ")
print()
PY
cp gsdf_amd/csrc/libgsdfhip.so /tmp/lib_base.so
for v in base LB8; do
  [ $v = base ] && cp /tmp/lib_base.so gsdf_amd/csrc/libgsdfhip.so || cp tools/exp/lib_$v.so gsdf_amd/csrc/libgsdfhip.so
  for bpc in 4 8 12 16 24 32 64; do echo "$v scan_bpc $bpc: $(GSDF_HIP_FLAT_SCAN_BPC=$bpc timeout 300 python /tmp/flat_one.py)"; done
done
cp /tmp/lib_base.so gsdf_amd/csrc/libgsdfhip.so
