#!/bin/bash
# GSDF_HIP_CACHE_DIR across processes on a GPU box: the second process loads the code object from disk and gets the same mesh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export GSDF_HIP_CACHE_DIR=/tmp/gsdf_cache; rm -rf $GSDF_HIP_CACHE_DIR; mkdir -p $GSDF_HIP_CACHE_DIR
for i in 1 2; do
timeout 300 python - <<'PY'
import hashlib, numpy as np
from scaffold.builder import Builder
from gsdf_amd import hip
hip.init(0)
s = Builder().Scene("bolt")
sdf = hip.SDF3HIP(s); sdf.specialize()
oc = hip.OctreeHIP(sdf, np.float32(float(s.Diagonal()) / 600))
t = oc.RenderAll().reshape(-1, 9); t = t[np.lexsort(t.view(np.uint32).T[::-1])]
print("specialize_s %.3f" % sdf.info()["specialize_s"], "tris", oc.n_tris(), hashlib.sha256(t.tobytes()).hexdigest()[:16])
PY
done
ls -la $GSDF_HIP_CACHE_DIR | tail -3
