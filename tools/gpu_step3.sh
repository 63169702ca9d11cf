cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_nan.py -x -q 2>&1 | tail -15
for m in 1 2; do
GSDF_HIP_EVAL_FLAG=$m timeout 600 python - <<'PY'
import json,sys,os
sys.path.insert(0,'.')
import bench, numpy as np
from scaffold.builder import Builder
from gsdf_amd import hip
hip.init(0)
sh=Builder().Scene("npt-flange"); sdf=hip.SDF3HIP(sh); sdf.specialize()
r=bench.evaluate_dropin(hip,sdf,sh)
print("flag mode", os.environ["GSDF_HIP_EVAL_FLAG"], {k:(round(v["us_per_call"],1), round(v["evals_per_s"]/1e9,3)) for k,v in r.items() if isinstance(v,dict)})
PY
done
