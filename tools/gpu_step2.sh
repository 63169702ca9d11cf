cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-step2}
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_nan.py tests/test_gpu_eval.py tests/test_gpu_prune_bounds.py tests/test_gpu_capi_replay.py -x -q 2>&1 | tail -15
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -3 $OUT/bench.err; python - $OUT/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','triangles_per_s','evaluator_build')}); print(d['phase_ms_rank0']); print(json.dumps(d.get('evaluate_dropin'),indent=0)); print(d['roofline']['frac'], d['roofline_march'] and d['roofline_march']['frac'])
PY
GSDF_HIP_EVAL_FLAG=0 timeout 600 python - <<'PY'
import json,sys
sys.path.insert(0,'.')
import bench, numpy as np
from scaffold.builder import Builder
from gsdf_amd import hip
hip.init(0)
sh=Builder().Scene("npt-flange"); sdf=hip.SDF3HIP(sh); sdf.specialize()
print("without the completion flag:", json.dumps(bench.evaluate_dropin(hip,sdf,sh)))
PY
