#!/bin/bash
# The octree's prune phase: GPU tests that exercise it (default: speculative top + resolve; GSDF_HIP_PRUNE_SPEC=0: one launch per
# level), then the bench line's phase times both ways.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-prune}
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_mesh.py tests/test_gpu_prune_bounds.py tests/test_gpu_specialized.py tests/test_gpu_fuzz.py tests/test_gpu_capi_replay.py -m gpu -x -q > $OUT/pytest_spec.log 2>&1; grep -E "passed|failed" $OUT/pytest_spec.log | tail -2
GSDF_HIP_PRUNE_SPEC=0 timeout 1500 python -m pytest tests/test_gpu_mesh.py tests/test_gpu_prune_bounds.py -m gpu -x -q -k "not full_size" > $OUT/pytest_chain.log 2>&1; grep -E "passed|failed" $OUT/pytest_chain.log | tail -2
for v in 1 0; do
  GSDF_HIP_PRUNE_SPEC=$v timeout 600 python bench.py --steps 20 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('spec=$v', {k:d[k] for k in ('value','ms_per_step','phase_ms_rank0')})"
done
for sc in "bolt 2000" "knurled-cylinder 2000"; do set -- $sc
  timeout 600 python bench.py --scene $1 --resdiv $2 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', {k:d[k] for k in ('value','ms_per_step','triangles_per_step','phase_ms_rank0')})"
done
