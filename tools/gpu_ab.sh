#!/bin/bash
# A/B of this round's switches on the bench line: both column passes in one body (npt-flange)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-ab}
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q --durations=25 > $OUT/pytest.log 2>&1; grep -E "passed|failed|Error" $OUT/pytest.log | tail -3
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], {k:round(v,4) for k,v in d['phase_ms_rank0'].items()}, d['roofline']['kernel'])"; }
for rep in 1 2; do
  GSDF_HIP_NO_BOTH_PASSES=1 timeout 600 python bench.py --steps 20 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | line "npt-flange one-pass-per-body"
  timeout 600 python bench.py --steps 20 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | line "npt-flange default"
done
timeout 600 python bench.py --scene knurled-cylinder --resdiv 2000 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | line knurled
timeout 600 python bench.py --scene bolt --resdiv 2000 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | line bolt
