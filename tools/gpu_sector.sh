#!/bin/bash
# the circular arrays' sector gate (D_CIRC_ORDER / D_GATEOB): parity tests, then knurled-cylinder's bench line and kernel statistics
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-sector}
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; grep -E "passed|failed|Error" $OUT/pytest.log | tail -3
GSDF_HIP_NO_SECTOR_GATE=1 timeout 600 python bench.py --scene knurled-cylinder --resdiv 2000 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('knurled, no sector gate', {k:d[k] for k in ('value','ms_per_step','phase_ms_rank0')})"
for sc in "knurled-cylinder 2000" "npt-flange 1600" "bolt 2000"; do set -- $sc
  timeout 600 python bench.py --scene $1 --resdiv $2 --steps 10 --warmup 2 --no-cpu-baseline 2>$OUT/$1.err | tail -1 > $OUT/$1_bench.json
  python -c "
import json,sys; d=json.loads(open('$OUT/$1_bench.json').read()); print('$1', {k:d[k] for k in ('value','ms_per_step','triangles_per_step','phase_ms_rank0')}, d['roofline']['kernel'])"
done
