#!/bin/bash
# After the statistics atomics left leaf_eval_kernel: grid size of the leaf kernel (GSDF_HIP_LEAF_BPC workgroups per CU), then
# the compiler scheduling options again (tools/gpu_schedflags.sh), npt-flange / bolt / knurled-cylinder bench lines.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$1 |', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['phase_ms_rank0'].items() if k in ('eval_kernel','march_kernel','total_device')}, d['roofline']['kernel'])
except Exception as e: print('$1 | failed', e)"; }
for bpc in 16 32 64 128 256; do
  for sc in "npt-flange 1600" "bolt 2000" "knurled-cylinder 2000"; do set -- $sc
    GSDF_HIP_LEAF_BPC=$bpc timeout 600 python bench.py --scene $1 --resdiv $2 --steps 20 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | line "leaf_bpc=$bpc $1"
  done
done
bash tools/gpu_schedflags.sh
