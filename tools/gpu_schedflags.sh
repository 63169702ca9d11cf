#!/bin/bash
# Compiler scheduling options for the specialised kernels (GSDF_HIP_SPEC_FLAGS), timing only: npt-flange and bolt bench lines.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$1 |', d['ms_per_step'], {k:round(v,4) for k,v in d['phase_ms_rank0'].items() if k in ('eval_kernel','total_device')}, d['roofline']['kernel'])
except Exception as e: print('$1 | failed', e)"; }
for f in "" "-mllvm -amdgpu-sched-strategy=max-ilp" "-mllvm -amdgpu-sched-strategy=max-memory-clause" "-mllvm -amdgpu-sched-strategy=iterative-ilp" "-mllvm -amdgpu-sched-strategy=iterative-minreg" "-mllvm -amdgpu-schedule-metric-bias=0" "-mllvm -amdgpu-schedule-metric-bias=100" "-mllvm -enable-post-misched=0" "-mllvm -misched-postra-direction=bottomup" "-O2" "-mllvm -amdgpu-disable-clustered-low-occupancy-reschedule"; do
  for sc in "npt-flange 1600" "bolt 2000"; do set -- $sc
    GSDF_HIP_SPEC_FLAGS="$f" timeout 600 python bench.py --scene $1 --resdiv $2 --steps 20 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | line "[$f] $1"
  done
done
