#!/bin/bash
# prune phase against its grid size (GSDF_HIP_PRUNE_BPC = workgroups per CU of the per-level kernels), npt-flange resdiv 1600
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in 2 4 8 12 16 32; do
  GSDF_HIP_PRUNE_BPC=$v timeout 600 python bench.py --steps 20 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bpc=$v', d['ms_per_step'], d['phase_ms_rank0'])"
done
