#!/bin/bash
# The literal drop-in (32 768-point host-buffer Evaluate calls) with 4, 2 and 1 points per lane in the evaluating kernel
# (GSDF_HIP_BATCH_K): fewer points per lane = more workgroups reading the host buffer across PCIe at the same time.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for k in 4 2 1; do
  GSDF_HIP_BATCH_K=$k timeout 600 python bench.py --steps 3 --warmup 1 --preheat 5 --no-cpu-baseline --no-distinct-rows 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); e=d['evaluate_dropin']; print('K=$k', {k:(round(v['us_per_call'],1), round(v['evals_per_s']/1e9,3)) for k,v in e.items() if isinstance(v,dict)})"
done
