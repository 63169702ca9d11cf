#!/bin/bash
# Round 6: every measurement DESIGN.md / README.md quote, from ONE build, into gpurun_out/<tag>_* (copied to profiles/ by hand):
# default bench line, smoke, profile passes (kernel statistics + PMC) of the three single-GPU configs, dual contouring, the Evaluate
# micro-benchmark, the copy yardstick, the N > 1 path on one GPU (self-launched ranks over the ipc transport; RCCL at world size 1).
#   bash tools/gpu_evidence.sh <tag> [parts: bench prof dc eval dist]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=${1:-r6}; shift; PARTS="${*:-bench prof dc eval dist}"
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has bench; then
  timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/${T}_default_bench.json 2> gpurun_out/${T}_default_bench.err
  python -c "
import json; d=json.loads(open('gpurun_out/${T}_default_bench.json').read().strip().splitlines()[-1])
print('default', {k:d[k] for k in ('value','ms_per_step','triangles_per_s')}, d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline_march']['kernel_ms'], d['roofline_march']['frac'], d['roofline_march']['frac_of_copy'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], 'one_shot', d['one_shot']['ms'], 'dropin', d['evaluate_dropin']['blocking_pageable'])"
  timeout 600 python __graft_entry__.py smoke > gpurun_out/${T}_smoke.log 2>&1; tail -1 gpurun_out/${T}_smoke.log
  timeout 300 tools/ubench/copy_rate > gpurun_out/${T}_copy_rate.txt 2>&1
  for a in "--no-mesh-pipeline" "--mesh-depth 2" "--mode flat"; do timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-one-shot --no-evaluate-dropin --no-distinct-rows $a 2>/dev/null | tail -1 >> gpurun_out/${T}_other_lines.jsonl; done
fi
if has prof; then bash tools/gpu_profile.sh $T "npt-flange 1600" "bolt 2000" "knurled-cylinder 2000" 2>&1 | tail -12; fi
if has dc; then bash tools/gpu_dc_prof.sh $T 2>&1 | grep -v "^\"" | tail -6; fi
if has eval; then bash tools/gpu_eval_prof.sh ${T}_eval 2>&1 | tail -5; fi
if has dist; then
  for n in 2 3; do for args in "" "--payload triangles"; do
    echo "== $n ranks on one GPU (self-launched, ipc transport) $args"
    GSDF_HIP_COMM=ipc HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python bench.py --gpus $n --steps 10 --warmup 2 --preheat 10 --no-cpu-baseline $args 2>gpurun_out/${T}_dist${n}_ipc.err | grep "^{" | tail -1 | tee -a gpurun_out/${T}_dist${n}_ipc.log | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('n_gpus','value','ms_per_step','triangles_per_step')}, d['roofline']['bound'], {m:(round(v['ms_per_step'],3), v['bytes_received_per_rank']) for m,v in d['gather_modes'].items() if m!='note'})
except Exception as e: print('FAILED', e)"
  done; done
  bash tools/gpu_dist1.sh 2>&1 | tee gpurun_out/${T}_dist1_world1.log | cut -c1-260 | tail -16
fi
