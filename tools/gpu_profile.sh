#!/bin/bash
# Round 5: one profile pass per config -- bench line, rocprofv3 kernel statistics, PMC counters in separate passes (traffic, instruction
# classes, waits) summarised per kernel under the running kernels' code key -> gpurun_out/<tag>_<scene>/ (copy pmc_summary.json,
# kernel_stats.csv and bench.json to profiles/<tag>_<scene>_*).   bash tools/gpu_profile.sh <tag> "<scene> <resdiv>" ["<scene> <resdiv>" ...]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r5}; shift
[ $# -eq 0 ] && set -- "npt-flange 1600"
for sc in "$@"; do set -- $sc
  OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_$1; rm -rf $OUT; mkdir -p $OUT
  ARGS="--scene $1 --resdiv $2"
  timeout 900 python bench.py $ARGS --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err || tail -3 $OUT/bench.err
  # counters from blocking meshes (the kernel with the GPU to itself), as the line's roofline is
  P="python $GRAFT_REPO_ROOT/bench.py $ARGS --steps 5 --warmup 1 --preheat 4 --no-cpu-baseline --no-evaluate-dropin --no-distinct-rows --no-one-shot --no-mesh-pipeline"
  ( cd /tmp
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $P > $OUT/trace.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d $OUT/p1 -- $P > $OUT/p1.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $OUT/p2 -- $P > $OUT/p2.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 --output-format csv -d $OUT/p3 -- $P > $OUT/p3.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/p4 -- $P > $OUT/p4.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p5 -- $P > $OUT/p5.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/p6 -- $P > $OUT/p6.log 2>&1 )
  python - $OUT <<'PY'
import json, subprocess, sys
out = sys.argv[1]
b = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
s = subprocess.check_output([sys.executable, "tools/pmc_summarize.py", out, "--evals-per-launch", str(b["evals_per_step"]), "--kernel-ms", str(b["roofline"]["kernel_ms"]),
                             "--workload", b["config"]["workload"], "--code", b["config"].get("code") or "", "--command", "python bench.py <scene> --no-mesh-pipeline (tools/gpu_profile.sh)"])
open(out + "/pmc_summary.json", "wb").write(s)
j = json.loads(s)
for k in ("leaf_eval_kernel", "march_records_kernel", "prune_kernel"):
    if k in j:
        print(k, {n: (round(v, 3) if isinstance(v, float) and abs(v) < 1e4 else v) for n, v in j[k].items() if n in ("valu_lane_instr_per_eval", "valu_issue_frac_of_peak", "hbm_traffic_gb_per_launch", "wave_wait_any_frac", "valu_mix", "SQ_INSTS_VALU")})
print({k: b[k] for k in ("ms_per_step", "value")}, "kernel alone", b["roofline"]["kernel_ms"])
PY
  cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
  find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
  rm -rf $OUT/trace $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5 $OUT/p6
done
