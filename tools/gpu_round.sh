#!/bin/bash
# Full GPU pass: parity tests, bench line, rocprofv3 kernel stats + PMC counters. Outputs under gpurun_out/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r1}
mkdir -p $OUT
if [ -z "$SKIP_TESTS" ]; then  # SKIP_TESTS=1 SCENE_ARGS="--scene bolt --resdiv 2000": counters of another config only
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
fi
timeout 900 python bench.py ${SCENE_ARGS} > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json
cd /tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-distinct-rows ${SCENE_ARGS}"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_wait -- $BENCH > $OUT/pmc_wait.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 --output-format csv -d $OUT/pmc_mix -- $BENCH > $OUT/pmc_mix.log 2>&1
cd $GRAFT_REPO_ROOT
python - "$OUT" <<'PY'
import json, subprocess, sys
out = sys.argv[1]
b = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
ev = b["evals_per_step"] - 0  # leaf + prune evaluations; prune share is < 1 %
kms = (b["roofline"].get("alone") or b["roofline"])["kernel_ms"]  # the kernel with the GPU to itself (the timed loop keeps two meshes in flight)
s = subprocess.check_output([sys.executable, "tools/pmc_summarize.py", out, "--evals-per-launch", str(ev), "--kernel-ms", str(kms),
                             "--workload", b["config"]["workload"], "--code", b["config"].get("code") or ""])
open(out + "/pmc_summary.json", "wb").write(s)
print(s.decode()[:1500])
PY
find $OUT -name "*stats.csv" | head
# keep only the summaries (gpurun_out's merge limit is 64 MiB for everything a call writes): the per-dispatch CSVs are what
# pmc_summary.json and *_kernel_stats.csv were made from
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*agent_info.csv" -delete
du -sh $OUT
