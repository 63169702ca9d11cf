cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6i_interp; rm -rf $OUT; mkdir -p $OUT
P="python $GRAFT_REPO_ROOT/bench.py --interpreter --steps 5 --warmup 1 --preheat 4 --no-cpu-baseline --no-evaluate-dropin --no-distinct-rows --no-one-shot --no-mesh-pipeline"
$P 2>/dev/null | grep "^{" | tail -1 > $OUT/bench.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $P > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d $OUT/p1 -- $P > $OUT/p1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $OUT/p2 -- $P > $OUT/p2.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summarize.py $OUT --evals-per-launch 151903801 --workload interp > $OUT/pmc_summary.json
python - <<P
import json
j=json.load(open("$OUT/pmc_summary.json")); b=json.loads(open("$OUT/bench.json").read())
print(b["ms_per_step"], b["roofline"]["kernel"], b["roofline"]["kernel_ms"])
for k in ("leaf_eval_kernel","prune_kernel","march_records_kernel"):
    if k in j: print(k, {n:(round(v,3) if isinstance(v,float) else v) for n,v in j[k].items() if not isinstance(v,dict)})
P
head -8 $(find $OUT/trace -name "*kernel_stats.csv" | head -1) | cut -c1-150
