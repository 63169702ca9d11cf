#!/bin/bash
# Dual contouring (BASELINE configs[4]) under rocprofv3: bench line, per-kernel statistics and HBM / VALU counters of
# `bench.py --renderer dualcontour` on the reference-font text plate and on npt-flange at resdiv 800 -> gpurun_out/<tag>_dc_<scene>/
# (summaries are copied to profiles/ by hand).   usage: gpu_dc_prof.sh TAG
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r3}
for sc in "text-plate 800" "npt-flange 800"; do set -- $sc
  OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_dc_$1; rm -rf $OUT; mkdir -p $OUT
  ARGS="--renderer dualcontour --scene $1 --resdiv $2 --steps 5 --warmup 1 --preheat 3 --no-cpu-baseline"
  timeout 600 python bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err; cut -c1-400 $OUT/bench.json
  ( cd /tmp
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/trace.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/pmc_sq.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_wait -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/pmc_wait.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_BRANCH SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_mix -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/pmc_mix.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
    timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/pmc_write.log 2>&1 )
  python tools/pmc_summarize.py $OUT --command "python bench.py $ARGS" --workload "$(python -c "import json;print(json.load(open('$OUT/bench.json'))['config']['workload'])")" \
    --code "$(python -c "import json;print(json.load(open('$OUT/bench.json'))['config'].get('code') or '')")" > $OUT/pmc_summary.json
  cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
  find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
  head -12 $OUT/kernel_stats.csv | cut -c1-160
done
