#!/bin/bash
# Extra PMC passes for the leaf kernel: instruction cache, branch count, instruction mix, scalar/LDS waits.
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc2}
mkdir -p $OUT
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT/p1 -- $BENCH > $OUT/p1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/p2 -- $BENCH > $OUT/p2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES --output-format csv -d $OUT/p3 -- $BENCH > $OUT/p3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/p4 -- $BENCH > $OUT/p4.log 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_summarize.py $OUT leaf_kernel | tee $OUT/summary.json
find $OUT -name "*kernel_trace.csv" -delete
tail -2 $OUT/p1.log
