#!/bin/bash
# FlatRenderer on device: bench line, rocprofv3 kernel stats, HBM traffic counters. Outputs under gpurun_out/<name>.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-flat}
mkdir -p $OUT
timeout 600 python bench.py --mode flat > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json
cd /tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --mode flat --steps 5 --warmup 1"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summarize.py $OUT --command "python bench.py --mode flat --steps 5 --warmup 1" --workload "FlatRenderer on device, npt-flange resdiv 1600" > $OUT/pmc_summary.json
cat $OUT/pmc_summary.json | head -60
find $OUT -name "*kernel_stats.csv" | head -1 | xargs head -8
find $OUT -name "*kernel_trace.csv" -size +4M -delete
find $OUT -name "*counter_collection.csv" -size +16M -delete
du -sh $OUT
