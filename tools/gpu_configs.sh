#!/bin/bash
# bench line + rocprofv3 kernel stats for the other BASELINE configs on one GPU (bolt, knurled-cylinder at resdiv 2000)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-configs}
mkdir -p $OUT
for sc in bolt knurled-cylinder; do
  timeout 600 python bench.py --scene $sc --resdiv 2000 --no-cpu-baseline > $OUT/${sc}_bench.json 2> $OUT/${sc}_bench.err
  cat $OUT/${sc}_bench.json | cut -c1-400
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${sc}_trace -- python $GRAFT_REPO_ROOT/bench.py --scene $sc --resdiv 2000 --steps 5 --warmup 1 --preheat 5 --no-cpu-baseline > $OUT/${sc}_trace.log 2>&1)
  find $OUT/${sc}_trace -name "*kernel_stats.csv" | head -1 | xargs head -4 | cut -c1-200
  find $OUT/${sc}_trace -name "*kernel_trace.csv" -size +4M -delete
done
