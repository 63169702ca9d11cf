#!/bin/bash
# dual contouring: per-kernel durations (rocprofv3 --kernel-trace --stats) of the two profiled scenes -> stdout
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for sc in "text-plate 800" "npt-flange 800"; do set -- $sc
  OUT=/tmp/dctrace_$1; rm -rf $OUT
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python $GRAFT_REPO_ROOT/bench.py --renderer dualcontour --scene $1 --resdiv $2 --steps 5 --warmup 1 --preheat 3 --no-cpu-baseline > /tmp/dctrace.log 2>&1)
  echo "== $1"; python - $(find $OUT -name "*kernel_stats.csv" | head -1) <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-28s calls %4s avg %9.1f us  min %9.1f" % (r["Name"].split("(")[0][-28:], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
done
