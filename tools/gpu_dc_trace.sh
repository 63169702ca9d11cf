# dual contouring: per-kernel durations (rocprofv3 --kernel-trace --stats) of the two profiled scenes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for sc in "npt-flange 800" "text-plate 800"; do set -- $sc
OUT=$GRAFT_REPO_ROOT/gpurun_out/dctrace_$1; rm -rf $OUT; mkdir -p $OUT
( cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $GRAFT_REPO_ROOT/bench.py --renderer dualcontour --scene $1 --resdiv $2 --steps 5 --warmup 1 --preheat 3 --no-cpu-baseline > $OUT/trace.log 2>&1 )
python - $(find $OUT/trace -name "*kernel_stats.csv" | head -1) <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(r["Name"][:60].ljust(60), r["Calls"], "avg us", round(float(r["AverageNs"])/1e3,1), "min", round(float(r["MinNs"])/1e3,1))
PY
rm -rf $OUT/trace
done
