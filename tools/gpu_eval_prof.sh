#!/bin/bash
# Evaluate micro-benchmark under rocprofv3: kernel stats + HBM traffic of the position/distance streams.
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-eval}
rm -rf $OUT; mkdir -p $OUT
for sc in sphere npt-flange; do
  B="python $GRAFT_REPO_ROOT/bench.py --mode eval --scene $sc --steps 10 --warmup 2 --no-cpu-baseline"
  $B > $OUT/${sc}_bench.json 2> $OUT/${sc}_bench.err
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${sc}_trace -- $B > $OUT/${sc}_trace.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${sc}_fetch -- $B > $OUT/${sc}_fetch.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${sc}_write -- $B > $OUT/${sc}_write.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - "$OUT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
res = {}
for sc in ("sphere", "npt-flange"):
    b = json.loads(open(f"{out}/{sc}_bench.json").read().strip().splitlines()[-1])
    def avg(d, name):
        vals = []
        for f in glob.glob(f"{out}/{sc}_{d}/**/*counter_collection.csv", recursive=True):
            per = {}
            for r in csv.DictReader(open(f)):
                if "eval_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name:
                    per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
            vals += list(per.values())
        return sum(vals) / len(vals) if vals else None
    ks = None
    for f in glob.glob(f"{out}/{sc}_trace/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "eval_kernel" in r["Name"]:
                ks = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"])}
    fetch, write = avg("fetch", "FETCH_SIZE"), avg("write", "WRITE_SIZE")
    n = int(b["config"]["workload"].split(",")[1].split()[0])
    res[sc] = {"bench": {k: b[k] for k in ("value", "ms_per_step")}, "roofline": b["roofline"], "kernel_stats": ks, "points_per_launch": n,
               "algorithmic_gb_per_launch": 16.0 * n / 1e9,
               "hbm_traffic_gb_per_launch": ((2 * fetch + write) * 1024 / 1e9) if fetch is not None and write is not None else None,
               "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write}
    if ks and res[sc]["hbm_traffic_gb_per_launch"]:
        res[sc]["hbm_gbps_from_counters"] = res[sc]["hbm_traffic_gb_per_launch"] / (ks["avg_ns"] * 1e-9)
open(out + "/eval_summary.json", "w").write(json.dumps(res, indent=1))
print(json.dumps(res, indent=1)[:2500])
PY
find $OUT -name "*kernel_trace.csv" -delete
