#!/bin/bash
# Round 6, last pass on the final commit: the GPU suite's log, then the bench lines again now that profiles/ holds the PMC summaries of
# THIS build (roofline.counters: "its code key equals the running kernels'") -> gpurun_out/<tag>_*
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=${1:-r6k}
if [ "$2" != "nosuite" ]; then python -m pytest tests -m gpu -q > gpurun_out/${T}_gpu_suite.log 2>&1; grep -E "passed|failed|error" gpurun_out/${T}_gpu_suite.log | tail -3; fi
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/${T}_default_bench.json 2> gpurun_out/${T}_default_bench.err
for sc in "npt-flange 1600" "bolt 2000" "knurled-cylinder 2000"; do set -- $sc
  timeout 900 python bench.py --scene $1 --resdiv $2 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 > gpurun_out/${T}_$1_bench.json; done
for sc in "text-plate 800" "npt-flange 800"; do set -- $sc
  timeout 600 python bench.py --renderer dualcontour --scene $1 --resdiv $2 --steps 20 --warmup 2 --preheat 10 --no-cpu-baseline 2>/dev/null | grep "^{" | tail -1 > gpurun_out/${T}_dc_$1_bench.json; done
python - <<P
import json,glob
for f in sorted(glob.glob("gpurun_out/${T}_*bench.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f.split("/")[-1], round(d["ms_per_step"],4), "%.3g"%d["value"], r["bound"], r["kernel"][:28], round(r["kernel_ms"],4), r["frac"] and round(r["frac"],3), (r.get("valu") or {}).get("code_match"), (r.get("counters") or "")[:60])
P
