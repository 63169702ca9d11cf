#!/bin/bash
# Every kind of bench line at bench.py's DEFAULT --steps / --warmup, with the wall clock of each process: that the defaults finish in
# seconds on every mode (round 6: K = 1000).   bash tools/gpu_default_lines.sh [tag]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=${1:-r6q}
P='import json,sys,time
l=[x for x in sys.stdin.read().splitlines() if x.startswith(chr(123))]
d=json.loads(l[-1]); print(d["n_gpus"], d["steps"], d["warmup"], round(d["ms_per_step"],4), "%.4g"%d["value"], d["roofline"]["bound"], d["roofline"].get("frac"), {m:round(v["ms_per_step"],3) for m,v in (d.get("gather_modes") or {}).items() if m!="note"}, d["config"]["workload"][:60])'
run() { local t0=$(date +%s.%N); "$@" 2>gpurun_out/${T}_last.err | tee -a gpurun_out/${T}_default_lines.jsonl | python -c "$P"; echo "   wall $(python -c "import time;print(round(time.time()-$t0,1))") s: $*"; }
for i in 1 2 3; do run python bench.py; done
GSDF_HIP_COMM=ipc run python bench.py --gpus 2
run python bench.py --renderer dualcontour --scene text-plate --resdiv 800 --no-cpu-baseline
run python bench.py --mode flat --no-cpu-baseline
run python bench.py --mode eval --no-cpu-baseline
run python bench.py --scene knurled-cylinder --resdiv 2000 --no-cpu-baseline
run python bench.py --scene bolt --resdiv 2000 --no-cpu-baseline
