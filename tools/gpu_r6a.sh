#!/bin/bash
# round 6, first pass: the self-launching N > 1 bench on one GPU, the default line, the copy yardstick, the other line kinds
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
T=${1:-r6a}
timeout 900 python -m pytest tests/test_gpu_gather.py -q -x -k "self_launches" 2>&1 | tail -15
timeout 300 tools/ubench/copy_rate > gpurun_out/${T}_copy_rate.txt 2>&1; tail -62 gpurun_out/${T}_copy_rate.txt | sort -k9 -n -r | head -8
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; python - <<P
import json
d=json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","triangles_per_s")}); print(json.dumps(d["roofline"])[:1500]); print(json.dumps(d.get("roofline_march"))[:900])
P
timeout 600 python bench.py --steps 10 --warmup 2 --no-mesh-pipeline --no-cpu-baseline --no-one-shot --no-evaluate-dropin > gpurun_out/${T}_nopipe.json 2>>gpurun_out/${T}_bench.err; cut -c1-700 gpurun_out/${T}_nopipe.json
timeout 600 python bench.py --renderer dualcontour --scene npt-flange --resdiv 800 --steps 10 --warmup 2 > gpurun_out/${T}_dc.json 2>>gpurun_out/${T}_bench.err; python - <<P
import json
d=json.loads(open("gpurun_out/${T}_dc.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], json.dumps(d["roofline"])[:900])
P
timeout 600 python bench.py --mode flat --steps 5 --warmup 1 > gpurun_out/${T}_flat.json 2>>gpurun_out/${T}_bench.err; cut -c1-1500 gpurun_out/${T}_flat.json
tail -5 gpurun_out/${T}_bench.err
