#!/bin/bash
# N > 1 on ONE GPU (ranks over the ipc test transport): every rank's meshes pipelined three deep (bench.py: rank_pipeline) against one
# blocking mesh per step, both payloads, and with the pool as small as it was (GSDF_HIP_POOL_MAX=4); then RCCL at world size 1 and the
# gather tests.   bash tools/gpu_rank_pipeline.sh
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
P='import json,sys
l=[x for x in sys.stdin.read().splitlines() if x.startswith(chr(123))]
d=json.loads(l[-1]); print(d["n_gpus"], d["steps"], round(d["ms_per_step"],4), {m:round(v["ms_per_step"],3) for m,v in (d.get("gather_modes") or {}).items() if m!="note"}, d["triangles_per_step"], d["config"]["steps"][:60])'
for n in 2 3; do for a in "" "--no-mesh-pipeline" "--payload triangles" "POOL4"; do
  echo "== $n ranks $a"
  if [ "$a" = "POOL4" ]; then export GSDF_HIP_POOL_MAX=4; a=""; else unset GSDF_HIP_POOL_MAX; fi
  GSDF_HIP_COMM=ipc HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python bench.py --gpus $n --steps 200 --no-cpu-baseline $a 2>gpurun_out/rt.err | python -c "$P" || grep -i "fault\|error" gpurun_out/rt.err | head -3
done; done
unset GSDF_HIP_POOL_MAX
bash tools/gpu_dist1.sh 2>&1 | cut -c1-200 | tail -16
timeout 900 python -m pytest tests/test_gpu_gather.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
