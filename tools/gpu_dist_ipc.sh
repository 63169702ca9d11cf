#!/bin/bash
# bench.py --gpus N as the driver launches it (torch.distributed.run, one process per rank), on ONE GPU: the ranks share device 0 over
# the library's inter-process transport (GSDF_HIP_COMM=ipc). The N > 1 code path end to end -- gloo rendezvous of the id, per-process
# HIP contexts, barriers, counts, plan, transfers, marching after the gather, the N > 1 JSON line -- not a scaling measurement.
# bash tools/gpu_dist_ipc.sh <tag> [N ...]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r5}; shift
[ $# -eq 0 ] && set -- 2 3
for n in "$@"; do for args in "--gather all" "--gather all --payload triangles" "--gather root" "--gather none"; do
echo "== $n ranks on one GPU, $args"
GSDF_HIP_COMM=ipc HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n \
  bench.py --gpus $n --steps 10 --warmup 2 --preheat 10 --no-cpu-baseline $args 2>gpurun_out/${TAG}_dist${n}_ipc.err | tail -1 | tee -a gpurun_out/${TAG}_dist${n}_ipc.log | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('n_gpus','value','ms_per_step','triangles_per_step','gather')}, d['config'].get('devices'), d['config']['sharding'][:160])
except Exception as e: print('FAILED', e)"
tail -3 gpurun_out/${TAG}_dist${n}_ipc.err | cut -c1-300
done; done
