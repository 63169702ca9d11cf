#!/bin/bash
# bench.py's multi-GPU code path (gloo control plane + RCCL gather inside the library, each gather mode, pipelined and not; then
# its torch.distributed fallback) at world size 1 -- what one GPU allows: catches everything but the cross-rank behaviour itself
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() {
  GSDF_BENCH_FORCE_DIST=1 GSDF_BENCH_FORCE_TORCH_GATHER=$FB timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>&1 | grep "^{" | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('value','ms_per_step','phase_ms_rank0','gather')}, d['config']['sharding'])
except Exception as e: print('FAILED', e)"
}
FB="" ; for m in all root none; do echo "== gather $m"; run --gather $m; echo "== gather $m, not pipelined"; run --gather $m --no-gather-pipeline; done
echo "== gather all, payload triangles"; run --gather all --payload triangles
FB=1; echo "== torch.distributed fallback"; run
