#!/bin/bash
# bench.py's multi-GPU code path (gloo control plane + RCCL gather inside the library; then its torch.distributed fallback)
# at world size 1 -- what one GPU allows: catches everything but the cross-rank behaviour itself
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for fb in "" 1; do
  echo "== force_torch_gather='$fb'"
  GSDF_BENCH_FORCE_DIST=1 GSDF_BENCH_FORCE_TORCH_GATHER=$fb timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -3 | cut -c1-700
done
