#!/bin/bash
# The smallest useful pass through the kernel instantiations the default configuration does not reach (two and one point per lane,
# tiny cube queues, the fused leaf kernel): the scene parity tests (all three share_corners settings) and the random-tree fuzz.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for k in 2 1; do echo "GSDF_HIP_BATCH_K=$k"; GSDF_HIP_BATCH_K=$k timeout 900 python -m pytest tests/test_gpu_mesh.py tests/test_gpu_fuzz.py -m gpu -q -x -k "identical_to_oracle or random_trees" 2>&1 | tail -1; done
echo "GSDF_HIP_QCAP_MIN=4096"; GSDF_HIP_QCAP_MIN=4096 timeout 600 python -m pytest tests/test_gpu_mesh.py -m gpu -q -x -k "identical_to_oracle" 2>&1 | tail -1
echo "GSDF_HIP_FUSED_LEAF=1"; GSDF_HIP_FUSED_LEAF=1 timeout 600 python -m pytest tests/test_gpu_mesh.py tests/test_gpu_fuzz.py -m gpu -q -x -k "identical_to_oracle or random_trees" 2>&1 | tail -1
