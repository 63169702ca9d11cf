#!/bin/bash
# Reduced tools/gpu_variants.sh: the mesh + fuzz tests through the kernel instantiations the default configuration does not
# reach (K = 2, 1 column bricks; tiny cube queues; the fused leaf kernel)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for k in 2 1; do echo "GSDF_HIP_BATCH_K=$k"; GSDF_HIP_BATCH_K=$k timeout 900 python -m pytest tests/test_gpu_mesh.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -1; done
echo "GSDF_HIP_QCAP_MIN=4096"; GSDF_HIP_QCAP_MIN=4096 timeout 600 python -m pytest tests/test_gpu_mesh.py -m gpu -q -x 2>&1 | tail -1
echo "GSDF_HIP_FUSED_LEAF=1"; GSDF_HIP_FUSED_LEAF=1 timeout 600 python -m pytest tests/test_gpu_mesh.py -m gpu -q -x 2>&1 | tail -1
