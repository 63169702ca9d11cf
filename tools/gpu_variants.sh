#!/bin/bash
# The -m gpu suite through kernel instantiations the default configuration does not reach: forced points per lane
# (GSDF_HIP_BATCH_K=2/1), forced workgroups per CU of the leaf kernel, tiny cube queues (overflow -> grow -> rerun).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for k in 2 1; do echo "GSDF_HIP_BATCH_K=$k"; GSDF_HIP_BATCH_K=$k timeout 1100 python -m pytest tests -m gpu -q 2>&1 | tail -1; done
for w in 2 3; do echo "GSDF_HIP_LEAF_WAVES=$w"; GSDF_HIP_LEAF_WAVES=$w timeout 600 python -m pytest tests/test_gpu_mesh.py tests/test_gpu_specialized.py -m gpu -q 2>&1 | tail -1; done
echo "GSDF_HIP_QCAP_MIN=4096"; GSDF_HIP_QCAP_MIN=4096 timeout 600 python -m pytest tests/test_gpu_mesh.py -m gpu -q 2>&1 | tail -1
