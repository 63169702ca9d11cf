# pipelined meshes per second against the occupancy knobs of the kernels that run beside the evaluating kernel
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in 7 5 4 3; do echo "GSDF_HIP_MARCH_BPC=$v"; GSDF_HIP_MARCH_BPC=$v timeout 300 python tools/gpu_pipe.py 80 2>/dev/null | grep -E "blocking|two in flight"; done
for v in 2 8; do echo "GSDF_HIP_PRUNE_BPC=$v"; GSDF_HIP_PRUNE_BPC=$v timeout 300 python tools/gpu_pipe.py 80 2>/dev/null | grep -E "blocking|two in flight"; done
for v in 32 16; do echo "GSDF_HIP_LEAF_BPC=$v"; GSDF_HIP_LEAF_BPC=$v timeout 300 python tools/gpu_pipe.py 80 2>/dev/null | grep -E "blocking|two in flight"; done
