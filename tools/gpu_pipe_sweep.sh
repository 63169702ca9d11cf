# pipelined meshes per second against the occupancy knobs of the kernels that run beside the evaluating kernel
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { echo "$*"; env "$@" timeout 300 python tools/gpu_pipe.py 80 2>/dev/null | grep -E "blocking|two in flight" | tail -2; }
run GSDF_HIP_PRUNE_BPC=2 GSDF_HIP_LEAF_BPC=32 GSDF_HIP_MARCH_BPC=4
run GSDF_HIP_PRUNE_BPC=2 GSDF_HIP_LEAF_BPC=32 GSDF_HIP_MARCH_BPC=7
run GSDF_HIP_PRUNE_BPC=1 GSDF_HIP_LEAF_BPC=32 GSDF_HIP_MARCH_BPC=4
run GSDF_HIP_PRUNE_BPC=2 GSDF_HIP_LEAF_BPC=24 GSDF_HIP_MARCH_BPC=4
run GSDF_HIP_PRUNE_BPC=2 GSDF_HIP_LEAF_BPC=48 GSDF_HIP_MARCH_BPC=4
run GSDF_HIP_PRUNE_BPC=3 GSDF_HIP_LEAF_BPC=32 GSDF_HIP_MARCH_BPC=5
