# the GPU suite + smoke, log under gpurun_out/<tag>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-tests}
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q ${PYTEST_ARGS:--x} > $OUT/pytest_gpu.log 2>&1; tail -25 $OUT/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
