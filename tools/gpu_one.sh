# one test file / expression on the GPU box: bash tools/gpu_one.sh "<pytest args>"
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest $1 -x -q 2>&1 | tail -${2:-25}
