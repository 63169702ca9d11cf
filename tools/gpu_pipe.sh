cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python tools/gpu_pipe.py 60
