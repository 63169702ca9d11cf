cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','triangles_per_s','phase_ms_rank0')}, d['roofline']['kernel_ms'])"
