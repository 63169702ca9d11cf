cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for w in 0 4; do
echo "LEAF_WAVES=$w"
GSDF_HIP_LEAF_WAVES=$w timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','triangles_per_s','phase_ms_rank0')}, d['roofline']['kernel_ms'])"
done
