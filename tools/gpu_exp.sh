cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for sc in bolt knurled-cylinder; do
for K in 2 4; do
GSDF_HIP_BATCH_K=$K timeout 600 python bench.py --scene $sc --resdiv 2000 --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$sc@2000 K=$K', 'Gevals/s', round(d['value']/1e9,2), 'ms', round(d['ms_per_step'],2), 'Mtris', d['triangles_per_step']/1e6, d['phase_ms_rank0']['leaf'])"
done
done
