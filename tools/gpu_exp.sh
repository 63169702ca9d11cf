cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for sc in sphere npt-flange bolt knurled-cylinder; do
timeout 300 python bench.py --mode eval --scene $sc --steps 20 --warmup 3 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$sc', 'Gevals/s', round(d['value']/1e9,2), 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'HBM frac', round(d['roofline']['frac'],4))"
done
for sc in bolt knurled-cylinder; do
timeout 600 python bench.py --scene $sc --resdiv 2000 --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$sc@2000', 'Gevals/s', round(d['value']/1e9,2), 'ms', round(d['ms_per_step'],2), 'Mtris', d['triangles_per_step']/1e6, 'Mevals', d['evals_per_step']/1e6, d['phase_ms_rank0'])"
done
timeout 600 python bench.py --share-corners --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('flange shared', 'Gevals/s', round(d['value']/1e9,2), 'ms', round(d['ms_per_step'],2), 'Gtris/s', round(d['triangles_per_s']/1e9,3))"
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('flange default', 'Gevals/s', round(d['value']/1e9,2), 'ms', round(d['ms_per_step'],2), 'Gtris/s', round(d['triangles_per_s']/1e9,3))"
