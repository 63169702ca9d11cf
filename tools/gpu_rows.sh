#!/bin/bash
# Distinct z rows (gsdf_mesh_opts.share_corners = 2): the parity tests that exercise it, then the bench line (its distinct_rows
# section) for the three single-GPU configs.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-rows}
mkdir -p $OUT
if [ -z "$SKIP_TESTS" ]; then
timeout 1500 python -m pytest tests/test_gpu_mesh.py tests/test_gpu_specialized.py -m gpu -x -q -k "identical or full_size" > $OUT/pytest_rows.log 2>&1; tail -4 $OUT/pytest_rows.log
fi
for sc in ${SCENES:-npt-flange:1600 bolt:2000 knurled-cylinder:2000}; do set -- ${sc/:/ }
  GSDF_HIP_DEBUG=1 timeout 600 python bench.py --scene $1 --resdiv $2 --steps 20 --warmup 2 --no-cpu-baseline --no-evaluate-dropin 2>$OUT/err_$1.txt | tail -1 > $OUT/bench_$1.json
  grep "rows\|true, true>" $OUT/err_$1.txt | tail -6
  python - $OUT/bench_$1.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read())
print(d['config']['workload'][:40], 'ms/mesh', round(d['ms_per_step'],4), 'alone', {k:round(v,4) for k,v in d['roofline']['alone'].items() if isinstance(v,float)})
for nm in ('distinct_rows','distinct_points'):
  r=d.get(nm)
  if r: print('  '+nm+': ms/mesh', round(r['ms_per_step'],4), 'evals performed', r['evals_performed_per_step'], 'of', d['evals_per_step'], 'alone', r['alone'], r['kernel'])
PY
done
