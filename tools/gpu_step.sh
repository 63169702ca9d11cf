# bash tools/gpu_step.sh <tag> "<pytest args or empty>" "<bench args>"
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-step}
mkdir -p $OUT
if [ -n "$2" ]; then timeout 2400 python -m pytest $2 -x -q 2>&1 | tail -15; fi
timeout 900 python bench.py $3 > $OUT/bench.json 2> $OUT/bench.err; tail -3 $OUT/bench.err; python - $OUT/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','triangles_per_s','evaluator_build')}); print(d['phase_ms_rank0'])
e=d.get('evaluate_dropin')
if e: print({k:(round(v["us_per_call"],1), round(v["evals_per_s"]/1e9,3)) for k,v in e.items() if isinstance(v,dict)})
print('roofline', round(d['roofline']['frac'],3), d['roofline_march'] and round(d['roofline_march']['frac'],3), 'batch', d.get('batch_throughput',{}).get('ms_per_mesh'))
PY
