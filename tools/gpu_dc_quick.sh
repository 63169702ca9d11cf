cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_mesh.py tests/test_gpu_fuzz.py -x -q -k "dual or dc" 2>&1 | tail -4
for sc in "text-plate 800" "npt-flange 800"; do set -- $sc
timeout 300 python bench.py --renderer dualcontour --scene $1 --resdiv $2 --steps 10 --warmup 2 --preheat 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), {k:round(v['ms'],4) for k,v in d['stages'].items()})"
done
