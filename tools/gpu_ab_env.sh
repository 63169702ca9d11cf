# A/B of an environment knob over the three single-GPU configs: bash tools/gpu_ab_env.sh "KNOB=1" ["KNOB2=.."]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for sc in "npt-flange 1600" "bolt 2000" "knurled-cylinder 2000"; do set -- $sc
for e in "X=0" "$AB"; do
env $e timeout 600 python bench.py --scene $1 --resdiv $2 --steps 20 --no-cpu-baseline --no-evaluate-dropin 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 [$e]', round(d['ms_per_step'],4), 'ms/step  alone: kernel', round(d['roofline']['alone']['kernel_ms'],4), 'device', round(d['roofline']['alone']['ms_per_mesh_device'],4), d['roofline']['kernel'][:40])"
done; done
