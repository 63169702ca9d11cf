# brick masks (dev_ops.h: D_SKIP): parity tests that mesh through the octree, then the three single-GPU configs with and without masks
# bash tools/gpu_masks.sh <tag> ["pytest args"]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-masks}
mkdir -p $OUT
T="${2-tests/test_gpu_mesh.py tests/test_gpu_specialized.py tests/test_gpu_fuzz.py tests/test_gpu_prune_bounds.py}"
if [ -n "$T" ]; then timeout 2400 python -m pytest $T -x -q -m gpu 2>&1 | tail -15 | tee $OUT/tests.log; fi
for knob in 0 1; do
echo "GSDF_HIP_NO_BRICK_MASKS=$knob"
for sc in "npt-flange 1600" "bolt 2000" "knurled-cylinder 2000"; do set -- $sc
GSDF_HIP_NO_BRICK_MASKS=$knob timeout 600 python bench.py --scene $1 --resdiv $2 --steps 20 --no-cpu-baseline --no-evaluate-dropin --no-distinct-rows --no-one-shot > $OUT/bench_$1_$knob.json 2>$OUT/bench_$1_$knob.err || tail -5 $OUT/bench_$1_$knob.err
python - $OUT/bench_$1_$knob.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['config']['workload'][:40], round(d['ms_per_step'],4), 'ms/step', '%.3g evals/s'%d['value'], 'tris', d.get('triangles_per_step'), 'alone: kernel', round(d['roofline']['alone']['kernel_ms'],4), 'device', round(d['roofline']['alone']['ms_per_mesh_device'],4), d['phase_ms_rank0'])
PY
done
done
