# the three single-GPU configs through bench.py (pipelined), short
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-cfg}
mkdir -p $OUT
for sc in "npt-flange 1600" "bolt 2000" "knurled-cylinder 2000"; do set -- $sc
timeout 600 python bench.py --scene $1 --resdiv $2 --steps 20 --no-cpu-baseline --no-evaluate-dropin --no-one-shot --no-batch-throughput > $OUT/bench_$1.json 2>/dev/null
python - $OUT/bench_$1.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['config']['workload'][:40], round(d['ms_per_step'],4), 'ms/step', '%.3g evals/s'%d['value'], 'alone: kernel', round(d['roofline']['alone']['kernel_ms'],4), 'device', round(d['roofline']['alone']['ms_per_mesh_device'],4))
PY
done
