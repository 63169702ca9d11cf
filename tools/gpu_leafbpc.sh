cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for sc in "bolt 2000" "knurled-cylinder 2000"; do set -- $sc
for v in 32 64; do
GSDF_HIP_LEAF_BPC=$v timeout 600 python bench.py --scene $1 --resdiv $2 --steps 20 --no-cpu-baseline --no-evaluate-dropin --no-batch-throughput 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1 LEAF_BPC=$v', round(d['ms_per_step'],4), 'ms/step  alone: kernel', round(d['roofline']['alone']['kernel_ms'],4), 'device', round(d['roofline']['alone']['ms_per_mesh_device'],4))"
done; done
