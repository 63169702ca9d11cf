cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for cfg in "4 2" "4 3" "2 3" "2 4" "1 4"; do
set -- $cfg
GSDF_HIP_BATCH_K=$1 GSDF_HIP_LEAF_WAVES=$2 timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('K W = $cfg', round(d['ms_per_step'],3), d['phase_ms_rank0'])"
done
