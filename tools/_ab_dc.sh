cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cp gsdf_amd/csrc/libgsdfhip.so /tmp/lib_orig.so
for v in dcbase dcxatomic; do
cp tools/variants/lib_$v.so gsdf_amd/csrc/libgsdfhip.so
for sc in text-plate npt-flange; do
timeout 300 python bench.py --interpreter --renderer dualcontour --scene $sc --resdiv 800 --steps 10 --warmup 2 --preheat 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $sc', round(d['ms_per_step'],4), {k:round(v['ms'],4) for k,v in d['stages'].items()})"
done
done
cp /tmp/lib_orig.so gsdf_amd/csrc/libgsdfhip.so
bash tools/gpu_dc_quick.sh
