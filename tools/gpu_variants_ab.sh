# A/B of library builds: tools/variants/lib_<name>.so copied over the shipped library one after the other; one bench line each.
# bash tools/gpu_variants_ab.sh "<names>" "<bench args>"
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cp gsdf_amd/csrc/libgsdfhip.so /tmp/lib_orig.so
for v in $1; do
cp tools/variants/lib_$v.so gsdf_amd/csrc/libgsdfhip.so
timeout 600 python bench.py ${2:---scene npt-flange --resdiv 1600} --steps 20 --no-cpu-baseline --no-evaluate-dropin --no-distinct-rows --no-one-shot > /tmp/b_$v.json 2>/tmp/b_$v.err || tail -3 /tmp/b_$v.err
python - /tmp/b_$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d['roofline']; m=d['roofline_march']
print(sys.argv[2], 'ms/step', round(d['ms_per_step'],4), 'tris', d['triangles_per_step'], 'eval alone', round(r['kernel_ms'],4), 'device alone', round(r['ms_per_mesh_device_alone'],4), 'march alone', round(m['alone']['kernel_ms'],4), 'march overlapped', round(m['kernel_ms'],4))
PY
done
cp /tmp/lib_orig.so gsdf_amd/csrc/libgsdfhip.so
