#!/bin/bash
# one short bench line per single-GPU config (octree) + the two dual-contouring scenes: kernel alone, per mesh.   usage: gpu_quick3.sh ["pytest args"]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
if [ -n "$1" ]; then timeout 1500 python -m pytest $1 -x -q 2>&1 | tail -4; fi
for sc in "npt-flange 1600" "bolt 2000" "knurled-cylinder 2000"; do set -- $sc
timeout 600 python bench.py --scene $1 --resdiv $2 --steps 20 --no-cpu-baseline --no-evaluate-dropin --no-distinct-rows --no-one-shot 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; m=d['roofline_march']
print('$1', 'ms/step', round(d['ms_per_step'],4), 'tris', d['triangles_per_step'], 'eval alone', round(r['kernel_ms'],4), 'device alone', round(r['ms_per_mesh_device_alone'],4), 'march alone', round(m['kernel_ms'],4), 'of copy', round(m.get('frac_of_copy') or 0,3))"
done
for sc in "text-plate 800" "npt-flange 800"; do set -- $sc
timeout 300 python bench.py --renderer dualcontour --scene $1 --resdiv $2 --steps 10 --warmup 2 --preheat 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dc $1', round(d['ms_per_step'],4), {k:round(v['ms'],4) for k,v in d['stages'].items() if 'ms' in v})"
done
