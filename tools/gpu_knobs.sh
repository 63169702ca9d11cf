#!/bin/bash
# round 6: the grid-size knobs again, now that three meshes are in flight and the marching kernel is a different one
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { python bench.py --steps 30 --no-cpu-baseline --no-evaluate-dropin --no-distinct-rows --no-one-shot 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; m=d['roofline_march']
print('$1', 'ms/step', round(d['ms_per_step'],4), 'eval alone', round(r['kernel_ms'],4), 'march', round(m['kernel_ms'],4), 'device alone', round(r['ms_per_mesh_device_alone'],4))"; }
run "default"
for v in 3 5 6; do GSDF_HIP_MARCH_BPC=$v run "MARCH_BPC=$v"; done
for v in 16 24 48 64; do GSDF_HIP_LEAF_BPC=$v run "LEAF_BPC=$v"; done
for v in 1 3 4; do GSDF_HIP_PRUNE_BPC=$v run "PRUNE_BPC=$v"; done
run "default"
