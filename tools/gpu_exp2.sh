#!/bin/bash
# Developer experiments on the specialised leaf kernel: where does its time go? (env knobs, same bench command)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() {  # label, env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline ${SCENE_ARGS} 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$label', 'ms/step %.3f' % d['ms_per_step'], 'kernel %.3f ms' % r['kernel_ms'], r['kernel'], 'tris', int(d['triangles_per_step']))"
}
run base A=1
run no_emit GSDF_HIP_SPEC_FLAGS=-DGSDF_EXP_NO_EMIT
run no_flush GSDF_HIP_SPEC_FLAGS=-DGSDF_EXP_NO_FLUSH
run no_build GSDF_HIP_SPEC_FLAGS=-DGSDF_EXP_NO_BUILD
run w3 GSDF_HIP_LEAF_WAVES=3
SCENE_ARGS="--scene bolt --resdiv 2000"
run bolt A=1
SCENE_ARGS="--scene knurled-cylinder --resdiv 2000"
run knurled A=1
