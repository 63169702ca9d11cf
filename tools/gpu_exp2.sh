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
run w4_scratch GSDF_HIP_EXP_ALLOW_SCRATCH=1
run w2 GSDF_HIP_LEAF_WAVES=2
run bpc8 GSDF_HIP_LEAF_BPC=8
run bpc16 GSDF_HIP_LEAF_BPC=16
run bpc128 GSDF_HIP_LEAF_BPC=128

