#!/bin/bash
# Developer experiments on the leaf phase: same bench command under different env knobs
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() {  # label, env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline ${SCENE_ARGS} 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; ph=d['phase_ms_rank0']
print('$label', 'ms/step %.3f' % d['ms_per_step'], 'kernel %.3f ms' % r['kernel_ms'], r['kernel'], 'phases', {k: round(v,3) for k,v in ph.items()}, 'tris', int(d['triangles_per_step']), d['config']['evaluator'][:60])"
}
run base A=1
run base2 A=1
SCENE_ARGS="--scene bolt --resdiv 2000"
run bolt A=1
SCENE_ARGS="--scene knurled-cylinder --resdiv 2000"
run knurled A=1
