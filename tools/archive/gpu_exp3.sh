#!/bin/bash
# Developer experiment: what the per-block triangle counts and the group-sum atomics cost in leaf_eval_kernel (timing only:
# the marching kernel has nothing to do without them)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() {  # label, env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline ${SCENE_ARGS} 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; ph=d['phase_ms_rank0']
print('$label', 'ms/step %.3f' % d['ms_per_step'], 'kernel %.3f ms' % r['kernel_ms'], r['kernel'], 'phases', {k: round(v,3) for k,v in ph.items()}, 'tris', int(d['triangles_per_step']))"
}
run base A=1
run no_psum GSDF_HIP_SPEC_FLAGS=-DGSDF_EXP_NO_PSUM
run no_ntri_no_psum "GSDF_HIP_SPEC_FLAGS=-DGSDF_EXP_NO_PSUM -DGSDF_EXP_NO_NTRI"
run base A=1
