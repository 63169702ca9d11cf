#!/bin/bash
# Developer experiment: march_records_kernel without its output stream / with a third of its record traffic (timing only;
# the library is rebuilt on the box with the experiment's -D and restored afterwards)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
build() { (cd gsdf_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -Wno-unused-function -I$GRAFT_REPO_ROOT/include $1 gsdf_hip.hip compile.cpp specialize.cpp -lhiprtc -ldl -o libgsdfhip.so 2>&1 | grep -E "error" ); }
run() {
  timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); ph=d['phase_ms_rank0']
print('$1', 'ms/step %.3f' % d['ms_per_step'], {k: round(v,3) for k,v in ph.items()}, 'tris', int(d['triangles_per_step']))"
}
run base
build -DGSDF_EXP_MARCH_NO_STORE; run no_store
build -DGSDF_EXP_MARCH_ONE_LINE; run one_line
build "-DGSDF_EXP_MARCH_ONE_LINE -DGSDF_EXP_MARCH_NO_STORE"; run neither
build ""; run base_again
