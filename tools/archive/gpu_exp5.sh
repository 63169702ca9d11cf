#!/bin/bash
# Developer experiment: flat_march_kernel as a pure stream of corner 0 / without its output stream (timing only; the
# library is rebuilt on the box with the experiment's -D and restored afterwards), and the grid size knob
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
build() { (cd gsdf_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -Wno-unused-function -I$GRAFT_REPO_ROOT/include $1 gsdf_hip.hip compile.cpp specialize.cpp -lhiprtc -ldl -o libgsdfhip.so 2>&1 | grep -E "error" ); }
run() {
  timeout 300 python bench.py --mode flat --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$1', 'ms/step %.3f' % j['ms_per_step'], 'lattice %.3f' % j['lattice_pass']['kernel_ms'], 'march %.3f' % j['roofline']['kernel_ms'], 'tris', j['triangles'])"
}
for b in 4 6 8 12 16 32; do GSDF_HIP_FLAT_BPC=$b run bpc$b; done
export GSDF_HIP_FLAT_BPC=${FLAT_BPC_EXP:-8}
build -DGSDF_EXP_FLAT_STREAM_ONLY; run stream_only
build -DGSDF_EXP_FLAT_NO_STORE; run no_store
build ""; run base_again
