#!/bin/bash
# The polygon culling's wave reductions (interp.h: wave_box / wave_minmax, poly_cull): parity tests that reach it, the concave-polygon
# fuzz, then the three single-GPU configs.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-cull}
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_mesh.py tests/test_gpu_specialized.py tests/test_gpu_fuzz.py -m gpu -x -q -k "identical or full_size or random or fuzz" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
bash tools/gpu_fuzz_poly.sh ${FUZZ:-9000:9030} 600 2>&1 | tail -3
SKIP_TESTS=1 bash tools/gpu_rows.sh ${1:-cull}
