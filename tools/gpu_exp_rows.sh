#!/bin/bash
# What a z row of a brick costs against what its (x, y) column costs: the evaluating kernel with only the first four rows
# (GSDF_EXP_ROWS=4: timing only, the meshes are wrong) beside the real one; the distinct-rows kernel with every brick forced to
# 5, 6 or 8 rows (GSDF_EXP_DZ_FORCE = 0, 1, 7).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for sc in ${SCENES:-"npt-flange 1600" "knurled-cylinder 2000"}; do set -- $sc
 for fl in "" "-DGSDF_EXP_ROWS=4"; do
  GSDF_HIP_SPEC_FLAGS="$fl" timeout 600 python bench.py --scene $1 --resdiv $2 --steps 10 --warmup 2 --no-cpu-baseline --no-evaluate-dropin --no-mesh-pipeline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 [$fl]', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['phase_ms_rank0'].items()})"
 done
 for fl in "" "-DGSDF_EXP_DZ_FORCE=0" "-DGSDF_EXP_DZ_FORCE=1" "-DGSDF_EXP_DZ_FORCE=7"; do
  GSDF_HIP_SPEC_FLAGS="$fl" timeout 600 python bench.py --scene $1 --resdiv $2 --steps 10 --warmup 2 --no-cpu-baseline --no-evaluate-dropin --no-mesh-pipeline --share-corners 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 rows [$fl]', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['phase_ms_rank0'].items()})"
 done
done
