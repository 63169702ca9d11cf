#!/bin/bash
# The prune phase as one persistent launch (prune_all_kernel, default) against the chain of launches (GSDF_HIP_PRUNE_ALL=0):
# GPU tests that exercise it, then the bench line both ways, two meshes in flight and one blocking mesh at a time.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/${1:-pall}
mkdir -p $OUT
if [ -z "$SKIP_TESTS" ]; then
timeout 1500 python -m pytest tests/test_gpu_mesh.py tests/test_gpu_prune_bounds.py tests/test_gpu_specialized.py tests/test_gpu_fuzz.py tests/test_gpu_gather.py tests/test_gpu_capi_replay.py -m gpu -x -q > $OUT/pytest_all.log 2>&1; tail -3 $OUT/pytest_all.log
fi
for v in ${VARIANTS:-1 0}; do
 for bpc in ${BPCS:-2}; do
  for mode in "" "--no-mesh-pipeline"; do
  GSDF_HIP_PRUNE_ALL=$v GSDF_HIP_PRUNE_ALL_BPC=$bpc timeout 600 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-evaluate-dropin $BENCH_ARGS $mode 2>$OUT/err_$v.txt | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('prune_all=$v bpc=$bpc $mode', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['phase_ms_rank0'].items()}, 'alone', d['roofline'].get('alone',{}).get('ms_per_mesh_device'))"
  done
 done
done
