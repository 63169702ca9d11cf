# PC sampling of the evaluating kernel (rocprofv3 beta): where the wave cycles of leaf_eval_kernel go, by instruction.
# One blocking mesh at a time; the specialised code object is kept (GSDF_HIP_CACHE_DIR) so that PCs can be mapped to its ISA.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pcs}
rm -rf $OUT; mkdir -p $OUT/cache
export GSDF_HIP_CACHE_DIR=$OUT/cache
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
ARGS="--steps 6 --warmup 1 --preheat 4 --no-cpu-baseline --no-evaluate-dropin --no-mesh-pipeline ${SCENE_ARGS}"
cd /tmp
for m in stochastic host_trap; do
  unit=cycles; iv=1048576; if [ $m = host_trap ]; then unit=time; iv=1; fi
  timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $unit --pc-sampling-method $m --pc-sampling-interval $iv --kernel-trace --output-format csv -d $OUT/$m -- python $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/$m.log 2>&1
  echo "$m rc=$?"; tail -3 $OUT/$m.log | cut -c1-300
  find $OUT/$m -type f | head; 
done
cd $GRAFT_REPO_ROOT
ls -la $OUT/cache | head
# keep it small
find $OUT -name "*.csv" -size +40M -delete
du -sh $OUT
