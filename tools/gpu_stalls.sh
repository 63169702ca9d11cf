# Stall attribution of leaf_eval_kernel from counters (PC sampling is not supported on this box: "Given PC sampling configuration
# is not supported on any of the agents", rocprofv3 of ROCm 7.0.2): instruction classes, who occupies the issue cycles, what the
# waves wait for, how many lanes are active. One blocking mesh at a time. -> gpurun_out/<tag>_<scene>_stalls/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r4}
for sc in "npt-flange 1600" "bolt 2000" "knurled-cylinder 2000"; do set -- $sc
  OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_$1_stalls; rm -rf $OUT; mkdir -p $OUT
  ARGS="--scene $1 --resdiv $2 --steps 4 --warmup 1 --preheat 4 --no-cpu-baseline --no-evaluate-dropin --no-mesh-pipeline"
  timeout 300 python bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err
  ( cd /tmp
    P="python $GRAFT_REPO_ROOT/bench.py $ARGS"
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d $OUT/p1 -- $P > $OUT/p1.log 2>&1
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU --output-format csv -d $OUT/p2 -- $P > $OUT/p2.log 2>&1
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p3 -- $P > $OUT/p3.log 2>&1
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 --output-format csv -d $OUT/p4 -- $P > $OUT/p4.log 2>&1
    timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_LEVEL_WAVES SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/p5 -- $P > $OUT/p5.log 2>&1 )
  python - $OUT <<'PY'
import json, subprocess, sys
out = sys.argv[1]
b = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
s = subprocess.check_output([sys.executable, "tools/pmc_summarize.py", out, "--evals-per-launch", str(b["evals_per_step"]), "--kernel-ms", str(b["roofline"]["kernel_ms"]),
                             "--workload", b["config"]["workload"], "--code", b["config"].get("code") or "", "--command", "python bench.py (see tools/gpu_stalls.sh)"])
open(out + "/stalls.json", "wb").write(s)
j = json.loads(s)["leaf_eval_kernel"]
print(out.split("/")[-1], {k: round(v, 3) if isinstance(v, float) and abs(v) < 1e4 else v for k, v in j.items()})
PY
  find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +8M -delete
done
