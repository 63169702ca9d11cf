cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_tmp; rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/log.txt 2>&1
f=$(ls $OUT/*/*kernel_trace.csv | head -1)
python - "$f" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
seq=[(r["Kernel_Name"][:14], int(r["End_Timestamp"])-int(r["Start_Timestamp"]), int(r["Start_Timestamp"])) for r in rows]
idx=[i for i,s in enumerate(seq) if "leaf_k" in s[0]]
i1=idx[-1]; i0=idx[-2]+1
t0=seq[i0][2]
for s in seq[i0:i1+1]:
    print(s[0], "dur %.1f us" % (s[1]/1e3), "start +%.1f us" % ((s[2]-t0)/1e3))
PY
rm -rf $OUT
