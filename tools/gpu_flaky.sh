cd $GRAFT_REPO_ROOT
fail=0
for i in $(seq 1 25); do
  timeout 300 python -m pytest tests/test_gpu_mesh.py -m gpu -x -q -k "overflow" > /tmp/run_$i.log 2>&1 || { fail=$((fail+1)); tail -5 /tmp/run_$i.log; }
done
echo "overflow test: $fail failures of 25"
