#!/bin/bash
# A/B of specialised-build flag sets on ONE box: per scene and repetition, one short bench line per GSDF_HIP_SPEC_FLAGS value ("-" = none).
#   bash tools/gpu_ab_flags.sh "-" "-DGSDF_NO_ATAN2_FAST" ...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for rep in 1 2; do
for sc in "npt-flange 1600" "bolt 2000" "knurled-cylinder 2000"; do
for fl in "$@"; do
  if [ "$fl" = "-" ]; then unset GSDF_HIP_SPEC_FLAGS; else export GSDF_HIP_SPEC_FLAGS="$fl"; fi
  set -- $sc "$@"; scene=$1; rd=$2; shift 2
  timeout 600 python bench.py --scene $scene --resdiv $rd --steps 20 --no-cpu-baseline --no-evaluate-dropin --no-distinct-rows --no-one-shot 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; m=d['roofline_march']
print('$scene', '[$fl]', 'ms/step', round(d['ms_per_step'],4), 'eval alone', round(r['kernel_ms'],4), r['kernel'], 'march', round(m['kernel_ms'],4))"
done; done; done
