#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per dispatch for kernels matching a substring: pmc_summarize.py DIR [substr]"""
import csv, glob, json, os, sys
from collections import defaultdict
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else "leaf_kernel"
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    per = defaultdict(float)
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (disp, name), v in per.items():
        acc[name][0] += v; acc[name][1] += 1
out = {k: v[0] / v[1] for k, v in sorted(acc.items())}
print(json.dumps(out, indent=1))
