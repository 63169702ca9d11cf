#!/usr/bin/env python3
"""Static instruction mix of one kernel in an llvm-objdump -d listing: tools/isa_stats.py file.s kernel_substring"""
import collections, re, sys
path, want = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
cur, ops = None, collections.Counter()
for l in open(path):
    m = re.match(r"^[0-9a-f]+ <(.*)>:", l)
    if m:
        cur = m.group(1); continue
    if cur is None or want not in cur: continue
    m = re.match(r"^\s+([a-z_0-9]+)", l)
    if m: ops[m.group(1)] += 1
tot = sum(ops.values())
cls = collections.Counter()
for k, v in ops.items():
    c = ("valu" if k.startswith("v_") else "salu" if k.startswith("s_") and not k.startswith(("s_load", "s_waitcnt", "s_nop", "s_cbranch", "s_branch", "s_barrier", "s_buffer"))
         else "smem" if k.startswith(("s_load", "s_buffer")) else "lds" if k.startswith("ds_") else "vmem" if k.startswith(("global_", "scratch_", "flat_", "buffer_")) else k.split("_")[1] if k.startswith("s_") else k)
    cls[c] += v
print("total", tot, dict(cls))
f64 = sum(v for k, v in ops.items() if "f64" in k)
lane = sum(v for k, v in ops.items() if k in ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32"))
mov = sum(v for k, v in ops.items() if k.startswith("v_mov") or k.startswith("v_accvgpr"))
cnd = sum(v for k, v in ops.items() if k.startswith("v_cndmask"))
cmp_ = sum(v for k, v in ops.items() if k.startswith("v_cmp"))
print(f"f64 {f64}  lane-moves {lane}  v_mov {mov}  cndmask {cnd}  v_cmp {cmp_}")
for k, v in ops.most_common(top): print(f"{v:6d} {k}")
