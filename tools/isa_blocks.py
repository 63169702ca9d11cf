#!/usr/bin/env python3
"""Host only: basic-block histogram of one kernel in an ISA listing made by tools/specdev.py --isa (llvm-objdump format): the largest
blocks with their telling opcodes, and the kernel's expensive operations (divisions, square roots, float64, DPP reductions) in program
order -- what showed round 6 the float64 atan2 standing twice in npt-flange's leaf kernel.   tools/isa_blocks.py f.s <kernel name prefix>"""
import collections
import re
import sys

L = open(sys.argv[1]).read().split("\n")
pref = sys.argv[2]
st = [i for i, l in enumerate(L) if re.match(r"^[0-9a-f]{16} <" + re.escape(pref), l)][0]
en = len(L)
for i in range(st + 1, len(L)):
    if re.match(r"^[0-9a-f]{16} <", L[i]):
        en = i
        break
base = int(L[st].split()[0], 16)
ins = []
for l in L[st + 1:en]:
    m = re.match(r"^\t(\S+)(.*?)//\s*([0-9A-F]+):", l)
    if m:
        ins.append((int(m.group(3), 16) - base, m.group(1), l))
targets = set()
for off, op, l in ins:
    m = re.search(r"\+0x([0-9a-f]+)>", l)
    if op.startswith(("s_cbranch", "s_branch")) and m:
        targets.add(int(m.group(1), 16))
blocks, cur = [], []
for off, op, l in ins:
    if off in targets and cur:
        blocks.append(cur)
        cur = []
    cur.append((off, op))
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
        blocks.append(cur)
        cur = []
if cur:
    blocks.append(cur)
print(L[st][:120])
print(len(ins), "instructions,", len(blocks), "blocks")
KEY = ("v_div_scale", "v_sqrt", "v_rcp", "ds_", "v_min_f32_dpp", "v_max_f32_dpp", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_cndmask", "v_readlane", "s_nop", "v_cmp", "v_mov", "v_floor", "v_sin", "v_cos")
for b in sorted(blocks, key=lambda b: -len(b))[:int(sys.argv[3]) if len(sys.argv) > 3 else 14]:
    c = collections.Counter(i[1] for i in b)
    k2 = {}
    for k, v in c.items():
        if k.startswith(KEY):
            kk = re.sub(r"_e(32|64)$", "", k)
            k2[kk] = k2.get(kk, 0) + v
    print(hex(b[0][0]), len(b), "valu", sum(v for k, v in c.items() if k.startswith("v_")), dict(sorted(k2.items(), key=lambda kv: -kv[1])[:8]))
seq = []
for off, op, l in ins:
    if op.startswith(("v_div_scale_f32", "v_sqrt_f32", "v_div_scale_f64", "v_rcp_f64", "v_sqrt_f64", "v_min_f32_dpp", "v_floor", "v_cvt_f64_f32", "v_cvt_f32_f64", "v_cvt_u32_f64", "v_cvt_f64_u32", "v_sin", "v_cos", "v_rndne")):
        o2 = re.sub(r"_e(32|64)$", "", op).replace("v_", "")
        if seq and seq[-1][1] == o2:
            seq[-1][2] += 1
        else:
            seq.append([off, o2, 1])
print(" ".join(f"{hex(o)}:{op}x{n}" for o, op, n in seq))
