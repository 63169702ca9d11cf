#!/usr/bin/env python3
"""List the lowered device program of a scene (host-only, no GPU needed): tools/disasm.py npt-flange"""
import os, re, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scaffold.builder import Builder
from gsdf_amd import hip

src = open(os.path.join(os.path.dirname(__file__), "..", "gsdf_amd", "csrc", "dev_ops.h")).read()
body = src[src.index("enum DevOp"):src.index("D_OP_COUNT")]
body = re.sub(r"//[^\n]*", "", body)
names = re.findall(r"\bD_[A-Z0-9_]+", body)
tab = src[src.index("kDevOpParams[D_OP_COUNT]"):]
tab = re.sub(r"/\*.*?\*/", "", tab[tab.index("{") + 1:tab.index("}")])
nparams = [int(x) for x in re.findall(r"\d+", tab)]
assert len(names) == len(nparams), (len(names), len(nparams))

def listing(code):
    pc, out = 0, []
    f = code.view(np.float32)
    while True:
        w = int(code[pc]); op = w & 0x0fff; slot = w >> 16
        fl = ("|HXY" if w & 0x4000 else "") + ("|SWAP" if w & 0x8000 else "") + ("|SHXY" if w & 0x2000 else "") + ("|SHZ" if w & 0x1000 else "")
        name = names[op]
        if name == "D_POLY2D":
            nv = int(code[pc + 1]) & 0x7fffffff
            q0 = (pc + 4 + 7) & ~7
            out.append(f"{pc:5d} {name}{fl} slot={slot} nv={nv} fast={int(code[pc+1])>>31}")
            pc = q0 + 8 * nv
        elif name in ("D_UBOUND2D", "D_UBOUND3D"):
            nb = int(code[pc + 1]); out.append(f"{pc:5d} {name}{fl} slot={slot} boxes={nb}"); pc += 2 + (4 if name == "D_UBOUND2D" else 6) * nb
        elif name == "D_LINES2D":
            ns = int(code[pc + 1]); out.append(f"{pc:5d} {name}{fl} ns={ns}"); pc += 3 + 5 * ns
        elif name == "D_SKIP":
            sw = int(code[pc + 3])
            out.append(f"{pc:5d} {name}{fl} id={int(code[pc+1])} subst={f[pc+2]:.3g} -> {pc + (sw & 0xffffff)}"); pc += 4
        elif name == "D_LIP_DOM":
            k, ia, ib, pi = (int(code[pc + 1 + i]) for i in (0, 1, 2, 4))
            kinds = ["min", "max", "diff", "sunion", "sdiff", "sinter"]
            out.append(f"{pc:5d} {name}{fl} slot={slot} {kinds[k]} ida={ia if ia != 255 else '-'} idb={ib if ib != 255 else '-'} kk={f[pc+4]:.6g} depth={pi & 0xffff}{' 2d' if pi & 0x10000 else ''}"); pc += 6
        elif name.startswith("D_GATE"):
            n = nparams[op]
            sw = int(code[pc + n])
            out.append(f"{pc:5d} {name}{fl} slot={slot} " + " ".join(f"{f[pc+1+i]:.6g}" for i in range(n - 1)) + f" -> {pc + (sw & 0xffffff)}" + (f" id={(sw >> 24) - 1}" if sw >> 24 else ""))
            pc += 1 + n
        else:
            n = nparams[op]
            out.append(f"{pc:5d} {name}{fl} slot={slot} " + " ".join(f"{f[pc+1+i]:.6g}" for i in range(n)))
            pc += 1 + n
        if name == "D_END":
            break
    return out

if __name__ == "__main__":
    sh = Builder().Scene(sys.argv[1] if len(sys.argv) > 1 else "npt-flange")
    code, slots = hip.lower(sh)
    print(f"{len(code)} words, {slots} LDS slots")
    print("\n".join(listing(code)))
