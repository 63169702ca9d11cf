"""Host lowering (gsdf_amd/csrc/compile.cpp) checked without a GPU through gsdf_hip_lower: instruction
stream shape, LDS slot allocation, the position-save elision and the hypot(x,y) sharing flags."""
import os
import re

import numpy as np
import pytest

import corpus
from gsdf_amd import hip
from gsdf_amd.builder import Builder

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_HDR = open(os.path.join(ROOT, "gsdf_amd", "csrc", "dev_ops.h")).read()
_ENUM = _HDR[_HDR.index("enum DevOp"):_HDR.index("D_OP_COUNT")]
NAMES = [n for n in dict.fromkeys(re.findall(r"\b(D_[A-Z0-9_]+)\b", re.sub(r"//[^\n]*", "", _ENUM)))]
NPAR = [int(x) for x in re.findall(r"\*/\s*(\d+)", re.search(r"kDevOpParams\[D_OP_COUNT\] = \{(.*?)\};", _HDR, re.S).group(1))]
FLAG_SHZ, FLAG_SHXY, FLAG_HXY, FLAG_SWAP, OP_MASK = 0x1000, 0x2000, 0x4000, 0x8000, 0x0FFF


def decode(code):
    out, pc = [], 0
    while pc < len(code):
        w = int(code[pc])
        op = w & OP_MASK
        name, n = NAMES[op], NPAR[op]
        if name == "D_POLY2D":
            nv = int(code[pc + 1]) & 0x7FFFFFFF
            n = ((pc + 4 + 7) & ~7) - (pc + 1) + 8 * nv
        elif name == "D_LINES2D":
            n = 2 + 5 * int(code[pc + 1])
        elif name in ("D_UBOUND2D", "D_UBOUND3D"):
            n = 1 + (4 if name == "D_UBOUND2D" else 6) * int(code[pc + 1])
        out.append((name, bool(w & FLAG_HXY), bool(w & FLAG_SWAP), w >> 16, pc))
        pc += 1 + n
        if name == "D_END":
            break
    assert pc <= len(code)   # read-only tables (D_CIRC_PRE sin/cos) may follow D_END
    return out


def test_opcode_table_is_consistent():
    assert len(NAMES) == len(NPAR)
    assert NAMES[0] == "D_END"


def test_npt_flange_program():
    b = Builder()
    code, slots = hip.lower(b.Scene("npt-flange"))
    ins = decode(code)
    names = [i[0] for i in ins]
    assert names[0] == "D_SCALE_PRE" and names[-2:] == ["D_MULR", "D_END"]
    assert names.count("D_POLY2D") == 1 and names.count("D_SCREW_PRE") == 1
    assert names.count("D_CYL0") + names.count("D_CYLR") == 3
    # the hole cylinder is evaluated first (position-preserving), so the root difference needs no position save
    assert names.count("D_SAVEP3") == 1 and names.count("D_LOADP3") == 1
    assert [i for i in ins if i[0] == "D_COMBINE_DIFF"][-1][2] is True     # swapped operands
    # hypot(x,y) is computed once and reused by the other two cylinders and the screw (z-only translate keeps it)
    users = [i for i in ins if i[0] in ("D_CYL0", "D_CYLR", "D_SCREW_PRE")]
    assert [u[1] for u in users].count(False) == 1 and [u[1] for u in users].count(True) == 3
    assert slots == 6   # 1 root distance + 3 saved position + 2 nested partial results
    # polygon edge records start on a 32-byte boundary
    poly = [i for i in ins if i[0] == "D_POLY2D"][0]
    assert ((poly[4] + 4 + 7) & ~7) % 8 == 0


def test_saved_positions_are_not_duplicated():
    """A frame that needs the position an enclosing frame already saved reuses that slot: the deep example
    trees keep one copy of the root position (bolt 13 -> 6 slots, knurled-cylinder 18 -> 10: 4 points per lane)."""
    b = Builder()
    for scene, max_slots, saves in (("bolt", 6, 1), ("knurled-cylinder", 10, 3)):
        code, slots = hip.lower(b.Scene(scene))
        ins = decode(code)
        assert slots <= max_slots, (scene, slots)
        assert [i[0] for i in ins].count("D_SAVEP3") == saves, scene
        # every load reads a slot that some earlier instruction saved
        saved = set()
        for name, _, _, slot, _ in ins:
            if name in ("D_SAVEP3", "D_SAVEP2"):
                saved.add(slot)
            if name in ("D_LOADP3", "D_LOADP2"):
                assert slot in saved, (scene, slot)
    # a position rewritten between two frames is saved again (different value)
    sh = b.Union(b.Translate(b.Union(b.Translate(b.NewSphere(1), 1, 0, 0), b.Translate(b.NewSphere(1), 0, 1, 0)), 0, 0, 1),
                 b.Translate(b.NewSphere(1), 0, 0, 2))
    names = [i[0] for i in decode(hip.lower(sh)[0])]
    assert names.count("D_SAVEP3") == 2


def flags_of(code):
    out, pc = [], 0
    for name, _, _, _, at in decode(code):
        out.append((name, bool(int(code[at]) & FLAG_SHXY), bool(int(code[at]) & FLAG_SHZ)))
    return out


def test_corner_pair_sharing_flags():
    """D_FLAG_SHXY / D_FLAG_SHZ: set only while P.xy (resp. P.z) is still a function of the entry x,y (entry z)."""
    b = Builder()
    fl = flags_of(hip.lower(b.Scene("npt-flange"))[0])
    # scale and z-translate are component-wise: all three cylinders and the screw may share hypot/atan2 between z-pairs
    assert [f[1] for f in fl if f[0] in ("D_CYL0", "D_CYLR", "D_SCREW_PRE")] == [True] * 4
    fl = flags_of(hip.lower(b.Scene("knurled-cylinder"))[0])
    assert [f[2] for f in fl if f[0] == "D_TWIST"] == [True, True]          # twist angle depends on entry z only
    assert [f[1] for f in fl if f[0] == "D_CIRC_PRE"] == [False, False]     # after the twist x,y depend on z
    assert [f[1] for f in fl if f[0] in ("D_CYL0", "D_CYLR")][:2] == [True, True]
    fl = flags_of(hip.lower(b.Scene("bolt"))[0])
    assert not any(f[1] or f[2] for f in fl)                                # a general rotation comes first
    # a rotation about z keeps z-only and xy-only dependencies apart; a second saved/restored branch gets its own
    sh = b.Union(b.Translate(b.Rotate(b.NewCylinder(1, 2, 0), 0.3, (0, 0, 1)), 0.5, 0, 0), b.Translate(b.NewCylinder(1, 2, 0), 0, 0, 1))
    fl = flags_of(hip.lower(sh)[0])
    cyl = [f for f in fl if f[0] == "D_CYL0"]
    assert len(cyl) == 2 and cyl[0][1] != cyl[1][1]  # the transformed branch (4x4 matrix) is not shareable, the other is


def test_far_child_skip_in_wide_unions():
    """Wide unions (>= 4 children) of exact-boxed children: D_UBOUND* starts the running minimum at an upper bound of the
    union, every boxed child is preceded by D_SKIPFAR* whose skip lands on the instruction after the child's
    D_COMBINE_MIN; the boxes are the children's (translated polygon bounds)."""
    b = Builder()
    code, _ = hip.lower(b.Scene("glyph-plate"))
    ins = decode(code)
    starts = {i[4] for i in ins}
    f = code.view(np.float32)
    ub = [i for i in ins if i[0] == "D_UBOUND2D"]
    nb = int(code[ub[0][4] + 1])
    assert len(ub) == 1 and nb == 18                         # the 6 "D" glyphs are differences: lower bound only, no upper bound
    ub_boxes = {tuple(f[ub[0][4] + 2 + 4 * k: ub[0][4] + 6 + 4 * k]) for k in range(nb)}
    skips = [i for i in ins if i[0] == "D_SKIPFAR2D"]
    assert len(skips) == 24                                  # every glyph, the first one included
    for k, (name, _, _, slot, pc) in enumerate(skips):
        assert slot == ub[0][3]                              # all test against the slot the bound was stored in
        target = pc + int(code[pc + 5])
        assert target in starts                              # lands on an instruction boundary ...
        prev = max(p for p in starts if p < target)
        assert [i[0] for i in ins if i[4] == prev] == ["D_COMBINE_MIN"]   # ... right after the child's combine
        assert [i[3] for i in ins if i[4] == prev] == [slot]              # same running-minimum slot
        x0, y0, x1, y1 = f[pc + 1:pc + 5]
        assert x1 - x0 == 6.0 and y1 - y0 == 10.0            # glyph cell of the scene (threads.hpp: 6 x 10)
        assert ((x0, y0, x1, y1) in ub_boxes) == (k % 4 != 2)  # the bound lists the boxes of the solid glyphs (G, S, F)
    # children without an exact-distance guarantee are never skipped: smoothing, scaling, approximate primitives
    def mk(child):
        return b.Union2D(b.NewCircle(1), b.Translate2D(b.NewCircle(1), 3, 0), b.Translate2D(b.NewCircle(1), 6, 0), child)
    names = lambda sh: [i[0] for i in decode(hip.lower(sh)[0])]
    assert names(mk(b.Translate2D(b.NewCircle(1), 9, 0))).count("D_SKIPFAR2D") == 4
    assert names(mk(b.Translate2D(b.NewEllipse(1, 0.5), 9, 0))).count("D_SKIPFAR2D") == 3
    assert names(mk(b.Offset2D(b.NewCircle(1), 0.1))).count("D_SKIPFAR2D") == 3
    # uniform scaling and rigid motions keep the field exact: their (scaled / rotated, conservatively boxed) children qualify
    assert names(mk(b.Scale2D(b.NewCircle(1), 2.0))).count("D_SKIPFAR2D") == 4
    rot = hip.lower(mk(b.Translate2D(b.Rotate2D(b.NewRectangle(2, 1), 0.5), 9, 0)))[0]
    ir = decode(rot)
    assert [i[0] for i in ir].count("D_SKIPFAR2D") == 4
    fr = rot.view(np.float32)
    boxes = [tuple(fr[i[4] + 1:i[4] + 5]) for i in ir if i[0] == "D_SKIPFAR2D"]
    rb = max(boxes, key=lambda q: q[0])                      # the rotated rectangle, translated to x = 9
    half_w = 0.5 * (2 * abs(np.cos(0.5)) + 1 * abs(np.sin(0.5)))
    assert abs((rb[2] - rb[0]) / 2 - half_w) < 1e-3 and abs((rb[0] + rb[2]) / 2 - 9) < 1e-3
    three = names(b.Union2D(b.NewCircle(1), b.Translate2D(b.NewCircle(1), 3, 0), b.Translate2D(b.NewCircle(1), 6, 0)))
    assert three.count("D_SKIPFAR2D") == 0 and three.count("D_UBOUND2D") == 0
    # fewer than three boxed children: no upper bound, the first child is always evaluated
    two = names(b.Union2D(b.NewCircle(1), b.Translate2D(b.NewCircle(1), 3, 0), b.NewEllipse(1, 0.5), b.NewEllipse(2, 0.5)))
    assert two.count("D_UBOUND2D") == 0 and two.count("D_SKIPFAR2D") == 1
    # a difference may be empty: it can be skipped (lower bound) but contributes no upper bound
    dd = b.Union2D(b.NewCircle(1), b.Translate2D(b.NewCircle(1), 3, 0), b.Translate2D(b.NewCircle(1), 6, 0),
                   b.Translate2D(b.Difference2D(b.NewCircle(1), b.NewCircle(2)), 9, 0))
    cdd = hip.lower(dd)[0]
    idd = decode(cdd)
    assert [i[0] for i in idd].count("D_SKIPFAR2D") == 4 and int(cdd[[i[4] for i in idd if i[0] == "D_UBOUND2D"][0] + 1]) == 3
    # 3-D: extruded exact shapes and boxes qualify
    sh3 = b.Union(b.NewSphere(1), b.Translate(b.NewBox(1, 1, 1, 0), 3, 0, 0), b.Translate(b.Extrude(b.NewRectangle(1, 1), 1), 6, 0, 0),
                  b.Translate(b.NewTorus(1, 0.2), 9, 0, 0))
    assert names(sh3).count("D_SKIPFAR3D") == 3 and names(sh3).count("D_UBOUND3D") == 1   # the torus makes no claim


def test_hxy_not_reused_across_xy_changes():
    b = Builder()
    c = b.NewCylinder(1, 2, 0)
    s = b.Union(c, b.Translate(c, 0.5, 0, 0), b.Translate(c, 0, 0, 3))
    ins = decode(hip.lower(s)[0])
    cyls = [i for i in ins if i[0] == "D_CYL0"]
    assert len(cyls) == 3
    # order: non-clobbering child first; the z-only translate may reuse, the x translate may not
    flags = {}
    prev = None
    for i in ins:
        if i[0] == "D_TRANSLATE":
            prev = i
        if i[0] == "D_CYL0":
            flags[prev[4] if prev else -1] = i[1]
    assert list(flags.values()).count(True) <= 1
    for name, sh in corpus.shapes3d()[1] + corpus.shapes2d()[1]:
        decode(hip.lower(sh)[0])  # every corpus program decodes cleanly to D_END


def test_multi_evaluation_nodes_are_unrolled():
    b = Builder()
    box = b.NewBox(1, 1, 1, 0)
    arr = b.Array(box, 2, 2, 2, 3, 3, 3)
    names = [i[0] for i in decode(hip.lower(arr)[0])]
    assert names.count("D_ARRAY_PRE") == 8 and names.count("D_BOX") == 8 and names.count("D_COMBINE_MIN") == 8
    circ = b.CircularArray(b.Translate(box, 2, 0, 0), 5, 7)
    names = [i[0] for i in decode(hip.lower(circ)[0])]
    assert names.count("D_CIRC_PRE") == 1 and names.count("D_BOX") == 2
    k = b.Scene("knurled-cylinder")
    names = [i[0] for i in decode(hip.lower(k)[0])]
    assert names.count("D_TWIST") == 2 and names.count("D_CIRC_PRE") == 2 and names.count("D_BOX") == 4  # shared subtree re-emitted


def test_bad_trees_raise():
    from gsdf_amd._ctypes_common import GsdfNode, GsdfTree, OP
    import ctypes as C
    nodes = (GsdfNode * 2)()
    nodes[0].op = OP["SPHERE"]
    nodes[1].op = OP["EXTRUSION"]  # extrusion of a 3D child
    nodes[1].nchild = 1
    links = (C.c_uint32 * 1)(0)
    t = GsdfTree(nodes, 2, links, 1, None, 0, 1)
    with pytest.raises(hip.HipError) as e:
        hip.lower(t)
    assert e.value.code == -4 and "dimension" in e.value.msg
