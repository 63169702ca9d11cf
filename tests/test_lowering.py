"""Host lowering (gsdf_amd/csrc/compile.cpp) checked without a GPU through gsdf_hip_lower: instruction
stream shape, LDS slot allocation, the position-save elision and the hypot(x,y) sharing flags."""
import os
import re

import numpy as np
import pytest

import corpus
from gsdf_amd import hip
from scaffold.builder import Builder

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


def at(ins, target, field=0):
    """The instruction a gate's skip lands on: the child's combine (an interval-mode-only D_LIP_DOM in front of it is stepped over)."""
    k = [j for j, i in enumerate(ins) if i[4] == target]
    assert len(k) == 1, target
    j = k[0] + (1 if ins[k[0]][0] == "D_LIP_DOM" else 0)
    return [ins[j][field]]


def skipw(code, at_):
    """A gate's skip distance (the word's top byte is the child's brick-mask number + 1)."""
    return int(code[at_]) & 0xffffff


def test_opcode_table_is_consistent():
    assert len(NAMES) == len(NPAR)
    assert NAMES[0] == "D_END"


def test_npt_flange_program():
    b = Builder()
    code, slots = hip.lower(b.Scene("npt-flange"))
    ins = decode(code)
    names = [i[0] for i in ins]
    # (the root Scale sits between an interval-stack push and pop: the cube's radius comes back exactly after the node)
    assert names[:2] == ["D_LIP_PUSH", "D_SCALE_PRE"] and names[-3:] == ["D_MULR", "D_LIP_POP", "D_END"]
    assert names.count("D_POLY2D") == 1 and names.count("D_SCREW_PRE") == 1
    assert names.count("D_CYL0") + names.count("D_CYLR") == 3
    # the hole cylinder is evaluated first (position-preserving), so the root difference needs no position save
    assert names.count("D_SAVEP3") == 1 and names.count("D_LOADP3") == 1
    assert [i for i in ins if i[0] == "D_COMBINE_DIFF"][-1][2] is True     # swapped operands
    # hypot(x,y) is computed once and reused by the other two cylinders and the screw (z-only translate keeps it)
    users = [i for i in ins if i[0] in ("D_CYL0", "D_CYLR", "D_SCREW_PRE")]
    assert [u[1] for u in users].count(False) == 1 and [u[1] for u in users].count(True) == 3
    assert slots == 7   # 1 root distance + 3 saved position + 2 nested partial results + the screw's |z| - L
    # gates: the cheap plate is evaluated before the threaded pipe, which is skipped where the plate's value proves the smooth
    # union's blend weight clamps (box of the nut cylinder); inside it the screw + thread polygon is skipped where the nut
    # cylinder's value proves max(nut, -screw) = nut (z-cylinder of the screw, exact radius from the cached hypot)
    gates = [i for i in ins if i[0].startswith("D_GATE")]
    assert [g[0] for g in gates] == ["D_GATE3D", "D_GATEZC"] and gates[1][1] is True
    f = code.view(np.float32)
    g0, g1 = gates[0][4], gates[1][4]
    assert f[g0 + 7] == 1.0 and abs(f[g0 + 8] - 1.002 * 0.2) < 1e-6 and f[g1 + 8] == -1.0 and f[g1 + 9] == 0.0
    assert skipw(code, g0 + 9) == 0xffff and skipw(code, g1 + 10) == 0xffff            # no enclosing-difference context here
    # (the skip word's top byte carries the child's brick-mask number + 1; the skip lands on the combine, or on the
    # interval-mode-only D_LIP_DOM in front of it)
    t0, t1 = g0 + (int(code[g0 + 12]) & 0xffffff), g1 + (int(code[g1 + 13]) & 0xffffff)
    assert [i[0] for i in ins if i[4] in (t0, t0 + 6)] == ["D_LIP_DOM", "D_COMBINE_SUNION"]
    assert [i[0] for i in ins if i[4] in (t1, t1 + 6)] == ["D_LIP_DOM", "D_COMBINE_DIFF"]
    assert at(ins, t0 + 6, 3) == [gates[0][3]] and at(ins, t1 + 6, 3) == [gates[1][3]]
    assert int(code[g0 + 12]) >> 24 and int(code[g1 + 13]) >> 24      # both gated children carry a number
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
    """Wide unions (>= 4 children) of children with a lower-bound region: D_UBOUND* starts the running minimum at an upper
    bound of the union, every such child is preceded by a D_GATE* whose skip lands on the child's D_COMBINE_MIN (which then
    runs on (a, L)); the boxes are the children's (translated polygon bounds)."""
    b = Builder()
    code, _ = hip.lower(b.Scene("glyph-plate"))
    ins = decode(code)
    f = code.view(np.float32)
    ub = [i for i in ins if i[0] == "D_UBOUND2D"]
    nb = int(code[ub[0][4] + 1])
    assert len(ub) == 1 and nb == 18                         # the 6 "D" glyphs are differences: lower bound only, no upper bound
    ub_boxes = {tuple(f[ub[0][4] + 2 + 4 * k: ub[0][4] + 6 + 4 * k]) for k in range(nb)}
    gates = [i for i in ins if i[0] == "D_GATE2D"]
    skips = [i for i in gates if i[3] == ub[0][3]]           # those testing against the slot the bound was stored in
    assert len(skips) == 24                                  # every glyph, the first one included
    for k, (name, _, _, slot, pc) in enumerate(skips):
        target = pc + skipw(code, pc + 10)
        assert at(ins, target, 0) == ["D_COMBINE_MIN"]   # lands on the child's combine ...
        assert at(ins, target, 3) == [slot]              # ... of the same running-minimum slot
        x0, y0, x1, y1 = f[pc + 1:pc + 5]
        assert x1 - x0 == 6.0 and y1 - y0 == 10.0            # glyph cell of the scene (threads.hpp: 6 x 10)
        assert f[pc + 5] == 1.0 and f[pc + 6] == 0.0         # union: compare with +a, no blend width
        assert ((x0, y0, x1, y1) in ub_boxes) == (k % 4 != 2)  # the bound lists the boxes of the solid glyphs (G, S, F)
    # the "D" glyphs are differences of two polygons: the inner one (the subtrahend) is gated by the outer one's value
    inner = [i for i in gates if i[3] != ub[0][3]]
    assert len(inner) == 6
    for (name, _, _, slot, pc) in inner:
        target = pc + skipw(code, pc + 10)
        assert at(ins, target, 0) == ["D_COMBINE_DIFF"] and f[pc + 5] == -1.0 and f[pc + 6] == 0.0
        assert at(ins, target, 2) == [False]   # never with swapped operands: only a subtrahend can be skipped
    # children without a lower-bound claim are never skipped: approximate primitives
    def mk(child):
        return b.Union2D(b.NewCircle(1), b.Translate2D(b.NewCircle(1), 3, 0), b.Translate2D(b.NewCircle(1), 6, 0), child)
    names = lambda sh: [i[0] for i in decode(hip.lower(sh)[0])]
    assert names(mk(b.Translate2D(b.NewCircle(1), 9, 0))).count("D_GATE2D") == 4
    assert names(mk(b.Translate2D(b.NewEllipse(1, 0.5), 9, 0))).count("D_GATE2D") == 3
    # an offset shifts the field: a negative one grows the region by |off|, a positive one keeps the child's region
    off = hip.lower(mk(b.Translate2D(b.Offset2D(b.NewCircle(1), -0.25), 9, 0)))[0]
    io = decode(off)
    assert [i[0] for i in io].count("D_GATE2D") == 4
    fo = off.view(np.float32)
    ob = max((tuple(fo[i[4] + 1:i[4] + 5]) for i in io if i[0] == "D_GATE2D"), key=lambda q: q[0])
    assert abs(ob[0] - (9 - 1.25)) < 1e-3 and abs(ob[2] - (9 + 1.25)) < 1e-3
    # uniform scaling and rigid motions keep the field exact: their (scaled / rotated, conservatively boxed) children qualify
    assert names(mk(b.Scale2D(b.NewCircle(1), 2.0))).count("D_GATE2D") == 4
    rot = hip.lower(mk(b.Translate2D(b.Rotate2D(b.NewRectangle(2, 1), 0.5), 9, 0)))[0]
    ir = decode(rot)
    assert [i[0] for i in ir].count("D_GATE2D") == 4
    fr = rot.view(np.float32)
    boxes = [tuple(fr[i[4] + 1:i[4] + 5]) for i in ir if i[0] == "D_GATE2D"]
    rb = max(boxes, key=lambda q: q[0])                      # the rotated rectangle, translated to x = 9
    half_w = 0.5 * (2 * abs(np.cos(0.5)) + 1 * abs(np.sin(0.5)))
    assert abs((rb[2] - rb[0]) / 2 - half_w) < 1e-3 and abs((rb[0] + rb[2]) / 2 - 9) < 1e-3
    # narrow unions of cheap children: evaluating a circle costs less than testing whether it matters
    three = names(b.Union2D(b.NewCircle(1), b.Translate2D(b.NewCircle(1), 3, 0), b.Translate2D(b.NewCircle(1), 6, 0)))
    assert three.count("D_GATE2D") == 0 and three.count("D_UBOUND2D") == 0
    # fewer than three solid boxed children: no upper bound; the children without a claim go first, the others are gated
    two = names(b.Union2D(b.NewCircle(1), b.Translate2D(b.NewCircle(1), 3, 0), b.NewEllipse(1, 0.5), b.NewEllipse(2, 0.5)))
    assert two.count("D_UBOUND2D") == 0 and two.count("D_GATE2D") == 2
    assert [n for n in two if n in ("D_ELLIPSE2D", "D_CIRCLE2D")] == ["D_ELLIPSE2D", "D_ELLIPSE2D", "D_CIRCLE2D", "D_CIRCLE2D"]
    # a difference may be empty: it can be skipped (lower bound) but contributes no upper bound
    dd = b.Union2D(b.NewCircle(1), b.Translate2D(b.NewCircle(1), 3, 0), b.Translate2D(b.NewCircle(1), 6, 0),
                   b.Translate2D(b.Difference2D(b.NewCircle(1), b.NewCircle(2)), 9, 0))
    cdd = hip.lower(dd)[0]
    idd = decode(cdd)
    assert [i[0] for i in idd].count("D_GATE2D") == 4 and int(cdd[[i[4] for i in idd if i[0] == "D_UBOUND2D"][0] + 1]) == 3
    # 3-D: extruded exact shapes and boxes qualify
    sh3 = b.Union(b.NewSphere(1), b.Translate(b.NewBox(1, 1, 1, 0), 3, 0, 0), b.Translate(b.Extrude(b.NewRectangle(1, 1), 1), 6, 0, 0),
                  b.Translate(b.NewTorus(1, 0.2), 9, 0, 0))
    assert names(sh3).count("D_GATE3D") == 3 and names(sh3).count("D_UBOUND3D") == 1   # the torus makes no claim


def test_gates_of_binary_combines():
    """A child that is expensive and has a lower-bound region is evaluated last and preceded by a gate: unions, the
    subtrahend of a (smooth) difference, either operand of a smooth union. Minuends, intersections and xor never."""
    b = Builder()
    poly = lambda: b.NewPolygon([(0, 0), (2, 0), (2.5, 1), (2, 2), (1, 2.5), (0, 2), (-0.5, 1)])   # 7 edges: worth a gate
    ext = lambda: b.Extrude(poly(), 1.0)
    sph = lambda: b.Translate(b.NewSphere(1), 4, 0, 0)

    def gate(sh):
        code = hip.lower(sh)[0]
        ins = decode(code)
        g = [i for i in ins if i[0].startswith("D_GATE")]
        f = code.view(np.float32)
        return ins, g, f, code

    # union: the cheap sphere first, the extrusion gated (sg = +1)
    ins, g, f, code = gate(b.Union(ext(), sph()))
    assert len(g) == 1 and g[0][0] == "D_GATE3D" and f[g[0][4] + 7] == 1.0 and f[g[0][4] + 8] == 0.0
    names = [i[0] for i in ins]
    assert names.index("D_SPHERE") < names.index("D_POLY2D")
    tgt = g[0][4] + skipw(code, g[0][4] + 12)
    assert at(ins, tgt, 0) == ["D_COMBINE_MIN"]
    # difference sphere - extrusion: the subtrahend is gated with sg = -1; extrusion - sphere: nothing (the minuend is the
    # result, and the sphere is not worth a test)
    ins, g, f, code = gate(b.Difference(sph(), ext()))
    assert len(g) == 1 and f[g[0][4] + 7] == -1.0
    assert [i[2] for i in ins if i[0] == "D_COMBINE_DIFF"] == [False]
    assert gate(b.Difference(ext(), sph()))[1] == []
    # smooth union: either order, gate carries 1.002 k; smooth difference: subtrahend only
    for sh in (b.SmoothUnion(0.3, ext(), sph()), b.SmoothUnion(0.3, sph(), ext())):
        ins, g, f, code = gate(sh)
        assert len(g) == 1 and f[g[0][4] + 7] == 1.0 and abs(f[g[0][4] + 8] - 1.002 * 0.3) < 1e-6
        assert at(ins, g[0][4] + skipw(code, g[0][4] + 12), 0) == ["D_COMBINE_SUNION"]
    ins, g, f, code = gate(b.SmoothDifference(0.3, sph(), ext()))
    assert len(g) == 1 and f[g[0][4] + 7] == -1.0 and abs(f[g[0][4] + 8] - 1.002 * 0.3) < 1e-6
    assert gate(b.SmoothDifference(0.3, ext(), sph()))[1] == []
    # a smooth union's own region is the hull of its operands grown by k / 4
    ins, g, f, code = gate(b.Union(b.Translate(b.SmoothUnion(0.4, ext(), b.NewSphere(1)), 10, 0, 0), sph()))
    outer = [q for q in g if at(ins, q[4] + skipw(code, q[4] + 12), 0) == ["D_COMBINE_MIN"]]
    assert len(outer) == 1
    x0, y0, z0, x1, y1, z1 = f[outer[0][4] + 1:outer[0][4] + 7]
    assert abs(x0 - (10 - 1 - 0.1)) < 1e-3 and abs(x1 - (10 + 2.5 + 0.1)) < 1e-3 and abs(z0 - (-1 - 0.1)) < 1e-3 and abs(y1 - (2.5 + 0.1)) < 1e-3
    # no gate where the child's value itself is needed
    for sh in (b.Intersection(sph(), ext()), b.Xor(sph(), ext()), b.SmoothIntersect(0.3, sph(), ext())):
        assert gate(sh)[1] == []


def test_gate_regions_of_screws_and_rotational_ops():
    """A screw is bounded from below outside its bounding cylinder (z-cylinder region, radial slope 1 / (1 + |tan taper|));
    twist and circular arrays rotate about z, so what they wrap is bounded outside the enclosing cylinder about the z axis."""
    b = Builder()
    code, _ = hip.lower(b.Scene("bolt"))
    ins = decode(code)
    f = code.view(np.float32)
    zc = [i for i in ins if i[0] == "D_GATEZC"]
    assert len(zc) == 1
    pc = zc[0][4]
    cx, cy, r, z0, z1, rs, rin, sg, kk = f[pc + 1:pc + 10]
    assert cx == 0 and cy == 0 and 1.5 < r < 1.56 and abs((z1 - z0) - 8.0) < 1e-3 and 0.99999 < rs < 1.0 and rin == 0 and sg == 1.0 and kk == 0.0
    assert at(ins, pc + skipw(code, pc + 13), 0) == ["D_COMBINE_MIN"]
    # npt-flange: tapered thread -> rs = 1 / (1 + 1/32)
    code, _ = hip.lower(b.Scene("npt-flange"))
    f = code.view(np.float32)
    pc = [i for i in decode(code) if i[0] == "D_GATEZC"][0][4]
    assert abs(f[pc + 6] - 32.0 / 33.0) < 1e-5
    # a twisted, circularly repeated box: bounded outside the cylinder through the box's farthest corner
    box = b.Translate(b.NewBox(1, 1, 4, 0), 3, 0, 0)
    knurl = b.Twist(b.CircularArray(box, 12, 12), 0.2)
    code, _ = hip.lower(b.Union(b.NewSphere(1), knurl))
    f = code.view(np.float32)
    g = [i for i in decode(code) if i[0] == "D_GATEZC"]
    assert len(g) == 1
    cx, cy, r, z0, z1, rs, rin = f[g[0][4] + 1:g[0][4] + 8]
    assert cx == 0 and cy == 0 and abs(r - np.hypot(3.5, 0.5)) < 1e-3 and abs(z0 + 2) < 1e-6 and abs(z1 - 2) < 1e-6 and rs == 1.0
    assert abs(rin - 2.5) < 1e-3                                                   # the boxes keep 2.5 away from the axis
    # knurled-cylinder: the knurl cutters are the subtrahend of a smooth difference that is itself the minuend of the next
    # one (the through hole, evaluated first): their gate carries that context (slot of the hole's value, 1.002 k, k / 4)
    code, _ = hip.lower(b.Scene("knurled-cylinder"))
    f = code.view(np.float32)
    ins = decode(code)
    gz = [i for i in ins if i[0] == "D_GATEZC"]
    ctx = [i for i in gz if skipw(code, i[4] + 10) != 0xffff]
    assert len(ctx) == 1
    pc = ctx[0][4]
    hole = [i for i in ins if i[0] == "D_SAVER"][0]                                 # the hole cylinder's value is saved first
    assert skipw(code, pc + 10) == hole[3] and f[pc + 8] == -1.0 and abs(f[pc + 9] - 1.002) < 1e-6
    assert abs(f[pc + 11] - 1.002) < 1e-6 and abs(f[pc + 12] - 0.25) < 1e-4 and 8.9 < f[pc + 7] < 8.94
    assert at(ins, pc + skipw(code, pc + 13), 0) == ["D_COMBINE_SDIFF"]


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


def test_sector_gate_of_circular_arrays():
    """A circular array evaluates its child in two neighbouring sectors and keeps the minimum (cpu_evaluators.go:1082-1090).
    Where the child is expensive (>= 150 estimated instructions) and has a (turned) box for a region, the lowering puts
    D_CIRC_ORDER behind D_CIRC_PRE (the wave starts with the copy nearer that box) and a D_GATEOB in front of the second copy,
    against the first one's value: the union rule, sg = 1, kk = 0, no context, skip target = the array's D_COMBINE_MIN."""
    b = Builder()
    star = b.NewPolygon([(1.2 * np.cos(t) * (1 if i % 2 else 0.5), 1.2 * np.sin(t) * (1 if i % 2 else 0.5)) for i, t in enumerate(np.linspace(0, 2 * np.pi, 12, endpoint=False))])
    tooth = b.Translate(b.Rotate(b.Extrude(star, 3.0), 0.5, (0, 0, 1)), 6.0, 0, 0)
    code, _ = hip.lower(b.Union(b.NewCylinder(5.5, 2.0, 0.0), b.CircularArray(tooth, 9, 9)))
    f = code.view(np.float32)
    ins = decode(code)
    names = [i[0] for i in ins]
    assert names.count("D_CIRC_PRE") == 1 and names.count("D_CIRC_ORDER") == 1 and names.count("D_GATEOB") == 1
    for k, i in enumerate(ins):
        if i[0] == "D_CIRC_PRE":
            assert ins[k + 1][0] == "D_CIRC_ORDER" and ins[k + 1][3] == i[3]         # same slots: p0 lives there
        if i[0] == "D_GATEOB":
            pc = i[4]
            cx, cy, c, s, hx, hy, z0, z1, sg, kk = f[pc + 1:pc + 11]
            assert abs(cx - 6) < 1e-4 and cy == 0 and abs(c - np.cos(0.5)) < 1e-5 and abs(abs(s) - np.sin(0.5)) < 1e-5   # the tooth's box, turned by 0.5 rad
            assert 1.0 < hx < 1.3 and 1.0 < hy < 1.3 and abs(z0 + 1.5) < 0.01 and abs(z1 - 1.5) < 0.01
            assert sg == 1.0 and kk == 0.0 and skipw(code, pc + 11) == 0xffff
            assert at(ins, pc + skipw(code, pc + 14), 0) == ["D_COMBINE_MIN"]
            assert ins[k - 1][0] == "D_LOADP3" and ins[k - 2][0] == "D_SAVER" and ins[k - 2][3] == i[3]  # a = the first copy's value
    # cheaper children (knurled-cylinder's cutter: a turned box, 95 instructions) or children without a box (a torus) get neither
    for tree in (b.Scene("knurled-cylinder"), b.CircularArray(b.Translate(b.NewTorus(1.0, 0.3), 3, 0, 0), 8, 8)):
        nm = [i[0] for i in decode(hip.lower(tree)[0])]
        assert "D_CIRC_ORDER" not in nm and "D_GATEOB" not in nm


def test_brick_mask_numbering():
    """Brick masks (dev_ops.h: D_SKIP / D_LIP_DOM): operand subtrees of combine frames under continuous position maps carry a
    number -- a D_SKIP in front of the subtree whose skip ends exactly where the subtree ends (the D_SAVER or, through an
    interval-mode-only D_LIP_DOM, the combine), a substitute on the side of the combine where a dominated operand lies, and a
    D_LIP_DOM that names both operands of its combine by role."""
    b = Builder()
    BIG = np.float32(1e30)
    for scene in ("npt-flange", "bolt", "knurled-cylinder"):
        code, _ = hip.lower(b.Scene(scene))
        ins = decode(code)
        f = code.view(np.float32)
        by_pc = {i[4]: k for k, i in enumerate(ins)}
        skips = [i for i in ins if i[0] == "D_SKIP"]
        ids = [int(code[i[4] + 1]) for i in skips]
        assert sorted(ids) == list(range(len(ids))) and 0 < len(ids) <= 16, (scene, ids)
        for i in skips:
            pc = i[4]
            assert abs(f[pc + 2]) == BIG
            end = pc + skipw(code, pc + 3)
            assert end in by_pc, (scene, pc)                      # an instruction boundary
            assert ins[by_pc[end]][0] in ("D_SAVER", "D_LIP_DOM") or ins[by_pc[end]][0].startswith("D_COMBINE"), (scene, ins[by_pc[end]][0])
            # a gate right behind a D_SKIP guards the same subtree and carries the same number
            nxt = ins[by_pc[pc] + 1]
            if nxt[0].startswith("D_GATE"):
                n = NPAR[NAMES.index(nxt[0])]
                w = int(code[nxt[4] + n])
                assert (w >> 24) - 1 == int(code[pc + 1]) and nxt[4] + (w & 0xffffff) == end
        # every D_LIP_DOM sits right in front of its combine, on the combine's slot, and names numbers that exist
        kinds = {"D_COMBINE_MIN": 0, "D_COMBINE_MAX": 1, "D_COMBINE_DIFF": 2, "D_COMBINE_SUNION": 3, "D_COMBINE_SDIFF": 4, "D_COMBINE_SINTER": 5}
        doms = [i for i in ins if i[0] == "D_LIP_DOM"]
        assert doms
        for i in doms:
            comb = ins[by_pc[i[4]] + 1]
            assert kinds[comb[0]] == int(code[i[4] + 1]) and comb[3] == i[3] and comb[2] == i[2]      # kind, slot, swap flag
            for q in (2, 3):
                v = int(code[i[4] + q])
                assert v == 0xff or v in ids
            # substitutes: a union's / smooth union's operands far above (+), an intersection's far below (-), a (smooth)
            # difference's minuend far below (-) and subtrahend far above (+)
            k = int(code[i[4] + 1])
            for role, q in (("a", 2), ("b", 3)):
                v = int(code[i[4] + q])
                if v == 0xff:
                    continue
                sub = f[[s[4] for s in skips if int(code[s[4] + 1]) == v][0] + 2]
                want = {0: BIG, 3: BIG, 1: -BIG, 5: -BIG}.get(k, -BIG if role == "a" else BIG)
                assert sub == want, (scene, k, role, sub)
    # nothing is numbered inside a position map that jumps: array cells, circular sectors, screw profiles
    two = lambda: b.Union(b.NewSphere(1), b.Translate(b.NewBox(1, 1, 1, 0), 1.5, 0, 0))
    for sh in (b.Array(two(), 4, 4, 4, 3, 3, 3), b.CircularArray(b.Translate(two(), 5, 0, 0), 6, 6)):
        names = [i[0] for i in decode(hip.lower(sh)[0])]
        assert names.count("D_SKIP") == 0 and names.count("D_LIP_DOM") == 0, names
    # ... and the frame around such a map is numbered as usual
    names = [i[0] for i in decode(hip.lower(b.Union(b.Array(two(), 4, 4, 4, 3, 3, 3), b.Translate(b.NewSphere(1), 20, 0, 0)))[0])]
    assert names.count("D_SKIP") == 2 and names.count("D_LIP_DOM") == 1
    # more than 16 candidates: the 16 most expensive ones
    wide = b.Union(*[b.Translate(b.NewSphere(1) if k % 2 else b.NewBox(1, 1, 1, 0.1), 3.0 * k, 0, 0) for k in range(24)])
    code = hip.lower(wide)[0]
    ins = decode(code)
    sk = [i for i in ins if i[0] == "D_SKIP"]
    assert len(sk) == 16
    follow = [ins[[j[4] for j in ins].index(i[4]) + 1:][:4] for i in sk]
    assert sum(any(j[0] == "D_BOX" for j in fl) for fl in follow) == 12   # all twelve boxes (45) before any sphere (25 + translate)
