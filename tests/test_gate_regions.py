"""The lower-bound regions behind the device's skip gates (gsdf_amd/csrc/compile.cpp: lower_region, D_GATE* in dev_ops.h)
checked against the oracle, host-only: for every node of every corpus / scene / random tree for which the lowering claims a
region, the subtree's field (the oracle evaluating the tree from that node) must be >= the region's lower bound at every
sampled point outside it -- with the slack the gate itself keeps (1e-3 relative + 2e-6 of the coordinates)."""
import ctypes as C

import numpy as np
import pytest

import corpus
import fuzz_trees
from gsdf_amd import hip
from gsdf_amd._ctypes_common import GsdfTree, OPS
from scaffold.builder import Builder
from oracle.oracle import OracleSDF

FIRST_2D = OPS.index("LINE2D")


def _reachable(t):
    seen, stack = [], [int(t.root)]
    mark = set()
    while stack:
        i = stack.pop()
        if i in mark:
            continue
        mark.add(i)
        seen.append(i)
        nd = t.nodes[i]
        for k in range(nd.nchild):
            stack.append(int(t.links[nd.link_off + k]))
    return seen


def _subtree(t, node):
    s = GsdfTree()
    C.memmove(C.byref(s), C.byref(t), C.sizeof(GsdfTree))
    s.root = node
    s._keepalive = t
    return s


def _region(t, node):
    kind, par = C.c_int(), (C.c_float * 8)()
    assert hip.lib().gsdf_hip_lower_region(C.byref(t), node, C.byref(kind), par) == 0, hip.lib().gsdf_hip_last_error()
    return kind.value, np.array(par[:], np.float64)


def _lower_bound(kind, g, pos, is2d):
    x, y = pos[:, 0].astype(np.float64), pos[:, 1].astype(np.float64)
    z = np.zeros_like(x) if is2d else pos[:, 2].astype(np.float64)
    if kind == 1:
        L = np.maximum(np.maximum(g[0] - x, x - g[3]), np.maximum(g[1] - y, y - g[4]))
        if not is2d:
            L = np.maximum(L, np.maximum(g[2] - z, z - g[5]))
        return L
    if kind == 3:                      # box turned about z: cx cy c s hx hy z0 z1 (Chebyshev distance in its own frame)
        dx, dy = x - g[0], y - g[1]
        L = np.maximum(np.abs(g[2] * dx + g[3] * dy) - g[4], np.abs(g[2] * dy - g[3] * dx) - g[5])
        return np.maximum(L, np.maximum(g[6] - z, z - g[7]))
    rad = np.hypot(x - g[0], y - g[1])
    L = g[5] * (rad - g[2])
    if not is2d:
        L = np.maximum(L, np.maximum(g[3] - z, z - g[4]))
    if g[6] > 0:                       # annulus: the shape also keeps rin away from the axis, at any z
        L = np.maximum(L, g[5] * (g[6] - rad))
    return L


def check_tree(t, rng, npts=3000):
    """Returns the number of (node, point) claims checked."""
    checked = 0
    for node in _reachable(t):
        kind, g = _region(t, node)
        if kind == 0:
            continue
        is2d = t.nodes[node].op >= FIRST_2D
        if kind == 1:
            lo, hi = g[:3].copy(), g[3:6].copy()
        elif kind == 3:
            r = float(np.hypot(g[4], g[5]))
            lo, hi = np.array([g[0] - r, g[1] - r, g[6]]), np.array([g[0] + r, g[1] + r, g[7]])
        else:
            lo = np.array([g[0] - g[2], g[1] - g[2], max(g[3], -1e3)])
            hi = np.array([g[0] + g[2], g[1] + g[2], min(g[4], 1e3)])
        ext = np.maximum(hi - lo, 1e-3)
        ext[2] = max(ext[2], 1e-3)
        ctr = (lo + hi) / 2
        # points in a shell around the region: within half its size, and a far set
        pos = np.concatenate([ctr + (rng.random((npts, 3)) - 0.5) * ext * 2.0, ctr + (rng.random((npts // 3, 3)) - 0.5) * ext * 8.0]).astype(np.float32)
        if is2d:
            pos = np.ascontiguousarray(pos[:, :2])
        L = _lower_bound(kind, g, pos, is2d)
        out = L > 0
        if not out.any():
            continue
        val = OracleSDF(_subtree(t, node)).Evaluate(pos).astype(np.float64)
        S = np.abs(pos.astype(np.float64)).sum(axis=1)
        slack = 1e-3 * L + 2e-6 * S
        bad = out & ~(val >= L - slack)
        assert not bad.any(), (OPS[t.nodes[node].op], node, kind, g.tolist(), pos[bad][:3].tolist(), val[bad][:3].tolist(), L[bad][:3].tolist())
        checked += int(out.sum())
    return checked


def test_regions_of_the_example_scenes_hold():
    b = Builder()
    rng = np.random.default_rng(5)
    for name in ("npt-flange", "bolt", "knurled-cylinder", "glyph-plate"):
        t = b.Scene(name).tree()
        assert check_tree(t, rng) > 1000, name


def test_regions_of_the_corpus_hold():
    rng = np.random.default_rng(6)
    n = 0
    b3, s3 = corpus.shapes3d()
    b2, s2 = corpus.shapes2d()
    for name, sh in s3 + s2:
        n += check_tree(sh.tree(), rng, 1500)
    assert n > 50000


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_regions_of_random_trees_hold(seed):
    rng = np.random.default_rng(seed)
    _, shapes = fuzz_trees.random_shapes(seed, 40, depth=3)
    _, shapes2 = fuzz_trees.random_shapes2d(seed, 25, depth=3)
    n = sum(check_tree(sh.tree(), rng, 800) for sh in shapes + shapes2)
    assert n > 20000


def test_screw_and_rotational_regions():
    """The new kinds of claim, on their own: screws (straight and tapered), twisted / circularly repeated parts, smooth
    unions (hull grown by k / 4), negative offsets."""
    b = Builder()
    rng = np.random.default_rng(7)
    flange = b.Scene("npt-flange").tree()
    kinds = {OPS[flange.nodes[i].op]: _region(flange, i)[0] for i in _reachable(flange)}
    assert kinds["SCREW"] == 2 and kinds["SMOOTH_UNION"] == 1 and kinds["DIFF"] == 1
    box = b.Translate(b.NewBox(1, 1, 4, 0), 3, 0, 0)
    parts = [b.Twist(b.CircularArray(box, 12, 12), 0.2), b.Twist(box, -0.4), b.CircularArray(box, 5, 7),
             b.SmoothUnion(0.5, b.NewSphere(1), b.Translate(b.NewBox(1, 2, 1, 0.1), 1.5, 0, 0)),
             b.Offset(b.NewBox(1, 2, 1, 0), -0.3), b.Offset(b.NewSphere(1), 0.2),
             b.Symmetry(b.Translate(b.NewSphere(0.5), 1, 2, 0.5), True, False, True),
             b.Rotate(b.Twist(box, 0.3), 0.7, (0, 0, 1))]
    for sh in parts:
        t = sh.tree()
        assert _region(t, int(t.root))[0] != 0, sh
        assert check_tree(t, rng, 4000) > 2000, sh
    # boxes on a circle around the axis: an annulus -- also bounded from below near the axis (what lets knurled-cylinder's
    # through hole skip the knurl cutters)
    t = b.CircularArray(box, 12, 12).tree()
    k, g = _region(t, int(t.root))
    assert k == 2 and abs(g[6] - 2.5) < 1e-3 and abs(g[2] - np.hypot(3.5, 0.5)) < 1e-3
    t = b.Scene("knurled-cylinder").tree()
    rins = [_region(t, i)[1][6] for i in _reachable(t) if OPS[t.nodes[i].op] in ("TWIST", "CIRCARRAY")]
    assert rins and all(8.9 < r < 8.94 for r in rins)
    # 2-D circular arrays and annuli carry an "unbounded" z range; turned (Rotate2D) or scaled they keep their radii instead of
    # being padded by it (the pad was 2.4e34, the gate could never fire)
    r2 = b.Translate2D(b.NewRectangle(1.0, 0.6), 3.0, 0.0)
    for sh2 in (b.Rotate2D(b.CircularArray2D(r2, 8, 8), 0.4), b.Scale2D(b.CircularArray2D(r2, 8, 8), 1.5),
                b.Rotate2D(b.Scale2D(b.CircularArray2D(r2, 6, 6), 0.5), -1.1)):
        t = sh2.tree()
        k, g = _region(t, int(t.root))
        assert k == 2 and 1.0 < g[2] < 6.0 and g[6] > 0.5 and g[3] <= -3e38 and g[4] >= 3e38, (k, g)
        assert check_tree(t, rng, 3000) > 1000
    # a z-cylinder does not survive a rotation about another axis: no claim
    t = b.Rotate(b.Twist(box, 0.3), 0.7, (1, 0, 0)).tree()
    assert _region(t, int(t.root))[0] == 0


def test_cyclic_node_graph_is_refused():
    """A malformed blob whose links form a cycle must be rejected by validation, not recursed into."""
    b = Builder()
    t = b.Union(b.NewSphere(1), b.NewSphere(2)).tree()
    links = (C.c_uint32 * t.n_links)(*[t.links[i] for i in range(t.n_links)])
    root = int(t.root)
    links[t.nodes[root].link_off] = root   # the union's first child is the union itself
    bad = _subtree(t, root)
    bad.links = C.cast(links, C.POINTER(C.c_uint32))
    bad._links = links
    n, sl = C.c_uint32(), C.c_uint32()
    assert hip.lib().gsdf_hip_lower(C.byref(bad), None, 0, C.byref(n), C.byref(sl)) != 0
    assert b"cycle" in hip.lib().gsdf_hip_last_error()


def test_turned_boxes_keep_their_orientation():
    """A rotation about z makes a box a TURNED box (kind 3) instead of its sqrt-2-wider axis-aligned hull: what lets a circular
    array skip the farther of its two sector copies (knurled-cylinder's cutters: a 45-degree box 16 from the axis, 24 sectors
    of 15 degrees). The claim is checked like every other; nested rotations, translations, scalings and offsets keep the kind,
    mirrors / unions / twists fall back to hulls that still hold."""
    b = Builder()
    rng = np.random.default_rng(17)
    box = b.NewBox(2.0, 1.0, 3.0, 0.0)
    t1 = b.Translate(b.Rotate(box, 0.6, (0, 0, 1)), 4.0, -1.0, 0.5)
    k, g = _region(t1.tree(), t1.tree().root)
    assert k == 3 and abs(g[0] - 4.0) < 1e-5 and abs(np.hypot(g[2], g[3]) - 1) < 1e-6 and abs(abs(g[2]) - np.cos(0.6)) < 1e-5
    shapes = [t1,
              b.Rotate(b.Translate(b.Rotate(box, 0.6, (0, 0, 1)), 4.0, -1.0, 0.5), -1.1, (0, 0, 1)),
              b.Scale(b.Offset(t1, -0.2), 1.7),
              b.Rotate(box, np.pi / 2, (0, 0, 1)),                      # a quarter turn stays axis-aligned (kind 1)
              b.Union(t1, b.NewSphere(1.0)), b.Symmetry(t1, True, False, False), b.Twist(t1, 0.2),
              b.CircularArray(b.Translate(b.Rotate(b.NewBox(3, 3, 8, 0), np.pi / 4, (0, 0, 1)), 6, 0, 0), 12, 12),
              b.Rotate(t1, 0.8, (1, 0, 0.2))]
    kinds = [_region(s.tree(), s.tree().root)[0] for s in shapes]
    assert kinds[:4] == [3, 3, 3, 1] and kinds[4] == 1 and kinds[5] == 1 and kinds[6] == 2 and kinds[7] == 2 and kinds[8] == 1
    assert sum(check_tree(s.tree(), rng, 2000) for s in shapes) > 10000
