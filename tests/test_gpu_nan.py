"""Where the device may differ from the reference in the presence of NaN, stated as a relation and asserted.

The reference's helpers take minima and maxima with math32.Min / Max (gsdf.go:141-189 and every combine of
cpu_evaluators.go), which return NaN when either operand is NaN; the device uses v_min_f32 / v_max_f32 (and v_med3_f32 for
clamps), which return the other operand. A NaN can only arise from degenerate parameters (a polygon edge of length zero, a
line whose ends coincide, a blend width or scale of zero, a circle posing as an ellipse, a bezier whose control point is the
chord's midpoint) or from non-finite positions; the reference's constructors refuse most of these, a tree handed to the C ABI
need not have come from them. The relation, checked here on such trees and positions with the interpreter and with the
specialised kernels:

    finite positions: the device's distance equals the reference's, bit for bit, wherever the reference's is not NaN;
    where the reference's is NaN the device returns NaN or the value the same formulas give with the NaN operand of a
    min / max / clamp dropped -- a stand-in, never a trap;
    non-finite positions: equal wherever the reference's distance is finite (math32.Hypot / Atan2 have Inf / NaN special
    cases of their own which the device's forms do not reproduce).

The ordinary parity tests mask positions where BOTH sides are NaN (tests/test_gpu_eval.py: _mismatch); nothing is masked here.
"""
import ctypes as C

import numpy as np
import pytest

from par import pmap

import corpus
from gsdf_amd._ctypes_common import GsdfNode, GsdfTree, OP
from oracle.oracle import OracleSDF
from scaffold.builder import Builder
from tree_edit import clone, first

pytestmark = pytest.mark.gpu


def degenerate_trees():
    b = Builder()
    out = []
    # polygon with a repeated vertex: an edge of length zero, 0 / 0 in its projection (cpu_evaluators.go:803-806)
    t = clone(b.Union(b.Extrude(b.NewPolygon([(0, 0), (2, 0), (2, 1), (1, 1.5), (0, 1)]), 1.0), b.Translate(b.NewSphere(0.4), 1, 1, 0.6)).tree())
    n = t.nodes[first(t, "POLY2D")]
    t.aux[n.aux_off + 4], t.aux[n.aux_off + 5] = t.aux[n.aux_off + 2], t.aux[n.aux_off + 3]
    out.append(("polygon-repeated-vertex", t))
    # line whose ends coincide (NewLine2D turns it into a circle, primitives2d.go:24-29; the node itself divides by 0)
    t = clone(b.Extrude(b.NewLine2D(0.2, 0.1, 1.0, 0.5, 0.2), 0.5).tree())
    n = t.nodes[first(t, "LINE2D")]
    n.p[2], n.p[3] = n.p[0], n.p[1]
    out.append(("line-of-length-zero", t))
    # blend width zero: (b - a) / 0 is NaN where the two fields are equal, +-Inf elsewhere
    for name, mk in (("smooth-union-k0", b.SmoothUnion), ("smooth-diff-k0", b.SmoothDifference), ("smooth-intersect-k0", b.SmoothIntersect)):
        t = clone(mk(0.1, b.NewSphere(1.0), b.Translate(b.NewSphere(1.0), 1.0, 0, 0)).tree())
        t.nodes[t.root].p[0] = 0.0
        out.append((name, t))
    # scale by zero: positions times Inf (NaN on the axis planes), distance times 0
    t = clone(b.Union(b.Scale(b.NewBox(1, 1, 1, 0.1), 2.0), b.Translate(b.NewSphere(0.5), 2, 0, 0)).tree())
    t.nodes[first(t, "SCALE")].p[0] = 0.0
    out.append(("scale-zero", t))
    # an "ellipse" with equal axes: l = b^2 - a^2 = 0 (cpu_evaluators.go:759-762)
    t = clone(b.Extrude(b.NewEllipse(1.0, 0.5), 0.4).tree())
    n = t.nodes[first(t, "ELLIPSE2D")]
    n.p[1] = n.p[0]
    out.append(("ellipse-circle", t))
    # bezier whose control point is the midpoint of its chord: kk = 1 / 0 (cpu_evaluators.go:585-593)
    t = clone(b.Extrude(b.NewQuadraticBezier2D((0, 0), (1, 1.5), (2, 0), 0.1), 0.4).tree())
    n = t.nodes[first(t, "QUADBEZIER2D")]
    n.p[2], n.p[3] = 1.0, 0.0
    out.append(("bezier-straight", t))
    # shell of thickness zero: positions times Inf
    t = clone(b.Shell(b.NewBox(1, 1, 1, 0.1), 0.1).tree())
    t.nodes[first(t, "SHELL")].p[0] = 0.0
    out.append(("shell-zero", t))
    return out


def positions(bb, rng, n=4000):
    c, h = (bb[:3] + bb[3:]) / 2, np.maximum((bb[3:] - bb[:3]) / 2, 0.25) * np.float32(1.3)
    p = (c + (rng.random((n, 3), np.float32) * 2 - 1) * h).astype(np.float32)
    g = np.float32(0.25) * rng.integers(-8, 9, (n // 2, 3)).astype(np.float32)   # lattice: exact zeros, ties, symmetric pairs
    return np.concatenate([p, g]).astype(np.float32)


def relation(dev, ref, what, nonfinite=False):
    dev, ref = np.asarray(dev, np.float32), np.asarray(ref, np.float32)
    differ = dev.view(np.uint32) != ref.view(np.uint32)
    refnan = ~np.isfinite(ref) if nonfinite else np.isnan(ref)
    bad = differ & ~refnan
    assert not bad.any(), (what, int(bad.sum()), np.flatnonzero(bad)[:5].tolist(), dev[bad][:5].tolist(), ref[bad][:5].tolist())
    return int(refnan.sum()), int((refnan & ~np.isnan(dev)).sum())


def test_degenerate_trees_differ_only_where_the_reference_is_nan(gpu):
    rng = np.random.default_rng(41)
    seen_nan = 0
    for name, t in degenerate_trees():
        ref = OracleSDF(t)
        pos = positions(np.array(t.bb[:], np.float32), rng)
        dref = ref.Evaluate(pos)
        sdf = gpu.SDFHIP(t)
        n_nan, n_standin = relation(sdf.Evaluate(pos), dref, (name, "interpreter"))
        sdf.specialize()
        n2, _ = relation(sdf.Evaluate(pos), dref, (name, "specialised"))   # (built with -fno-honor-nans: its stand-ins may differ from the interpreter's)
        assert n2 == n_nan, name
        seen_nan += n_nan
    assert seen_nan > 100                                # the family does produce NaNs in the reference


# Nodes that take the SIGN of a value (ms1.Sign / math32.Copysign): of a NaN that is the NaN's sign bit, and the NaN that
# Inf - Inf or 0 * Inf yields is negative on amd64, positive on arm64 and on the GPU -- there the reference itself differs
# from one host CPU to the next, so trees with these nodes are left out of the non-finite-position relation.
SIGN_OF_NAN = {"ARRAY", "ARRAY2D", "HEX", "EQTRI2D", "DIAMOND2D", "HEX2D", "OCT2D", "ELLIPSE2D", "QUADBEZIER2D"}


def test_non_finite_positions_differ_only_where_the_reference_is_not_finite(gpu):
    """NaN / Inf coordinates into ordinary trees (every node type of the corpus and the benchmark scenes). Besides
    Min / Max, math32.Hypot and Atan2 have Inf / NaN special cases (Hypot(Inf, NaN) = +Inf ...) that the device's forms do not
    reproduce: here the two sides may differ wherever the reference's result is NaN or +-Inf, and nowhere else."""
    from gsdf_amd._ctypes_common import OPS
    rng = np.random.default_rng(43)
    b3, s3 = corpus.shapes3d()
    shapes = list(s3) + [(n, b3.Scene(n)) for n in ("npt-flange", "bolt", "knurled-cylinder")]
    cases = []
    for k, (name, sh) in enumerate(shapes):
        t = sh.tree()
        ops, todo = set(), [t.root]
        while todo:                                         # the nodes this shape reaches (the blob holds the whole builder)
            nd = t.nodes[todo.pop()]
            ops.add(OPS[nd.op])
            todo += [t.links[nd.link_off + c] for c in range(nd.nchild)]
        pos = positions(np.array(t.bb[:], np.float32), rng, 1500)
        bad = rng.integers(0, len(pos), 600)
        vals = np.float32([np.nan, np.inf, -np.inf, 0.0, -0.0])   # (not 3e38: finite, but squares overflow differently along the two paths)
        pos[bad, rng.integers(0, 3, 600)] = vals[rng.integers(0, len(vals), 600)]
        cases.append((k, name, t, ops, pos))

    def check(case):                                        # (every fourth tree is also built: side by side, tests/par.py)
        k, name, t, ops, pos = case
        failures = []
        dref = OracleSDF(t).Evaluate(pos)
        sdf = gpu.SDFHIP(t)
        builds = [("interpreter", sdf.Evaluate(pos))]
        if k % 4 == 0:
            sdf.specialize()
            builds.append(("specialised", sdf.Evaluate(pos)))
        for what, dev in builds:
            differ = (dev.view(np.uint32) != dref.view(np.uint32)) & np.isfinite(dref)
            if differ.any() and not (ops & SIGN_OF_NAN):
                i = int(np.flatnonzero(differ)[0])
                failures.append((name, what, int(differ.sum()), pos[i].tolist(), float(dev[i]), float(dref[i])))
        return failures, (0 if (ops & SIGN_OF_NAN) else 1), int((~np.isfinite(dref)).sum())
    res = pmap(check, cases, workers=8)
    failures = [f for r in res for f in r[0]]
    checked, total = sum(r[1] for r in res), sum(r[2] for r in res)
    assert not failures, failures
    assert total > 1000 and checked >= 25


POSITIONS_TIMES_INF = {"scale-zero", "shell-zero"}


def _sorted_bits(t):
    t = np.ascontiguousarray(t, np.float32).reshape(-1, 9).view(np.uint32)
    return t[np.lexsort(t.T[::-1])]


def test_degenerate_trees_decisions_are_ieee(gpu):
    """What DECIDES in the meshers and the image renderer -- a cube kept or dropped, a leaf marched or not, a corner inside or
    outside, a pixel flagged as not finite -- is a class / integer test on the value's bits (kernels_common.h: nb::), not a float
    comparison: the specialised kernels are built with -fno-honor-nans, under which a comparison with a NaN operand is whatever is
    cheapest (`!(|d| >= m)` had come out as `|d| < m` and dropped the cubes the reference keeps; `v != v` was folded away).
    The reference's decisions on a NaN: the cube is kept (|d| >= maxDist is false, octreerenderer.go:270-273), the leaf is not
    marched (|d0| <= cubeDiag is false, marchcubes.go:20-23), the corner is outside (d < 0 is false).
    Checked on the degenerate trees, whose fields hold NaN:
      interpreter kernels (strict build: a value does not depend on which kernel computed it): the flat renderer's active cubes
        and triangle count are IEEE's decisions on Evaluate's distances over its lattice; the centre tests lose no surface;
      specialised kernels: where the reference returns NaN the stand-in may differ from kernel to kernel of that build (first test
        of this file), so only what one kernel decides about its OWN values can be held against it -- the image renderer's red
        pixels are exactly the non-finite distances it returned; and where the build's Evaluate agrees with the interpreter's bit
        for bit on the lattice and around it, its meshes are the interpreter's."""
    import os
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gsdf_amd", "csrc", "mc_tables.h")).read()
    ntri = np.array([int(x) for x in re.search(r"GSDF_MC_NTRI\[256\]\s*=\s*\{([^}]*)\}", hdr).group(1).replace("\n", " ").split(",") if x.strip()], np.int64)
    assert ntri.shape == (256,) and ntri[0] == 0 and ntri[255] == 0 and ntri.max() == 5
    rng = np.random.default_rng(47)
    agreeing, checked, nan_lattices = [], 0, 0
    for name, t in degenerate_trees():
        bb = np.array(t.bb[:], np.float32)
        res = np.float32(float(np.linalg.norm(bb[3:] - bb[:3])) / 48)
        a, b = gpu.SDFHIP(t), gpu.SDFHIP(t)
        b.specialize()
        # the flat renderer's lattice (FlatRenderer.Reset, flatrenderer.go:36-80: the 1.01-scaled bounds, ceil(size / res) cubes per
        # axis, corner i at O + res * i)
        c = np.float32(0.5) * (bb[:3] + bb[3:])
        half = np.float32(0.5) * (np.float32(1.01) * (bb[3:] - bb[:3]))
        o, mx = (c - half).astype(np.float32), (c + half).astype(np.float32)
        nc = [int(np.ceil((mx[k] - o[k]) / res)) for k in range(3)]
        ax = [(o[k] + res * np.arange(nc[k] + 1, dtype=np.float32)).astype(np.float32) for k in range(3)]
        g = np.stack(np.meshgrid(*ax, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
        cube_diag = np.float32(2) * np.float32(1.73205080757) * res
        dists = {}
        for what, sdf in (("interpreter", a),):
            d = sdf.Evaluate(g).reshape(nc[0] + 1, nc[1] + 1, nc[2] + 1)
            dists[what] = d
            nan_lattices += int(np.isnan(d).any())
            # 1. the flat renderer's decisions are IEEE's on THIS build's own distances: a cube is marched iff |d(corner 0)| <=
            #    2 sqrt3 res (false for NaN), a corner is inside iff d < 0 (false for NaN); marchcubes.go:20-40
            with np.errstate(invalid="ignore"):
                act = np.abs(d[:-1, :-1, :-1]) <= cube_diag
                inside = d < 0
            corner = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
            idx = np.zeros(act.shape, np.int64)
            for k, (i, j, l) in enumerate(corner):
                idx |= inside[i:i + nc[0], j:j + nc[1], l:l + nc[2]].astype(np.int64) << k
            fl = gpu.FlatHIP(sdf, res)
            assert fl.stats.leaf_cubes == nc[0] * nc[1] * nc[2], (name, what)
            assert fl.stats.active_leaves == int(act.sum()) and fl.n_tris() == int(ntri[idx[act]].sum()), (name, what)
            # 2. the centre tests drop no cube that holds surface, NaN or not (a NaN centre value or bound keeps the cube, as the
            #    reference's `|d| >= maxDist` does): pruned == unpruned, bit for bit. Held for the trees whose NaNs come out of
            #    a primitive or a blend; a scale / shell factor of zero multiplies the POSITIONS by Inf (and the cube's radius by
            #    Inf, then by 0): there the point and the interval evaluation are two different piles of Inf - Inf.
            full = _sorted_bits(gpu.OctreeHIP(sdf, res, prune=False).RenderAll())
            if name not in POSITIONS_TIMES_INF:
                got = gpu.OctreeHIP(sdf, res)
                assert got.n_tris() == len(full) and (_sorted_bits(got.RenderAll()) == full).all(), (name, what)
            checked += len(full)
        # 3. where the two builds' distances agree bit for bit, so does everything derived from them
        pos = np.concatenate([g, positions(bb, rng, 20000)])
        if (a.Evaluate(pos).view(np.uint32) == b.Evaluate(pos).view(np.uint32)).all():
            agreeing.append(name)
            for mk in (lambda s_: gpu.OctreeHIP(s_, res), lambda s_: gpu.OctreeHIP(s_, res, assume_sdf=True), lambda s_: gpu.FlatHIP(s_, res)):
                ma, mb = mk(a), mk(b)
                assert ma.n_tris() == mb.n_tris() and ma.TotalPruned() == mb.TotalPruned() and ma.stats.active_leaves == mb.stats.active_leaves, name
                assert (_sorted_bits(ma.RenderAll()) == _sorted_bits(mb.RenderAll())).all(), name
    assert agreeing and checked > 5000 and nan_lattices >= 1, (agreeing, checked, nan_lattices)
    # 2-D: the same nodes under the image renderer (NaN / Inf pixels are red, image.go:104-112)
    b2 = Builder()
    trees = []
    t = clone(b2.NewLine2D(0.2, 0.1, 1.0, 0.5, 0.2).tree())
    n = t.nodes[first(t, "LINE2D")]
    n.p[2], n.p[3] = n.p[0], n.p[1]
    trees.append(("line-of-length-zero", t))
    t = clone(b2.NewEllipse(1.0, 0.5).tree())
    n = t.nodes[first(t, "ELLIPSE2D")]
    n.p[1] = n.p[0]
    trees.append(("ellipse-circle", t))
    t = clone(b2.NewPolygon([(0, 0), (2, 0), (2, 1), (1, 1.5), (0, 1)]).tree())
    n = t.nodes[first(t, "POLY2D")]
    t.aux[n.aux_off + 4], t.aux[n.aux_off + 5] = t.aux[n.aux_off + 2], t.aux[n.aux_off + 3]
    trees.append(("polygon-repeated-vertex", t))
    red = same_images = 0
    for name, t in trees:
        a, b = gpu.SDFHIP(t), gpu.SDFHIP(t)
        b.specialize()
        (da, ia), (db, ib) = a.render_image(96, 64), b.render_image(96, 64)
        for d_, i_ in ((da, ia), (db, ib)):                # each build: red exactly where ITS distance is not finite
            bad = ~np.isfinite(d_).reshape(-1)
            px = np.asarray(i_).reshape(-1, 4)
            assert (px[bad] == [255, 0, 0, 255]).all() and not (px[~bad] == [255, 0, 0, 255]).all(axis=1).any(), name
            red += int(bad.sum())
        if (da.view(np.uint32) == db.view(np.uint32)).all():
            assert (np.asarray(ia) == np.asarray(ib)).all(), name
            same_images += 1
    assert red > 0 and same_images >= 1
