"""Seeded random CSG trees for the composition tests: every operator of the builder with parameters in sane ranges.
Compositions (not single nodes) are what exercise the lowering's analyses -- slot allocation and reuse, child
reordering, the hypot cache, the corner-pair sharing flags."""
import math

import numpy as np

from scaffold.builder import Builder, ShapeError


def _prim3(b, r):
    k = r.integers(0, 9)
    u = lambda lo, hi: float(r.uniform(lo, hi))
    # threaded parts (forge/threads): expensive children with a z-cylinder lower bound -- what the skip gates are for
    if k == 7: return b.ScrewISO(u(0.6, 1.4), u(0.15, 0.4), bool(r.integers(0, 2)), u(0.5, 2.0))
    if k == 8: return b.ScrewNPT(float(r.choice([0.125, 0.25, 0.5])), u(0.3, 0.8))
    if k == 0: return b.NewSphere(u(0.3, 1.2))
    if k == 1: return b.NewBox(u(0.4, 1.5), u(0.4, 1.5), u(0.4, 1.5), u(0.0, 0.1))
    if k == 2: return b.NewCylinder(u(0.3, 1.0), u(0.5, 2.0), float(r.choice([0.0, 0.05])))
    if k == 3: return b.NewTorus(u(0.8, 1.2), u(0.1, 0.3))
    if k == 4: return b.NewHexagonalPrism(u(0.5, 1.2), u(0.4, 1.5))
    if k == 5: return b.Extrude(_shape2(b, r, 1), u(0.3, 1.5))
    return b.Revolve(b.Translate2D(b.NewCircle(u(0.1, 0.3)), 0.0, 0.0), u(0.4, 1.0))


def _prim2(b, r):
    k = r.integers(0, 7)
    u = lambda lo, hi: float(r.uniform(lo, hi))
    if k == 0: return b.NewCircle(u(0.2, 1.0))
    if k == 1: return b.NewRectangle(u(0.3, 1.5), u(0.3, 1.5))
    if k == 2: return b.NewHexagon(u(0.3, 1.0))
    if k == 3: return b.NewEllipse(u(0.4, 1.2), u(0.2, 0.8))
    if k == 4:
        n = int(r.integers(3, 9))
        ang = np.sort(r.uniform(0, 2 * math.pi, n))
        rad = r.uniform(0.4, 1.2, n)
        return b.NewPolygon([(float(c * math.cos(a)), float(c * math.sin(a))) for a, c in zip(ang, rad)])
    if k == 5: return b.NewEquilateralTriangle(u(0.4, 1.2))
    return b.NewOctagon(u(0.3, 1.0))


def _shape2(b, r, depth):
    if depth <= 0 or r.random() < 0.3:
        return _prim2(b, r)
    u = lambda lo, hi: float(r.uniform(lo, hi))
    k = r.integers(0, 10)
    a = _shape2(b, r, depth - 1)
    if k == 9:                                                                   # wide 2-D union (text-plate shape)
        parts = [a] + [b.Translate2D(_shape2(b, r, max(0, depth - 2)), u(-4, 4), u(-4, 4)) for _ in range(int(r.integers(3, 6)))]
        return b.Union2D(*parts)
    if k == 0: return b.Union2D(a, _shape2(b, r, depth - 1))
    if k == 1: return b.Difference2D(a, _shape2(b, r, depth - 1))
    if k == 2: return b.Intersection2D(a, b.Translate2D(_shape2(b, r, depth - 1), u(-0.2, 0.2), u(-0.2, 0.2)))
    if k == 3: return b.Translate2D(a, u(-0.5, 0.5), u(-0.5, 0.5))
    if k == 4: return b.Rotate2D(a, u(-3, 3))
    if k == 5: return b.Offset2D(a, u(-0.05, 0.1))
    if k == 6: return b.Symmetry2D(a, bool(r.integers(0, 2)), True)
    if k == 7: return b.Scale2D(a, u(0.5, 1.5))
    return b.CircularArray2D(b.Translate2D(a, u(1.0, 2.0), 0.0), int(r.integers(3, 7)), int(r.integers(3, 7)))


def _shape3(b, r, depth):
    if depth <= 0 or r.random() < 0.1:
        return _prim3(b, r)
    u = lambda lo, hi: float(r.uniform(lo, hi))
    k = r.integers(0, 20)
    a = _shape3(b, r, depth - 1)
    if k >= 18:   # a body with a cutter, then a cheap through hole: the cutter's gate gets the enclosing difference's context
        cut = b.Translate(_shape3(b, r, depth - 1), u(0.3, 0.9), u(-0.3, 0.3), u(-0.3, 0.3))
        hole = b.NewCylinder(u(0.1, 0.4), 6.0, 0.0) if r.random() < 0.6 else b.NewSphere(u(0.2, 0.5))
        if k == 18:
            kk = u(0.05, 0.3)
            return b.SmoothDifference(kk, b.SmoothDifference(u(0.05, 0.3), a, cut), hole)
        return b.Difference(b.Difference(a, cut), hole)
    if k >= 16:                                                                  # wide union: exercises the far-child skip
        parts = [a] + [b.Translate(_shape3(b, r, max(0, depth - 2)), u(-4, 4), u(-4, 4), u(-2, 2)) for _ in range(int(r.integers(3, 6)))]
        return b.Union(*parts)
    if k == 0: return b.Union(a, _shape3(b, r, depth - 1))
    if k == 1: return b.Union(a, _shape3(b, r, depth - 1), _shape3(b, r, depth - 2))
    if k == 2: return b.Difference(a, b.Translate(_shape3(b, r, depth - 1), u(-0.3, 0.3), u(-0.3, 0.3), u(-0.3, 0.3)))
    if k == 3: return b.Intersection(a, _shape3(b, r, depth - 1))
    if k == 4: return b.SmoothUnion(u(0.05, 0.4), a, _shape3(b, r, depth - 1))
    if k == 5: return b.SmoothDifference(u(0.05, 0.4), a, _shape3(b, r, depth - 1))
    if k == 6: return b.Translate(a, u(-0.6, 0.6), u(-0.6, 0.6), u(-0.6, 0.6))
    if k == 7: return b.Translate(a, 0.0, 0.0, u(-0.6, 0.6))                     # z-only: keeps hypot(x,y) and the xy sharing
    if k == 8: return b.Scale(a, u(0.5, 1.6))
    if k == 9: return b.Rotate(a, u(-3, 3), (0.0, 0.0, 1.0))
    if k == 10: return b.Rotate(a, u(-3, 3), (u(-1, 1), u(-1, 1), u(0.1, 1)))
    if k == 11: return b.Symmetry(a, bool(r.integers(0, 2)), bool(r.integers(0, 2)), True)
    if k == 12: return b.Twist(a, u(-0.5, 0.5))
    if k == 13: return b.Offset(a, u(-0.03, 0.08))
    if k == 14: return b.CircularArray(b.Translate(a, u(1.0, 2.0), 0.0, 0.0), int(r.integers(3, 7)), int(r.integers(3, 7)))
    return b.Xor(a, b.Translate(_shape3(b, r, depth - 1), u(0.2, 0.5), 0.0, 0.0))


def random_shapes(seed, count, depth=3):
    """`count` random 3-D shapes that the builder accepts (invalid parameter draws are skipped)."""
    r = np.random.default_rng(seed)
    b = Builder()
    out = []
    tries = 0
    while len(out) < count and tries < 20 * count:
        tries += 1
        try:
            sh = _shape3(b, r, depth)
            bb = sh.Bounds()
            if not np.isfinite(bb).all() or (bb[3:] - bb[:3]).max() > 50 or (bb[3:] - bb[:3]).min() <= 0:
                continue
            out.append(sh)
        except ShapeError:
            continue
    return b, out


def random_shapes2d(seed, count, depth=3):
    """`count` random 2-D shapes the builder accepts."""
    r = np.random.default_rng(seed)
    b = Builder()
    out = []
    tries = 0
    while len(out) < count and tries < 20 * count:
        tries += 1
        try:
            sh = _shape2(b, r, depth)
            bb = np.asarray(sh.Bounds(), np.float32)      # 2-D bounds come as (minx, miny, 0, maxx, maxy, 0)
            ext = bb[[3, 4]] - bb[[0, 1]]
            if not np.isfinite(bb).all() or ext.max() > 50 or ext.min() <= 0:
                continue
            out.append(sh)
        except ShapeError:
            continue
    return b, out
