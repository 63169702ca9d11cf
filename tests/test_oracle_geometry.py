"""Geometric known-answer tests that do not look at the restatement's formulas: what an exact signed distance field MUST satisfy,
checked on the oracle for every primitive (3-D and 2-D) and every operation that keeps a field exact.
  1. eikonal: |grad d| = 1 wherever d is differentiable (a wrong constant in a folded primitive -- the hexagon's k, the triangle's
     sqrt 3, the octagon's tangents -- bends the field: the norm leaves 1 on whole regions);
  2. closest point: q = p - d(p) grad d(p) lies ON the surface: d(q) = 0 (a wrong offset, radius or half-height moves the zero set
     away from where the field says it is);
  3. extent: the surface reached that way spans exactly the node's Bounds() -- which the scaffold computes from the constructor's
     documented dimensions (primitives.go / primitives2d.go Bounds methods), not from the evaluator: "h is the half height",
     "e was halved by the builder", "face to face, not side" kinds of mistakes show here.
Together with the closed-form values of test_oracle_golden.py::test_analytic_kats and the reference's own GLSL statement
(test_glsl_crosscheck.py) this is what stands behind the oracle where the reference holds no triangle count."""
import math
import zlib
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from scaffold.builder import Builder  # noqa: E402
from oracle.oracle import OracleSDF  # noqa: E402


def _shapes():
    b = Builder()
    octv = [(math.cos(2 * math.pi * i / 8), math.sin(2 * math.pi * i / 8)) for i in range(8)]
    star = [(math.cos(math.pi * i / 5) * (1.0 if i % 2 == 0 else 0.45), math.sin(math.pi * i / 5) * (1.0 if i % 2 == 0 else 0.45)) for i in range(10)]
    box = b.NewBox(1, 0.61, 0.8, 0.1)
    rect = b.NewRectangle(1, 0.61)
    prim3 = [("sphere", b.NewSphere(0.7)), ("box", b.NewBox(1, 0.47, 0.8, 0)), ("box_round", box), ("boxframe", b.NewBoxFrame(1, 0.47, 0.8, 0.1)),
             ("cylinder", b.NewCylinder(0.5, 0.9, 0)), ("cylinder_round", b.NewCylinder(0.5, 0.9, 0.12)), ("hexprism", b.NewHexagonalPrism(1, 0.47)),
             ("torus", b.NewTorus(1, 0.3)), ("triprism", b.NewTriangularPrism(1, 0.5))]
    prim2 = [("circle", b.NewCircle(0.8)), ("line", b.NewLine2D(-0.3, 0.1, 0.9, 0.47, 0.2)), ("rect", rect), ("arc", b.NewArc(1, math.pi / 3, 0.1)),
             ("arc_wide", b.NewArc(0.8, 4.0, 0.16)), ("hexagon", b.NewHexagon(1)), ("octagon", b.NewOctagon(1)), ("eqtri", b.NewEquilateralTriangle(1)),
             ("ellipse", b.NewEllipse(1, 2)), ("ellipse_flat", b.NewEllipse(1.7, 0.6)), ("poly_octagon", b.NewPolygon(octv)), ("poly_star", b.NewPolygon(star)),
             ("lines", b.NewLines2D([(octv[i - 1], octv[i]) for i in range(1, 6)], 0.1)), ("diamond", b.NewDiamond2D(1, 0.47)), ("roundedx", b.NewRoundedX(1, 0.1)),
             ("bezier", b.NewQuadraticBezier2D((1.0, 0.47), (2.0, 0.47), (1.0, 1.47), 0.1)), ("iso_thread", b.ISOThread(1, 0.1, True))]
    ops = [("translate", b.Translate(box, 0.5, -0.7, 0.8)), ("rotate", b.Rotate(box, 0.7, (0.3, 1.0, -0.4))), ("scale", b.Scale(box, 1.7)),
           ("elongate", b.Elongate(box, 0.2, 0.0, 0.35)), ("shell", b.Shell(b.NewSphere(0.8), 0.05)), ("extrude", b.Extrude(b.NewPolygon(star), 0.4)),
           ("revolve", b.Revolve(b.Translate2D(rect, 1.2, 0), 0)), ("revolve_off", b.Revolve(rect, 0.9)),
           ("translate2d", b.Translate2D(rect, 2, 0.3)), ("rotate2d", b.Rotate2D(rect, 0.6)), ("scale2d", b.Scale2D(rect, 0.4)),
           ("elongate2d", b.Elongate2D(b.NewCircle(0.3), 0.5, 0.2)), ("annulus", b.Annulus(b.NewCircle(0.7), 0.08)),
           ("circarray", b.CircularArray(b.Translate(b.NewSphere(0.2), 1.0, 0, 0), 5, 5)), ("array", b.Array(b.NewSphere(0.2), 0.9, 1.1, 1.3, 2, 3, 2))]
    return b, prim3 + prim2 + ops


_B, _SHAPES = _shapes()

# Where the reference's own Bounds() is NOT the zero set's extent (its Bounds methods, mirrored by the scaffold; the evaluators and the
# GLSL agree with each other on all of these -- they are the reference's conventions, found by this test, kept as they are):
#   boxframe     the frame's outer faces sit at dims/2 - 2e, e = thickness / 2 (primitives.go:292-297 hands gsdfBoxFrame3D that half
#                extent): the frame is 2 x thickness smaller than its box on every axis
#   arc          Bounds is the full circle's box, whatever the aperture (primitives2d.go:205-211)
#   shell        t (|d(p / t)| - t): the child is evaluated at p / t -- the shell of the child SCALED by t (cpu_evaluators.go:428-452,
#                operations.go:741-746 alike), far inside the child's bounds, which Shell keeps
#   revolve_off  Bounds subtracts `off` where the evaluator adds it (operations2d.go:168-175, "TODO" there): degenerate box here
#   rotate / circarray / array   boxes of boxes: loose by construction
LOOSE = {"boxframe", "arc", "arc_wide", "shell", "rotate", "circarray", "array", "revolve_off"}
SAMPLE_BOX = {"revolve_off": (np.array([-1.5, -0.4, -1.5]), np.array([1.5, 0.4, 1.5])), "shell": (np.array([-0.06] * 3), np.array([0.06] * 3))}


@pytest.mark.parametrize("name", [n for n, _ in _SHAPES])
def test_field_is_an_exact_distance_to_a_surface_that_fills_its_bounds(name):
    sh = dict(_SHAPES)[name]
    dim = 2 if sh.is2d else 3
    sdf = OracleSDF(sh.tree())
    bb = sh.Bounds().astype(np.float64)
    lo, hi = (bb[:3], bb[3:]) if dim == 3 else (bb[[0, 1]], bb[[3, 4]])
    blo, bhi = lo, hi
    if name in SAMPLE_BOX:
        lo, hi = SAMPLE_BOX[name]
    size = hi - lo
    diag = float(np.linalg.norm(size))
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    n = 6000
    p = (lo - 0.3 * size) + rng.random((n, dim)) * (1.6 * size)
    h = 2e-3 * diag
    ev = lambda q: sdf.Evaluate(np.ascontiguousarray(q, np.float32)).astype(np.float64)
    d = ev(p)
    g = np.zeros((n, dim))
    for a in range(dim):
        e = np.zeros(dim)
        e[a] = h
        g[:, a] = (ev(p + e) - ev(p - e)) / (2 * h)
    gn = np.linalg.norm(g, axis=1)
    # 1. eikonal, away from the field's creases (medial axis, the Voronoi borders of corners): there a central difference straddles
    # two branches; a crease is a set of measure zero, so nearly every sample must pass
    ok = np.abs(gn - 1) < 0.03
    assert ok.mean() > 0.93, (name, float(ok.mean()), float(np.median(gn)))
    assert abs(float(np.median(gn)) - 1) < 2e-3, (name, float(np.median(gn)))
    # 2. closest point: the foot of every sample with a clean gradient lies on the zero set
    q = p[ok] - d[ok, None] * g[ok] / gn[ok, None]
    dq = np.abs(ev(q))
    assert np.quantile(dq, 0.97) < 4e-3 * diag, (name, float(np.quantile(dq, 0.97)), diag)
    # 3. extent: the feet span the Bounds() in every direction (bounds are tight for these nodes; the feet of samples beyond a corner
    # or an extreme point ARE that corner or approach that point)
    feet = q[dq < 2e-3 * diag]
    assert len(feet) > 0.8 * ok.sum()
    span_lo, span_hi = feet.min(axis=0), feet.max(axis=0)
    tol = 0.02 * diag
    if name not in LOOSE:
        assert (np.abs(span_lo - lo) < tol).all() and (np.abs(span_hi - hi) < tol).all(), (name, span_lo, lo, span_hi, hi)
    elif name != "revolve_off":
        assert (span_lo > blo - tol).all() and (span_hi < bhi + tol).all(), (name, span_lo, blo, span_hi, bhi)  # inside its (loose) bounds at least
    # sign: the far corner of the sampling box is outside, and some sample is inside (every solid here has an interior)
    assert ev((hi + 0.3 * size)[None, :])[0] > 0 and d.min() < 0, name
