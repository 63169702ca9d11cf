"""Shape corpus shared by the CPU and GPU parity tests.

Mirrors the reference's own test corpus, /root/reference/gsdf_test.go: testPrimitives3D (:182-201),
testBinOp3D (:203-231), testRandomUnary3D (:255-283), testPrimitives2D (:285-353), testBinary2D
(:355-373), testRandomUnary2D (:233-253) with the same shapes and parameter ranges (numpy RNG instead
of Go's math/rand, fixed seed), plus the benchmark scenes and forge/threads parts.
"""
import math

import numpy as np

from scaffold.builder import Builder, NutCircular, NutHex, NutKnurl


def _rng():
    return np.random.default_rng(1)


def _f(rng):
    return float(np.float32(rng.random()))


def shapes3d(bld=None):
    b = bld or Builder()
    rng = _rng()
    out = []
    maxdim = 1.0
    dv = (maxdim, maxdim * 0.47, maxdim * 0.8)
    thick = maxdim / 10
    # testPrimitives3D
    out += [("sphere", b.NewSphere(1)), ("box", b.NewBox(dv[0], dv[1], dv[2], thick)),
            ("boxframe", b.NewBoxFrame(dv[0], dv[1], dv[2], thick)), ("cyl0", b.NewCylinder(dv[0], dv[1], 0)),
            ("cylr", b.NewCylinder(dv[0], dv[1], thick)), ("hexprism", b.NewHexagonalPrism(dv[0], dv[1])),
            ("torus", b.NewTorus(dv[0], dv[1])), ("triprism", b.NewTriangularPrism(1, 0.5))]
    # testBinOp3D
    s1 = b.NewSphere(1)
    s2 = b.Translate(b.NewBox(1, 0.6, .8, 0.1), 0.5, 0.7, 0.8)
    out += [("union", b.Union(s1, s2)), ("diff", b.Difference(s1, s2)), ("intersect", b.Intersection(s1, s2)),
            ("xor", b.Xor(s1, s2)), ("smoothunion", b.SmoothUnion(0.1, s1, s2)),
            ("smoothdiff", b.SmoothDifference(0.1, s1, s2)), ("smoothintersect", b.SmoothIntersect(0.1, s1, s2))]
    out.append(("union3", b.Union(s1, s2, b.Translate(b.NewTorus(1, 0.3), -0.5, 0.2, 0.1))))
    # testRandomUnary3D
    a = b.NewBox(1, 0.61, 0.8, 0.3)
    axis = (0.0, 0.0, 0.0)
    while math.sqrt(sum(x * x for x in axis)) < .5:
        axis = (_f(rng) * 3, _f(rng) * 3, _f(rng) * 3)
    angle = 0.0
    while abs(angle) < 1e-1 or abs(angle) > 1:
        angle = 2 * 3.14159 * (_f(rng) - 0.5)
    out.append(("rotate", b.Rotate(a, angle, axis)))
    bb = a.Bounds()
    size = bb[3:] - bb[:3]
    th = min(float(size.max()) / 128, _f(rng))
    shell = b.Shell(a, th)
    half = b.Translate(b.Translate(b.NewBox(size[0] * 20, size[1] / 3, size[2] * 20, 0), 0, size[1] / 3, 0), 0, size[1] / 3, 0)
    out.append(("shell", b.Difference(shell, half)))
    out.append(("elongate", b.Elongate(a, 0.3 * _f(rng), 0.3 * _f(rng), 0.3 * _f(rng))))
    mn = float(size.min())
    out.append(("round", b.Offset(a, -(mn / 64 + _f(rng) * (mn / 2 - mn / 64)))))
    out.append(("scale", b.Scale(a, 0.01 + _f(rng) * (3 - 0.01))))
    out.append(("symmetry", b.Symmetry(a, True, False, True)))
    out.append(("symmetry_xyz", b.Symmetry(b.Translate(a, 0.3, 0.2, 0.1), True, True, True)))
    out.append(("translate", b.Translate(a, 1.3 * _f(rng), -0.7 * _f(rng), 0.4)))
    out.append(("array", b.Array(a, _f(rng) + 0.1, _f(rng) + 0.1, _f(rng) + 0.1, 3, 2, 5)))
    for i in range(3):
        div = int(rng.integers(0, 16)) + 3
        n = int(rng.integers(0, div)) + 1
        out.append((f"circarray{i}", b.CircularArray(b.Translate(a, 1.5, 0, 0), n, div)))
    out.append(("twist", b.Twist(a, _f(rng))))
    s2d = b.NewRectangle(1, 0.57)
    out.append(("extrude", b.Extrude(s2d, 0.01 + _f(rng) * 3.99)))
    out.append(("revolve", b.Revolve(s2d, 0)))
    out.append(("revolve_off", b.Revolve(b.Translate2D(b.NewCircle(0.3), 1.0, 0.2), 0.25)))
    # forge/threads parts and benchmark scenes
    out.append(("screw_iso_ext", b.ScrewISO(1, 0.1, True, 2.0)))
    out.append(("screw_npt", b.ScrewNPT(0.5, 1.0)))
    out.append(("nut_hex", b.NutISO(3, 0.5, False, NutHex)))
    out.append(("nut_knurl", b.NutISO(4, 0.7, False, NutKnurl)))
    out.append(("hexhead", b.HexHead(2.0, 1.2, True, True)))
    out.append(("scene_npt_flange", b.Scene("npt-flange")))
    out.append(("scene_bolt", b.Scene("bolt")))
    out.append(("scene_knurled_cylinder", b.Scene("knurled-cylinder")))
    out.append(("scene_glyph_plate", b.Scene("glyph-plate")))
    return b, out


def shapes2d(bld=None):
    b = bld or Builder()
    rng = _rng()
    out = []
    maxdim = 1.0
    dv = (maxdim, maxdim * 0.47)
    thick = maxdim / 10
    octv = [(math.cos(2 * math.pi * i / 8), math.sin(2 * math.pi * i / 8)) for i in range(8)]
    segs = [(octv[i - 1], octv[i]) for i in range(8)]
    poly = b.NewPolygon(octv)
    out += [("circle", b.NewCircle(maxdim)), ("line", b.NewLine2D(0, 0, dv[0], dv[1], thick)),
            ("rect", b.NewRectangle(dv[0], dv[1])), ("arc", b.NewArc(dv[0], math.pi / 3, thick)),
            ("hexagon", b.NewHexagon(maxdim)), ("eqtri", b.NewEquilateralTriangle(maxdim)),
            ("ellipse", b.NewEllipse(1, 2)), ("poly", poly),
            ("poly_selfclosed", b.NewPolygon([(0, 0), (0, 1), (1, 1), (0, 0)])),
            ("lines", b.NewLines2D(segs, 0.1)), ("translatemulti", b.TranslateMulti2D(poly, octv)),
            ("octagon", b.NewOctagon(dv[0])), ("diamond", b.NewDiamond2D(dv[0], dv[1])),
            ("roundedx", b.NewRoundedX(dv[0], thick)),
            ("iso_thread_ext", b.ISOThread(1, 0.1, True)), ("iso_thread_int", b.ISOThread(1, 0.1, False))]
    out.append(("union_lines", b.Union2D(b.NewLine2D(1, 2, 3, 4, 0.5), b.NewLine2D(2, 3, 0, 0, 0.2), b.NewLine2D(2, 3, 4, 5, 0.2),
                                          b.NewLines2D([((0, 0), (1, 1)), ((2, 2), (3, 1))], 0.5))))
    # testBinary2D
    s2 = b.NewRectangle(1, 0.61)
    s1 = b.Translate2D(b.NewCircle(0.4), 0.45, 1)
    out += [("union2d", b.Union2D(s1, s2)), ("diff2d", b.Difference2D(s1, s2)), ("intersect2d", b.Intersection2D(s1, s2)),
            ("xor2d", b.Xor2D(s1, s2))]
    # testRandomUnary2D
    obj = b.Translate2D(b.NewRectangle(1, 0.61), 2, .3)
    for i in range(3):
        out.append((f"array2d{i}", b.Array2D(obj, _f(rng) + 0.1, _f(rng) + 0.1, int(rng.integers(0, 8)) + 1, int(rng.integers(0, 8)) + 1)))
        div = int(rng.integers(0, 16)) + 3
        out.append((f"circarray2d{i}", b.CircularArray2D(obj, int(rng.integers(0, div)) + 1, div)))
        out.append((f"rotate2d{i}", b.Rotate2D(obj, math.pi * _f(rng) + 0.001)))
        out.append((f"annulus{i}", b.Annulus(obj, _f(rng) + 1e-3)))
        out.append((f"offset2d{i}", b.Offset2D(obj, _f(rng) - 0.5)))
        out.append((f"scale2d{i}", b.Scale2D(obj, _f(rng) + 1e-2)))
        out.append((f"elongate2d{i}", b.Elongate2D(obj, _f(rng), _f(rng))))
    out.append(("symmetry2d_x", b.Symmetry2D(obj, True, False)))
    out.append(("symmetry2d_xy", b.Symmetry2D(obj, True, True)))
    return b, out


def bezier2d(bld=None):
    b = bld or Builder()
    return b, [("quadbezier", b.NewQuadraticBezier2D((1.0, 0.47), (2.0, 0.47), (1.0, 1.47), 0.1))]


def sample_points(shader, n_grid=9, n_rand=3000, seed=7):
    """Lattice over Bounds() (ms3.AppendGrid as in gsdf_test.go:435, includes exact zeros/edges) plus
    seeded uniform points in the bounds grown by 25 percent."""
    bb = shader.Bounds().astype(np.float64)
    dim = 2 if shader.is2d else 3
    lo, hi = bb[:3][:dim], bb[3:][:dim]
    axes = [np.linspace(lo[a], hi[a], n_grid) for a in range(dim)]
    grid = np.stack(np.meshgrid(*axes, indexing="ij"), -1).reshape(-1, dim)
    rng = np.random.default_rng(seed)
    c, h = (lo + hi) / 2, (hi - lo) / 2 * 1.25
    rnd = c + (rng.random((n_rand, dim)) * 2 - 1) * h
    sym = np.concatenate([np.zeros((1, dim)), np.eye(dim) * h, -np.eye(dim) * h])  # axis points: exact zeros
    return np.ascontiguousarray(np.concatenate([grid, rnd, sym]).astype(np.float32))
