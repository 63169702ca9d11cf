"""Host mirror of the reference Builder: argument validation, error behaviour, Bounds(), flattening."""
import math

import numpy as np
import pytest

import corpus
from gsdf_amd._ctypes_common import OP, OPS
from scaffold.builder import Builder, FlagNoDimensionPanic, ShapeError, NutCircular
from oracle.oracle import OracleSDF


def test_shape_errors_panic_by_default_and_accumulate_with_flag():
    b = Builder()
    for bad in (lambda: b.NewSphere(0), lambda: b.NewBox(1, 1, 1, 0.6), lambda: b.NewBox(-1, 1, 1, 0),
                lambda: b.NewCylinder(1, 1, 0.5), lambda: b.NewTorus(1, 0.6), lambda: b.NewCircle(-1),
                lambda: b.NewPolygon([(0, 0), (1, 0)]), lambda: b.NewPolygon([(0, 0), (0, 0), (1, 1), (2, 0)]),
                lambda: b.NewArc(1, 7.0, 0.1), lambda: b.Array(b.NewSphere(1), 1, 1, 1, 0, 1, 1),
                lambda: b.CircularArray(b.NewSphere(1), 5, 4), lambda: b.Twist(b.NewSphere(1), 0),
                lambda: b.Symmetry(b.NewSphere(1), False, False, False)):
        with pytest.raises(ShapeError):
            bad()
    with pytest.raises(ShapeError):
        b.Union(b.NewSphere(1))  # need at least 2 arguments
    with pytest.raises(ShapeError):
        b.Difference(b.NewSphere(1), b.NewCircle(1))  # 2D where 3D is required ("nil shader")
    b2 = Builder(FlagNoDimensionPanic)
    s = b2.NewSphere(-1)  # accumulates instead of raising (gsdf.go:33)
    assert s is not None and len(b2.Err()) == 1 and "sphere radius" in b2.Err()[0]


def test_union_splices_nested_unions():
    b = Builder()
    u1 = b.Union(b.NewSphere(1), b.NewSphere(2))
    u2 = b.Union(u1, b.NewSphere(3))
    t = u2.tree()
    root = t.nodes[t.root]
    assert OPS[root.op] == "UNION" and root.nchild == 3  # operations.go:44-50


def test_polygon_drops_closing_duplicate():
    b = Builder()
    p = b.NewPolygon([(0, 0), (0, 1), (1, 1), (0, 0)])
    t = p.tree()
    assert t.nodes[t.root].aux_len == 6  # 3 vertices


def test_line_degenerates_to_circle():
    b = Builder()
    s = b.NewLine2D(0, 0, 0, 0, 0.5)  # primitives2d.go:24-29
    assert OPS[b.op(s.id)] == "CIRCLE2D"


def test_bounds():
    b = Builder()
    np.testing.assert_array_equal(b.NewSphere(2).Bounds(), [-2, -2, -2, 2, 2, 2])
    np.testing.assert_array_equal(b.NewCylinder(1, 4, 0).Bounds(), [-1, -1, -2, 1, 1, 2])
    np.testing.assert_array_equal(b.NewTorus(2, 0.5).Bounds(), [-2.5, -2.5, -0.5, 2.5, 2.5, 0.5])
    np.testing.assert_array_equal(b.Translate(b.NewSphere(1), 1, 2, 3).Bounds(), [0, 1, 2, 2, 3, 4])
    np.testing.assert_array_equal(b.Scale(b.NewSphere(1), 3).Bounds(), [-3, -3, -3, 3, 3, 3])
    np.testing.assert_array_equal(b.Union(b.NewSphere(1), b.Translate(b.NewSphere(1), 3, 0, 0)).Bounds(), [-1, -1, -1, 4, 1, 1])
    np.testing.assert_array_equal(b.Difference(b.NewSphere(1), b.NewSphere(5)).Bounds(), [-1, -1, -1, 1, 1, 1])
    np.testing.assert_array_equal(b.Extrude(b.NewRectangle(2, 4), 6).Bounds(), [-1, -2, -3, 1, 2, 3])
    np.testing.assert_array_equal(b.Symmetry(b.Translate(b.NewSphere(1), 2, 0, 0), True, False, False).Bounds(), [-3, -1, -1, 3, 1, 1])
    rb = b.Rotate(b.NewBox(2, 4, 6, 0), math.pi / 2, (0, 0, 1)).Bounds()
    np.testing.assert_allclose(rb, [-2, -1, -3, 2, 1, 3], atol=1e-6)
    np.testing.assert_array_equal(b.Offset(b.NewBox(2, 2, 2, 0), -0.5).Bounds(), [-1.5, -1.5, -1.5, 1.5, 1.5, 1.5])


def test_bounds_contain_surface_for_corpus():
    # gsdf_test.go:772-838 test_bounds: the SDF is >= 0 on the faces of a slightly grown bounding box
    for fn in (corpus.shapes3d, corpus.shapes2d):
        _, shapes = fn()
        for name, sh in shapes:
            if name.startswith(("shell", "array", "elongate", "scene_bolt", "rotate", "nut_knurl", "revolve_off")):  # revolution.Bounds is marked TODO upstream
                continue  # the reference's own Bounds() are loose/approximate for these ops
            bb = sh.Bounds().astype(np.float64)
            dim = 2 if sh.is2d else 3
            lo, hi = bb[:3][:dim], bb[3:][:dim]
            grow = 0.02 * (hi - lo).max()
            pts = []
            for a in range(dim):
                for side, v in ((0, lo[a] - grow), (1, hi[a] + grow)):
                    g = np.random.default_rng(a * 2 + side).random((40, dim)) * (hi - lo) + lo
                    g[:, a] = v
                    pts.append(g)
            d = OracleSDF(sh.tree()).Evaluate(np.concatenate(pts).astype(np.float32))
            assert (d >= -1e-4 * (hi - lo).max()).all(), (name, d.min())


def test_transform_inverse_is_inverse():
    b = Builder()
    box = b.NewBox(1, 2, 3, 0)
    r = b.Rotate(box, 0.7, (1, 2, 3))
    t = r.tree()
    node = t.nodes[t.root]
    assert OPS[node.op] == "TRANSFORM" and node.aux_len == 16
    inv = np.array([t.aux[node.aux_off + i] for i in range(16)], np.float64).reshape(4, 4)
    ax = np.array([1, 2, 3]) / np.sqrt(14)
    c, s = math.cos(0.7), math.sin(0.7)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) * c + s * K + (1 - c) * np.outer(ax, ax)
    np.testing.assert_allclose(inv[:3, :3] @ R, np.eye(3), atol=1e-6)
    np.testing.assert_allclose(inv[3], [0, 0, 0, 1], atol=1e-7)


def test_iso_thread_polygon_shape():
    b = Builder()
    t = b.ISOThread(1.0, 0.1, True).tree()       # external: 8 vertices, two of them smoothed with 5 facets -> 8 - 2 + 12 = 18
    assert t.nodes[t.root].aux_len // 2 == 18
    t = b.ISOThread(1.0, 0.1, False).tree()      # internal: 7 vertices, one smoothed -> 7 - 1 + 6 = 12
    assert t.nodes[t.root].aux_len // 2 == 12
    v = np.array([t.aux[t.nodes[t.root].aux_off + i] for i in range(24)], np.float32).reshape(-1, 2)
    assert v[:, 0].min() == np.float32(-0.1) and v[:, 0].max() == np.float32(0.1) and v[:, 1].min() == 0


def test_scene_trees():
    b = Builder()
    s = b.Scene("npt-flange")
    t = s.tree()
    assert OPS[t.nodes[t.root].op] == "SCALE"
    ops = {OPS[t.nodes[i].op] for i in range(t.n_nodes)}
    assert {"SCALE", "DIFF", "SMOOTH_UNION", "CYLINDER", "TRANSLATE", "SCREW", "POLY2D"} <= ops
    k = b.Scene("knurled-cylinder")
    tk = k.tree()
    opsk = [OPS[tk.nodes[i].op] for i in range(tk.n_nodes)]
    assert opsk.count("TWIST") == 2 and opsk.count("CIRCARRAY") == 1  # the knurl subtree is shared, not copied
    bl = b.Scene("bolt")
    assert OPS[bl.tree().nodes[bl.tree().root].op] == "TRANSFORM"


def test_metric_f2f_and_nut_dimensions():
    b = Builder()
    nut = b.NutNPT(0.5, NutCircular)
    bb = nut.Bounds()
    f2f = np.float32(22.4 / 25.4)
    nr = np.float32(f2f / np.float32(math.sqrt(3)))
    assert abs(bb[3] - nr * np.float32(1.1)) < 1e-6
