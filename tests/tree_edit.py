"""Helpers for tests that need trees the reference's constructors would refuse: a tree blob copied into memory of its own,
free to be edited (degenerate parameters, a negative scale factor)."""
import ctypes as C

import numpy as np

from gsdf_amd._ctypes_common import GsdfNode, GsdfTree, OP


def clone(t):
    """A tree blob in memory of its own."""
    nodes = (GsdfNode * t.n_nodes)(*[t.nodes[i] for i in range(t.n_nodes)])
    links = (C.c_uint32 * max(1, t.n_links))(*[t.links[i] for i in range(t.n_links)])
    aux = (C.c_float * max(1, t.n_aux))(*[t.aux[i] for i in range(t.n_aux)])
    o = GsdfTree()
    o.nodes, o.n_nodes = C.cast(nodes, C.POINTER(GsdfNode)), t.n_nodes
    o.links, o.n_links = C.cast(links, C.POINTER(C.c_uint32)), t.n_links
    o.aux, o.n_aux = C.cast(aux, C.POINTER(C.c_float)), t.n_aux
    o.root = t.root
    for k in range(6):
        o.bb[k] = t.bb[k]
    o._keep = (nodes, links, aux)
    return o


def first(t, op):
    return next(i for i in range(t.n_nodes) if t.nodes[i].op == OP[op])


def negative_scale_trees():
    """Scale nodes with a NEGATIVE factor (the reference's Scale accepts any factor, gsdf.go Scale / cpu_evaluators.go:288-312:
    the shape is mirrored through the origin) around maps that stretch -- a twist, a screw, a non-rigid transform. The builder's
    Bounds() of such a node are inside out, so the trees are built with the positive factor, the factor's sign is flipped in the
    blob and the bounds are replaced by their hull with their mirror image. [(name, tree)]"""
    from scaffold.builder import Builder
    b = Builder()
    m = np.eye(4, dtype=np.float32)
    m[0, 0], m[1, 1], m[2, 2], m[0, 1], m[1, 2] = 1.4, 0.7, 1.2, 0.3, -0.4
    inner = [
        ("twist", b.Twist(b.Translate(b.NewBox(1.0, 0.5, 2.0, 0.05), 0.5, 0.1, 0.2), 2.2)),
        ("screw", b.Translate(b.ScrewPlasticButtress(1.6, 0.4, 1.2), 0.3, -0.2, 0.1)),
        ("affine", b.Transform(b.Translate(b.NewBox(0.8, 0.9, 0.7, 0.05), 0.3, 0.2, -0.1), m.reshape(-1))),
    ]
    out = []
    for name, sh in inner:
        for factor in (1.7, 0.6):
            t = clone(b.Union(b.Scale(sh, factor), b.Translate(b.NewSphere(0.3), 0.2, 0.3, -0.2)).tree())
            t.nodes[first(t, "SCALE")].p[0] = -factor
            bb = np.array(t.bb[:], np.float32)
            mx = np.maximum(np.abs(bb[:3]), np.abs(bb[3:]))
            for k in range(3):
                t.bb[k], t.bb[k + 3] = -float(mx[k]), float(mx[k])
            out.append((f"{name}-scale{-factor}", t))
    return out
