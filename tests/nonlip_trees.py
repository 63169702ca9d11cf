"""Seeded random trees whose fields are NOT 1-Lipschitz: twists, high-lead and tapered screws, buttress (asymmetric) thread
forms, knurls, non-rigid transforms, nested in scales / shells and combined with every boolean. What the octree's centre
tests have to survive: |grad| > 1, jumps across a screw's sawtooth seams. No (circular) arrays: their
sector seams are the one assumption the bounds make (the reference's own Bounds() of those nodes make it too)."""
import numpy as np

from scaffold.builder import Builder, ShapeError


def _base(b, r):
    u = lambda lo, hi: float(r.uniform(lo, hi))
    k = int(r.integers(0, 11))
    if k == 0: return b.Twist(b.NewBox(u(0.6, 1.4), u(0.3, 0.8), u(1.0, 2.5), 0.0), u(-2.5, 2.5))
    if k == 1: return b.Twist(b.Translate(b.NewCylinder(u(0.15, 0.3), u(1.5, 2.5), 0.0), u(0.4, 0.9), 0.0, 0.0), u(-4.0, 4.0))
    if k == 2: return b.Knurl(u(0.8, 1.6), u(0.6, 1.0), u(0.15, 0.3), u(0.04, 0.08), u(0.3, 1.2))
    if k == 3: return b.ScrewPlasticButtress(u(1.2, 2.4), u(0.25, 0.6), u(0.8, 1.6))
    if k == 4: return b.ScrewNPT(float(r.choice([0.125, 0.25, 0.5])), u(0.4, 0.8))
    if k == 5: return b.ScrewISO(u(0.8, 1.6), u(0.2, 0.5), bool(r.integers(0, 2)), u(0.8, 1.6))
    if k == 6:  # non-rigid affine map: stretch + shear (the builder inverts it)
        m = np.eye(4, dtype=np.float32)
        m[0, 0], m[1, 1], m[2, 2] = u(0.5, 1.8), u(0.5, 1.8), u(0.5, 1.8)
        m[0, 1], m[1, 2] = u(-0.5, 0.5), u(-0.5, 0.5)
        m[:3, 3] = [u(-0.3, 0.3), u(-0.3, 0.3), u(-0.3, 0.3)]
        return b.Transform(b.NewBox(u(0.5, 1.2), u(0.5, 1.2), u(0.5, 1.2), u(0.0, 0.1)), m.reshape(-1))
    if k == 7: return b.KnurledHead(u(0.8, 1.2), u(0.6, 1.0), u(0.15, 0.3))
    if k == 8: return b.Twist(b.Extrude(b.NewHexagon(u(0.4, 0.8)), u(1.5, 2.5)), u(-3.0, 3.0))
    if k == 9: return b.NewSphere(u(0.4, 1.0))
    return b.NewCylinder(u(0.4, 0.9), u(0.8, 2.0), float(r.choice([0.0, 0.05])))


def _shape(b, r, depth):
    if depth <= 0 or r.random() < 0.2:
        return _base(b, r)
    u = lambda lo, hi: float(r.uniform(lo, hi))
    a = _shape(b, r, depth - 1)
    k = int(r.integers(0, 14))
    if k == 0: return b.Union(a, b.Translate(_shape(b, r, depth - 1), u(-0.5, 0.5), u(-0.5, 0.5), u(-0.5, 0.5)))
    if k == 1: return b.Difference(a, b.Translate(_shape(b, r, depth - 1), u(-0.4, 0.4), u(-0.4, 0.4), u(-0.4, 0.4)))
    if k == 2: return b.Intersection(a, _shape(b, r, depth - 1))
    if k == 3: return b.SmoothUnion(u(0.05, 0.3), a, _shape(b, r, depth - 1))
    if k == 4: return b.SmoothDifference(u(0.05, 0.3), a, b.Translate(_shape(b, r, depth - 1), u(-0.4, 0.4), 0.0, u(-0.4, 0.4)))
    if k == 5: return b.SmoothIntersect(u(0.05, 0.3), a, _shape(b, r, depth - 1))
    if k == 6: return b.Xor(a, b.Translate(_shape(b, r, depth - 1), u(0.2, 0.5), 0.0, 0.0))
    if k == 7: return b.Scale(a, u(0.5, 2.0))
    if k == 8: return b.Shell(a, u(0.05, 0.15))
    if k == 9: return b.Rotate(a, u(-3, 3), (u(-1, 1), u(-1, 1), u(0.1, 1)))
    if k == 10: return b.Twist(a, u(-1.0, 1.0))
    if k == 11: return b.Offset(a, u(-0.03, 0.06))
    if k == 12: return b.Elongate(a, u(0.0, 0.4), u(0.0, 0.4), u(0.0, 0.4))
    return b.Translate(a, u(-0.5, 0.5), u(-0.5, 0.5), u(-0.5, 0.5))


def nonlip_shapes(seed, count, depth=2):
    r = np.random.default_rng(seed)
    b = Builder()
    out = []
    tries = 0
    while len(out) < count and tries < 30 * count:
        tries += 1
        try:
            sh = _shape(b, r, depth)
            bb = np.asarray(sh.Bounds(), np.float32)
            ext = bb[3:] - bb[:3]
            if not np.isfinite(bb).all() or ext.max() > 40 or ext.min() <= 0:
                continue
            out.append(sh)
        except ShapeError:
            continue
    return b, out
