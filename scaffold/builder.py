"""Python face of the C++ mirror of the reference's gsdf.Builder (scaffold/: test and benchmark scaffolding, not product).

Same method names and argument meaning as /root/reference/gsdf.go + primitives*.go + operations*.go
(`bld.NewSphere(r)`, `bld.Union(a, b, ...)`, `bld.Translate(s, x, y, z)` ...), so parity tests read
like the reference's own tests. All arithmetic (validation, Bounds, polygon smoothing, matrix
inverses) happens in the C++ library in float32; this file only marshals arguments.
"""
import ctypes as C
import os
import numpy as np

from gsdf_amd._ctypes_common import GsdfTree, OPS, FIRST_2D

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FlagNoDimensionPanic = 1 << 0
FlagUseShaderBuffers = 1 << 1
FlagNoShaderBuffers = 1 << 2

NutCircular, NutHex, NutKnurl = 1, 2, 3


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libgsdfhost.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(path)
        lib.gsdfb_new.restype = C.c_void_p
        lib.gsdfb_new.argtypes = [C.c_uint64]
        lib.gsdfb_free.argtypes = [C.c_void_p]
        lib.gsdfb_last_error.restype = C.c_char_p
        lib.gsdfb_op.restype = C.c_int
        lib.gsdfb_op.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int), C.c_int]
        lib.gsdfb_bounds.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
        lib.gsdfb_tree.argtypes = [C.c_void_p, C.c_int, C.POINTER(GsdfTree)]
        lib.gsdfb_num_errs.argtypes = [C.c_void_p]
        lib.gsdfb_err.restype = C.c_char_p
        lib.gsdfb_err.argtypes = [C.c_void_p, C.c_int]
        lib.gsdfb_num_nodes.argtypes = [C.c_void_p]
        lib.gsdfb_node_op.argtypes = [C.c_void_p, C.c_int]
        _LIB = lib
    return _LIB


class ShapeError(ValueError):
    """Raised where the reference Builder panics (shapeErrorf / nilsdf)."""


class Shader:
    """A node of a Builder's tree (glbuild.Shader3D / Shader2D)."""
    __slots__ = ("bld", "id")

    def __init__(self, bld, id_):
        self.bld, self.id = bld, id_

    @property
    def is2d(self):
        return self.bld.op(self.id) >= FIRST_2D

    def Bounds(self):
        """Min/max corners as a float32 array of 6 (2D: z entries are 0)."""
        return self.bld.bounds(self)

    def Diagonal(self):
        """ms3.Box.Diagonal() of Bounds(), float32 (Norm of Size, nested Hypot)."""
        b = self.Bounds().astype(np.float32)
        sz = (b[3:] - b[:3]).astype(np.float32)

        def hyp(p, q):
            p, q = np.float32(abs(p)), np.float32(abs(q))
            if p < q:
                p, q = q, p
            if p == 0:
                return np.float32(0)
            q = np.float32(q / p)
            return np.float32(p * np.sqrt(np.float32(np.float32(1) + np.float32(q * q))))
        return hyp(sz[0], hyp(sz[1], sz[2]))

    def tree(self):
        return self.bld.tree(self)

    def __repr__(self):
        return f"<Shader {OPS[self.bld.op(self.id)]}#{self.id}>"


class Builder:
    def __init__(self, flags=0):
        self._lib = _lib()
        self._h = self._lib.gsdfb_new(flags)

    def __del__(self):
        try:
            if self._h:
                self._lib.gsdfb_free(self._h)
                self._h = None
        except Exception:
            pass

    # ---- plumbing
    def _call(self, name, floats=(), ints=()):
        fa = (C.c_float * max(1, len(floats)))(*[float(x) for x in floats])
        ia = (C.c_int * max(1, len(ints)))(*[int(x) for x in ints])
        r = self._lib.gsdfb_op(self._h, name.encode(), fa, len(floats), ia, len(ints))
        if r < 0:
            raise ShapeError(self._lib.gsdfb_last_error().decode())
        return Shader(self, r)

    @staticmethod
    def _ids(*shaders):
        out = []
        for s in shaders:
            if s is None:
                out.append(-1)
            else:
                out.append(s.id)
        return out

    def op(self, id_):
        return self._lib.gsdfb_node_op(self._h, id_)

    def bounds(self, s):
        bb = (C.c_float * 6)()
        if self._lib.gsdfb_bounds(self._h, s.id, bb) != 0:
            raise ShapeError(self._lib.gsdfb_last_error().decode())
        return np.array(bb[:], dtype=np.float32)

    def tree(self, s):
        t = GsdfTree()
        if self._lib.gsdfb_tree(self._h, s.id, C.byref(t)) != 0:
            raise ShapeError(self._lib.gsdfb_last_error().decode())
        t._keepalive = self
        return t

    def Err(self):
        n = self._lib.gsdfb_num_errs(self._h)
        return [self._lib.gsdfb_err(self._h, i).decode() for i in range(n)]

    # ---- 3D primitives (primitives.go)
    def NewSphere(self, r): return self._call("NewSphere", [r])
    def NewBox(self, x, y, z, round_): return self._call("NewBox", [x, y, z, round_])
    def NewBoxFrame(self, x, y, z, e): return self._call("NewBoxFrame", [x, y, z, e])
    def NewTorus(self, R, r): return self._call("NewTorus", [R, r])
    def NewCylinder(self, r, h, rounding): return self._call("NewCylinder", [r, h, rounding])
    def NewHexagonalPrism(self, f2f, h): return self._call("NewHexagonalPrism", [f2f, h])
    def NewTriangularPrism(self, h, length): return self._call("NewTriangularPrism", [h, length])
    def NewBoundsBoxFrame(self, bb): return self._call("NewBoundsBoxFrame", list(bb))
    # ---- 3D ops (operations.go)
    def Union(self, *s): return self._call("Union", [], self._ids(*s))
    def Difference(self, a, b): return self._call("Difference", [], self._ids(a, b))
    def Intersection(self, a, b): return self._call("Intersection", [], self._ids(a, b))
    def Xor(self, a, b): return self._call("Xor", [], self._ids(a, b))
    def SmoothUnion(self, k, a, b): return self._call("SmoothUnion", [k], self._ids(a, b))
    def SmoothDifference(self, k, a, b): return self._call("SmoothDifference", [k], self._ids(a, b))
    def SmoothIntersect(self, k, a, b): return self._call("SmoothIntersect", [k], self._ids(a, b))
    def Scale(self, s, f): return self._call("Scale", [f], self._ids(s))
    def Symmetry(self, s, x, y, z): return self._call("Symmetry", [], self._ids(s) + [x, y, z])
    def Transform(self, s, m16): return self._call("Transform", list(m16), self._ids(s))
    def Rotate(self, s, radians, axis): return self._call("Rotate", [radians] + list(axis), self._ids(s))
    def Translate(self, s, x, y, z): return self._call("Translate", [x, y, z], self._ids(s))
    def Offset(self, s, off): return self._call("Offset", [off], self._ids(s))
    def Array(self, s, sx, sy, sz, nx, ny, nz): return self._call("Array", [sx, sy, sz], self._ids(s) + [nx, ny, nz])
    def Elongate(self, s, x, y, z): return self._call("Elongate", [x, y, z], self._ids(s))
    def Shell(self, s, t): return self._call("Shell", [t], self._ids(s))
    def CircularArray(self, s, n, div): return self._call("CircularArray", [], self._ids(s) + [n, div])
    def Twist(self, s, k): return self._call("Twist", [k], self._ids(s))
    def Extrude(self, s, h): return self._call("Extrude", [h], self._ids(s))
    def Revolve(self, s, off): return self._call("Revolve", [off], self._ids(s))
    # ---- 2D primitives (primitives2d.go)
    def NewLine2D(self, x0, y0, x1, y1, w): return self._call("NewLine2D", [x0, y0, x1, y1, w])
    def NewLines2D(self, segs, w): return self._call("NewLines2D", [w] + [c for s in segs for p in s for c in p])
    def NewArc(self, r, angle, thick): return self._call("NewArc", [r, angle, thick])
    def NewCircle(self, r): return self._call("NewCircle", [r])
    def NewEquilateralTriangle(self, h): return self._call("NewEquilateralTriangle", [h])
    def NewRectangle(self, x, y): return self._call("NewRectangle", [x, y])
    def NewHexagon(self, side): return self._call("NewHexagon", [side])
    def NewOctagon(self, c): return self._call("NewOctagon", [c])
    def NewEllipse(self, a, b): return self._call("NewEllipse", [a, b])
    def NewPolygon(self, verts): return self._call("NewPolygon", [c for v in verts for c in v])
    def NewDiamond2D(self, x, y): return self._call("NewDiamond2D", [x, y])
    def NewRoundedX(self, w, t): return self._call("NewRoundedX", [w, t])
    def NewQuadraticBezier2D(self, a, b, c, t): return self._call("NewQuadraticBezier2D", list(a) + list(b) + list(c) + [t])
    # ---- 2D ops (operations2d.go)
    def Union2D(self, *s): return self._call("Union2D", [], self._ids(*s))
    def Difference2D(self, a, b): return self._call("Difference2D", [], self._ids(a, b))
    def Intersection2D(self, a, b): return self._call("Intersection2D", [], self._ids(a, b))
    def Xor2D(self, a, b): return self._call("Xor2D", [], self._ids(a, b))
    def Array2D(self, s, sx, sy, nx, ny): return self._call("Array2D", [sx, sy], self._ids(s) + [nx, ny])
    def Offset2D(self, s, f): return self._call("Offset2D", [f], self._ids(s))
    def Translate2D(self, s, x, y): return self._call("Translate2D", [x, y], self._ids(s))
    def Rotate2D(self, s, theta): return self._call("Rotate2D", [theta], self._ids(s))
    def Symmetry2D(self, s, x, y): return self._call("Symmetry2D", [], self._ids(s) + [x, y])
    def Annulus(self, s, sub): return self._call("Annulus", [sub], self._ids(s))
    def CircularArray2D(self, s, n, div): return self._call("CircularArray2D", [], self._ids(s) + [n, div])
    def Scale2D(self, s, f): return self._call("Scale2D", [f], self._ids(s))
    def TranslateMulti2D(self, s, disp): return self._call("TranslateMulti2D", [c for d in disp for c in d], self._ids(s))
    def Elongate2D(self, s, x, y): return self._call("Elongate2D", [x, y], self._ids(s))
    # ---- forge/threads
    def ISOThread(self, D, P, ext): return self._call("threads.ISO.Thread", [D, P], [int(ext)])
    def ScrewISO(self, D, P, ext, length): return self._call("threads.Screw.ISO", [D, P, length], [int(ext)])
    def ScrewNPT(self, nominal, length): return self._call("threads.Screw.NPT", [nominal, length])
    def NutNPT(self, nominal, style): return self._call("threads.Nut.NPT", [nominal], [style])
    def NutISO(self, D, P, ext, style): return self._call("threads.Nut.ISO", [D, P], [int(ext), style])
    def BoltISO(self, D, P, ext, style, total, shank): return self._call("threads.Bolt.ISO", [D, P, total, shank], [int(ext), style])
    def HexHead(self, r, h, round_neg, round_pos): return self._call("threads.HexHead", [r, h], [int(round_neg), int(round_pos)])
    def KnurledHead(self, r, h, pitch): return self._call("threads.KnurledHead", [r, h, pitch])
    def ScrewPlasticButtress(self, D, P, length): return self._call("threads.Screw.PlasticButtress", [D, P, length])
    def Knurl(self, length, radius, pitch, height, theta): return self._call("threads.Knurl", [length, radius, pitch, height, theta])
    # ---- forge/textsdf (font.go): one line of text set in a TrueType font, as a 2-D shape
    def TextLine(self, ttf_bytes, text, reltol=0.0):
        """textsdf.Font{}.LoadTTFBytes(ttf); Configure(FontConfig{RelativeGlyphTolerance: reltol}); TextLine(text)."""
        buf = (C.c_uint8 * len(ttf_bytes)).from_buffer_copy(ttf_bytes)
        self._lib.gsdfb_textsdf_line.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_float, C.c_void_p, C.c_void_p]
        r = self._lib.gsdfb_textsdf_line(self._h, buf, len(ttf_bytes), text.encode("utf-8"), reltol, None, None)
        if r < 0:
            raise ShapeError(self._lib.gsdfb_last_error().decode())
        return Shader(self, r)

    def TextMetrics(self, ttf_bytes, two_chars):
        """(AdvanceWidth(c0), Kern(c0, c1)) of textsdf.Font for the first two characters."""
        buf = (C.c_uint8 * len(ttf_bytes)).from_buffer_copy(ttf_bytes)
        adv, kern = C.c_float(), C.c_float()
        self._lib.gsdfb_textsdf_line.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_float, C.c_void_p, C.c_void_p]
        r = self._lib.gsdfb_textsdf_line(self._h, buf, len(ttf_bytes), two_chars.encode("utf-8"), 0.0, C.byref(adv), C.byref(kern))
        if r < 0:
            raise ShapeError(self._lib.gsdfb_last_error().decode())
        return adv.value, kern.value

    # ---- benchmark scenes (examples/*)
    def Scene(self, name, *fargs, ints=()): return self._call("scene." + name, list(fargs), list(ints))
