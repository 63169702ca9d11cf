"""ctypes face of the ORACLE (oracle/liborc.so). TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (gsdf_amd/) never imports, links or executes anything under oracle/.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class _Mesh(C.Structure):
    _fields_ = [("tris", C.POINTER(C.c_float)), ("n_tris", C.c_uint64), ("cap", C.c_uint64), ("evals", C.c_uint64),
                ("pruned", C.c_uint64), ("levels", C.c_int), ("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int),
                ("t_eval_s", C.c_double), ("t_march_s", C.c_double),
                ("evals_rows", C.c_uint64), ("evals_points", C.c_uint64), ("evals_points_tails", C.c_uint64), ("evals_points_256", C.c_uint64)]


PRUNE_ASSUME_SDF = 1 << 30  # orc_eval.h: ORC_PRUNE_ASSUME_SDF


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liborc.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_sdf_create.restype = C.c_void_p
        L.orc_sdf_create.argtypes = [C.c_void_p]
        L.orc_sdf_destroy.argtypes = [C.c_void_p]
        L.orc_pool_create.restype = C.c_void_p
        L.orc_pool_create.argtypes = [C.c_size_t]
        L.orc_pool_destroy.argtypes = [C.c_void_p]
        for f in (L.orc_eval3, L.orc_eval2):
            f.restype = C.c_int
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_eval3_bounds.restype = C.c_int
        L.orc_eval3_bounds.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float]
        L.orc_render_flat.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_int, C.POINTER(_Mesh)]
        L.orc_render_octree.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_int, C.POINTER(_Mesh)]
        L.orc_render_dualcontour.argtypes = [C.c_void_p, C.c_float, C.c_int, C.POINTER(_Mesh)]
        L.orc_render_minecraft.argtypes = [C.c_void_p, C.c_float, C.POINTER(_Mesh)]
        L.orc_lsq_mgs64.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_mesh_free.argtypes = [C.POINTER(_Mesh)]
        L.orc_stl_size.restype = C.c_size_t
        L.orc_stl_size.argtypes = [C.c_uint64]
        L.orc_write_stl.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_march_cubes.restype = C.c_uint64
        L.orc_march_cubes.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_float, C.c_void_p]
        L.orc_normals_central_diff.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float]
        L.orc_math_apply.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.orc_mc_edge_table.restype = C.POINTER(C.c_uint16)
        L.orc_mc_tri_table.restype = C.POINTER(C.c_int8)
        _LIB = L
    return _LIB


class MeshResult:
    def __init__(self, m):
        n = int(m.n_tris)
        self.tris = np.ctypeslib.as_array(m.tris, shape=(n, 3, 3)).copy() if n else np.zeros((0, 3, 3), np.float32)
        self.n_tris = n
        self.evals = int(m.evals)
        self.pruned = int(m.pruned)
        self.levels = int(m.levels)
        self.grid = (int(m.nx), int(m.ny), int(m.nz))
        self.t_eval_s = float(m.t_eval_s)
        self.t_march_s = float(m.t_march_s)
        # octree: corner evaluations of the leaf phase under the device's share_corners options (orc_eval.h)
        self.evals_rows, self.evals_points = int(m.evals_rows), int(m.evals_points)
        self.evals_points_tails, self.evals_points_256 = int(m.evals_points_tails), int(m.evals_points_256)


class OracleSDF:
    """gleval.SDF3CPU / SDF2CPU restatement over a gsdf_tree (from scaffold.builder.Shader.tree())."""

    def __init__(self, tree, min_alloc=4096):
        self._L = lib()
        self._tree = tree
        self._h = self._L.orc_sdf_create(C.byref(tree))
        if not self._h:
            raise ValueError("malformed tree")
        self._pool = self._L.orc_pool_create(min_alloc)
        self.bb = np.array(tree.bb[:], np.float32)
        self.evals = 0

    def __del__(self):
        try:
            if self._h:
                self._L.orc_sdf_destroy(self._h)
                self._L.orc_pool_destroy(self._pool)
                self._h = None
        except Exception:
            pass

    def Bounds(self):
        return self.bb

    def Evaluate(self, pos, dist=None):
        """pos: (n,3) or (n,2) float32. Returns dist (n,) float32 (written into `dist` if given)."""
        pos = np.ascontiguousarray(pos, np.float32)
        n = pos.shape[0]
        if dist is None:
            dist = np.empty(n, np.float32)
        if dist.shape[0] != n:
            raise ValueError("position and distance buffer length mismatch")
        if n == 0:
            raise ValueError("empty buffers")
        f = self._L.orc_eval3 if pos.shape[1] == 3 else self._L.orc_eval2
        err = f(self._h, self._pool, pos.ctypes.data, dist.ctypes.data, n)
        if err:
            raise RuntimeError(f"oracle eval error {err}")
        self.evals += n
        return dist

    def EvaluateBounds(self, pos, h):
        """(lo, hi): bounds of the field over the ball of radius h around each point (orc_eval3_bounds)."""
        pos = np.ascontiguousarray(pos, np.float32)
        n = pos.shape[0]
        lo, hi = np.empty(n, np.float32), np.empty(n, np.float32)
        err = self._L.orc_eval3_bounds(self._h, self._pool, pos.ctypes.data, lo.ctypes.data, hi.ctypes.data, n, np.float32(h))
        if err:
            raise RuntimeError(f"oracle eval error {err}")
        return lo, hi

    def render_flat(self, res, batch=4096, nthreads=1):
        m = _Mesh()
        err = self._L.orc_render_flat(self._h, np.float32(res), batch, nthreads, C.byref(m))
        if err:
            raise RuntimeError(f"oracle flat renderer error {err}")
        r = MeshResult(m)
        self._L.orc_mesh_free(C.byref(m))
        return r

    def render_octree(self, res, batch=4096, prune=True, assume_sdf=False):
        """prune: True / False / bit mask of the levels to centre-test; assume_sdf: the reference's predicate verbatim
        (|d| >= size*sqrt3/2) instead of the field's bounds over the cube."""
        prune = int(prune) | (PRUNE_ASSUME_SDF if assume_sdf else 0)
        m = _Mesh()
        err = self._L.orc_render_octree(self._h, np.float32(res), batch, int(prune), C.byref(m))
        if err:
            raise RuntimeError(f"oracle octree renderer error {err}")
        r = MeshResult(m)
        self._L.orc_mesh_free(C.byref(m))
        return r

    def render_dualcontour(self, res, chiseled=False):
        m = _Mesh()
        err = self._L.orc_render_dualcontour(self._h, np.float32(res), int(chiseled), C.byref(m))
        if err:
            raise RuntimeError(f"oracle dual contour renderer error {err}")
        r = MeshResult(m)
        self._L.orc_mesh_free(C.byref(m))
        return r

    def render_minecraft(self, res):
        """glrender.minecraftRender (dual_contour.go:297-403)."""
        m = _Mesh()
        err = self._L.orc_render_minecraft(self._h, np.float32(res), C.byref(m))
        if err:
            raise RuntimeError(f"oracle minecraft renderer error {err}")
        r = MeshResult(m)
        self._L.orc_mesh_free(C.byref(m))
        return r

    def render_image(self, w, h):
        """glrender.ImageRendererSDF2.Render (image.go:76-118) + default conversion (:52-61), row by row."""
        bb = self.bb.astype(np.float32)
        f = np.float32
        dx, dy = f((bb[3] - bb[0]) / f(w)), f((bb[4] - bb[1]) / f(h))
        xmin = f(bb[0] + f(dx / f(2)))
        dist = np.empty((h, w), np.float32)
        for j in range(h):
            y = f(bb[4] - f(f(j) * dy))
            pos = np.empty((w, 2), np.float32)
            pos[:, 0] = (np.arange(w, dtype=np.float32) * dx).astype(np.float32) + xmin
            pos[:, 1] = y
            dist[j] = self.Evaluate(pos)
        rgba = np.zeros((h, w, 4), np.uint8)
        rgba[..., 3] = 255
        rgba[dist > 0, :3] = 255
        bad = ~np.isfinite(dist)
        rgba[bad] = (255, 0, 0, 255)
        return dist, rgba

    def normals_central_diff(self, pos, step):
        pos = np.ascontiguousarray(pos, np.float32)
        nrm = np.empty_like(pos)
        err = self._L.orc_normals_central_diff(self._h, self._pool, pos.ctypes.data, nrm.ctypes.data, pos.shape[0], np.float32(step))
        if err:
            raise RuntimeError(f"oracle normals error {err}")
        return nrm


def write_stl(tris):
    tris = np.ascontiguousarray(tris, np.float32).reshape(-1, 9)
    L = lib()
    buf = np.zeros(L.orc_stl_size(tris.shape[0]), np.uint8)
    err = L.orc_write_stl(tris.ctypes.data, tris.shape[0], buf.ctypes.data)
    if err:
        raise ValueError("empty triangle slice" if err == -1 else "too many triangles")
    return buf.tobytes()


def march_cubes(pos, dist, res):
    pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 8, 3)
    dist = np.ascontiguousarray(dist, np.float32).reshape(-1, 8)
    n = pos.shape[0]
    out = np.empty((5 * n, 3, 3), np.float32)
    nt = lib().orc_march_cubes(pos.ctypes.data, dist.ctypes.data, n, np.float32(res), out.ctypes.data)
    return out[:nt].copy()


def mc_tables():
    L = lib()
    e = np.ctypeslib.as_array(L.orc_mc_edge_table(), shape=(256,)).copy()
    t = np.ctypeslib.as_array(L.orc_mc_tri_table(), shape=(256, 16)).copy()
    return e, t


MATH_FN = {"hypot": 0, "atan2": 1, "sin": 2, "cos": 3, "acos": 4, "cbrt": 5, "sincos_s": 6, "sincos_c": 7,
           "min": 8, "max": 9, "pow13": 10, "round": 11, "floor": 12}


def math_apply(name, x, y=None):
    """Apply one of the oracle's math32 restatements elementwise (float32 in, float32 out)."""
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y if y is not None else np.zeros_like(x), np.float32)
    out = np.empty_like(x)
    lib().orc_math_apply(MATH_FN[name], x.ctypes.data, y.ctypes.data, out.ctypes.data, x.size)
    return out


def lsq_mgs64(A, b):
    """leastSquaresMGS64 (dual_contour_vertexplacement.go:152-223): float32 rows in, float32[3] out."""
    A = np.ascontiguousarray(A, np.float32).reshape(-1, 3)
    b = np.ascontiguousarray(b, np.float32)
    x = np.zeros(3, np.float32)
    lib().orc_lsq_mgs64(A.ctypes.data, b.ctypes.data, A.shape[0], x.ctypes.data)
    return x


class OracleBlockCachedSDF3:
    """gleval.BlockCachedSDF3 (gleval/gleval.go:110-218) restated over OracleSDF -- pure-Python loops, small cases only
    (test infrastructure like the rest of oracle/)."""

    def __init__(self, sdf, resX, resY, resZ):
        self.m = {}
        self.Reset(sdf, resX, resY, resZ)

    def Reset(self, sdf, resX, resY, resZ):                      # :126-146
        if resX <= 0 or resY <= 0 or resZ <= 0:
            raise ValueError("invalid resolution for BlockCachedSDF3")
        self.m.clear()
        self.sdf = sdf
        one = np.float32(1)
        self.mul = np.array([one / np.float32(resX), one / np.float32(resY), one / np.float32(resZ)], np.float32)
        self.hits = 0
        self.evals = 0

    def _key(self, p, bbmin):
        tp = self.mul * (p - bbmin)                              # MulElem(mul, Sub(p, bb.Min)) in float32
        return (int(tp[0]), int(tp[1]), int(tp[2]))              # Go int(float32): truncation toward zero

    def Evaluate(self, pos):                                     # :154-211
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
        if pos.shape[0] == 0:
            raise ValueError("empty buffers")
        bbmin = np.asarray(self.sdf.Bounds(), np.float32)[:3]
        dist = np.empty(pos.shape[0], np.float32)
        seek, idx = [], []
        for i in range(pos.shape[0]):
            k = self._key(pos[i], bbmin)
            if k in self.m:
                dist[i] = self.m[k]
            else:
                seek.append(pos[i])
                idx.append(i)
        if idx:
            sp = np.array(seek, np.float32)
            sd = self.sdf.Evaluate(sp)
            for i in range(len(idx)):
                self.m[self._key(sp[i], bbmin)] = sd[i]
            for i, d in enumerate(sd):
                dist[idx[i]] = d
        self.evals += pos.shape[0]
        self.hits += pos.shape[0] - len(idx)
        return dist
