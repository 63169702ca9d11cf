"""ctypes face of libgsdfhip.so (include/gsdf_hip.h): the MI355X HIP backend.

Mirrors the reference seam: `SDF3HIP` stands where `gleval.SDF3Compute` does (gleval/gpu.go:56-103:
Evaluate / Bounds / Evaluations, same error behaviour), `OctreeHIP` where `glrender.Octree` does
(glrender/octreerenderer.go: ReadTriangles iterator + RenderAll), `write_binary_stl` where
`glrender.WriteBinarySTL` does. There is NO CPU fallback: if the HIP library or a GPU is missing,
construction raises.
"""
import ctypes as C
import os
import numpy as np

from ._ctypes_common import GsdfTree

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libgsdfhip.so")
_LIB = None

ErrEmptyBuffers = "empty buffers"
ErrMismatchBufferLength = "position and distance buffer length mismatch"

# every symbol include/gsdf_hip.h declares
SYMBOLS = ["gsdf_hip_last_error", "gsdf_hip_init", "gsdf_hip_program_create", "gsdf_hip_program_destroy",
           "gsdf_hip_program_bounds", "gsdf_hip_program_is2d", "gsdf_hip_program_info", "gsdf_hip_evaluations", "gsdf_hip_lower", "gsdf_hip_lower_region", "gsdf_hip_eval3_submit", "gsdf_hip_eval_wait", "gsdf_hip_host_alloc", "gsdf_hip_host_register", "gsdf_hip_host_release", "gsdf_hip_comm_unique_id", "gsdf_hip_comm_create", "gsdf_hip_comm_rank", "gsdf_hip_comm_world", "gsdf_hip_comm_allreduce_sum_u64", "gsdf_hip_mesh_gatherv", "gsdf_hip_mesh_gatherv_start", "gsdf_hip_mesh_gatherv_wait", "gsdf_hip_comm_destroy", "gsdf_hip_selftest_div", "gsdf_hip_selftest_sqrt", "gsdf_hip_selftest_circ", "gsdf_hip_selftest_atan2", "gsdf_hip_selftest_cossin", "gsdf_hip_mesh_minecraft", "gsdf_hip_blockcache_create", "gsdf_hip_blockcache_reset", "gsdf_hip_blockcache_eval3", "gsdf_hip_blockcache_hits", "gsdf_hip_blockcache_evaluations", "gsdf_hip_blockcache_destroy", "gsdf_hip_program_specialize", "gsdf_hip_program_specialize_async", "gsdf_hip_program_specialize_poll", "gsdf_hip_program_is_specialized", "gsdf_hip_program_kernels", "gsdf_hip_specialize_source", "gsdf_hip_specialize_check",
           "gsdf_hip_eval3", "gsdf_hip_eval2", "gsdf_hip_eval3_dev", "gsdf_hip_eval2_dev", "gsdf_hip_normals3", "gsdf_hip_image2",
           "gsdf_hip_mesh_octree", "gsdf_hip_mesh_dualcontour", "gsdf_hip_mesh_flat", "gsdf_hip_mesh_stats_get", "gsdf_hip_mesh_read", "gsdf_hip_mesh_dev_tris",
           "gsdf_hip_mesh_stl", "gsdf_hip_mesh_host_tris", "gsdf_hip_mesh_host_stl", "gsdf_hip_mesh_destroy", "gsdf_hip_brick_owner", "gsdf_hip_slab_range",
           "gsdf_hip_mesh_payload", "gsdf_hip_mesh_march", "gsdf_hip_mesh_stage_ms", "gsdf_hip_mesh_octree_start", "gsdf_hip_mesh_octree_wait", "gsdf_hip_comm_transport", "gsdf_hip_gather_plan"]


PRUNE_ASSUME_SDF = 1 << 30  # gsdf_hip.h: GSDF_PRUNE_ASSUME_SDF


GATHER_ALL, GATHER_ROOT, GATHER_NONE = 0, 1, 2  # gsdf_hip.h
PAYLOAD_TRIANGLES, PAYLOAD_RECORDS = 0, 1            # gsdf_hip.h: gsdf_mesh_opts.payload
GOP_COPY, GOP_SEND, GOP_RECV = 0, 1, 2               # gsdf_hip.h: gsdf_gather_op.kind


class GatherOp(C.Structure):
    _fields_ = [("kind", C.c_int32), ("peer", C.c_int32), ("src_off", C.c_uint64), ("dst_off", C.c_uint64), ("bytes", C.c_uint64)]


class GatherStats(C.Structure):
    _fields_ = [("ms_counts", C.c_double), ("ms_payload", C.c_double), ("bytes_sent", C.c_uint64), ("bytes_received", C.c_uint64),
                ("ms_march", C.c_double)]


class MeshOpts(C.Structure):
    _fields_ = [("prune", C.c_int), ("shard_rank", C.c_int), ("shard_count", C.c_int), ("max_tris", C.c_uint64),
                ("stream", C.c_void_p), ("share_corners", C.c_int), ("host_output", C.c_int), ("payload", C.c_int), ("reserved", C.c_int)]


class MeshStats(C.Structure):
    _fields_ = [("n_tris", C.c_uint64), ("evals", C.c_uint64), ("pruned_leaves", C.c_uint64), ("leaf_cubes", C.c_uint64),
                ("active_leaves", C.c_uint64), ("levels", C.c_int), ("origin", C.c_float * 3), ("res", C.c_float),
                ("ms_total", C.c_double), ("ms_prune", C.c_double), ("ms_leaf", C.c_double), ("ms_march", C.c_double),
                ("evals_prune", C.c_uint64), ("evals_leaf", C.c_uint64), ("ms_emit", C.c_double), ("cut_leaves", C.c_uint64)]


class HipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"gsdf_hip error {code}: {msg}")
        self.code = code
        self.msg = msg


def lib():
    """Load libgsdfhip.so. Raises ImportError loudly if it was not built (never falls back)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: the HIP extension must be built "
                              "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.gsdf_hip_last_error.restype = C.c_char_p
        L.gsdf_hip_init.argtypes = [C.c_int]
        L.gsdf_hip_program_create.argtypes = [C.POINTER(GsdfTree), C.POINTER(C.c_void_p)]
        L.gsdf_hip_program_destroy.argtypes = [C.c_void_p]
        L.gsdf_hip_program_destroy.restype = None
        L.gsdf_hip_program_bounds.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.gsdf_hip_program_is2d.argtypes = [C.c_void_p]
        L.gsdf_hip_program_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.gsdf_hip_lower.argtypes = [C.POINTER(GsdfTree), C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.gsdf_hip_lower_region.argtypes = [C.POINTER(GsdfTree), C.c_uint32, C.POINTER(C.c_int), C.POINTER(C.c_float)]
        L.gsdf_hip_comm_unique_id.argtypes = [C.c_void_p]
        L.gsdf_hip_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.gsdf_hip_comm_rank.argtypes = [C.c_void_p]
        L.gsdf_hip_comm_world.argtypes = [C.c_void_p]
        L.gsdf_hip_comm_allreduce_sum_u64.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t]
        L.gsdf_hip_mesh_gatherv.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.gsdf_hip_mesh_gatherv_start.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.gsdf_hip_mesh_gatherv_wait.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(GatherStats)]
        L.gsdf_hip_comm_destroy.argtypes = [C.c_void_p]
        L.gsdf_hip_comm_transport.restype = C.c_char_p
        L.gsdf_hip_comm_transport.argtypes = [C.c_void_p]
        L.gsdf_hip_gather_plan.argtypes = [C.POINTER(C.c_uint64), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(GatherOp), C.c_size_t,
                                           C.POINTER(C.c_size_t), C.POINTER(C.c_uint64)]
        L.gsdf_hip_mesh_payload.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.gsdf_hip_mesh_march.argtypes = [C.c_void_p]
        L.gsdf_hip_mesh_stage_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_int)]
        L.gsdf_hip_eval3_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]
        L.gsdf_hip_eval_wait.argtypes = [C.c_void_p, C.c_int]
        L.gsdf_hip_host_alloc.restype = C.c_void_p
        L.gsdf_hip_host_alloc.argtypes = [C.c_size_t]
        L.gsdf_hip_host_register.argtypes = [C.c_void_p, C.c_size_t]
        L.gsdf_hip_host_release.argtypes = [C.c_void_p]
        L.gsdf_hip_selftest_div.argtypes = [C.c_float, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_float)]
        L.gsdf_hip_selftest_sqrt.argtypes = [C.POINTER(C.c_uint64)]
        L.gsdf_hip_selftest_circ.argtypes = [C.c_float, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.gsdf_hip_selftest_atan2.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.gsdf_hip_selftest_cossin.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.gsdf_hip_blockcache_create.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_void_p)]
        L.gsdf_hip_blockcache_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float]
        L.gsdf_hip_blockcache_eval3.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t]
        L.gsdf_hip_blockcache_hits.restype = C.c_uint64
        L.gsdf_hip_blockcache_hits.argtypes = [C.c_void_p]
        L.gsdf_hip_blockcache_evaluations.restype = C.c_uint64
        L.gsdf_hip_blockcache_evaluations.argtypes = [C.c_void_p]
        L.gsdf_hip_blockcache_destroy.restype = None
        L.gsdf_hip_blockcache_destroy.argtypes = [C.c_void_p]
        L.gsdf_hip_program_specialize.argtypes = [C.c_void_p]
        L.gsdf_hip_program_is_specialized.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.gsdf_hip_program_specialize_async.argtypes = [C.c_void_p]
        L.gsdf_hip_program_specialize_poll.argtypes = [C.c_void_p, C.c_int]
        L.gsdf_hip_program_kernels.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.gsdf_hip_specialize_source.argtypes = [C.POINTER(GsdfTree), C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.gsdf_hip_specialize_check.argtypes = [C.POINTER(GsdfTree), C.POINTER(C.c_size_t)]
        L.gsdf_hip_evaluations.restype = C.c_uint64
        L.gsdf_hip_evaluations.argtypes = [C.c_void_p]
        for f in (L.gsdf_hip_eval3, L.gsdf_hip_eval2):
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t]
        for f in (L.gsdf_hip_eval3_dev, L.gsdf_hip_eval2_dev):
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        L.gsdf_hip_normals3.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float]
        L.gsdf_hip_image2.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.gsdf_hip_mesh_octree.argtypes = [C.c_void_p, C.c_float, C.POINTER(MeshOpts), C.POINTER(C.c_void_p)]
        L.gsdf_hip_mesh_octree_start.argtypes = [C.c_void_p, C.c_float, C.POINTER(MeshOpts), C.POINTER(C.c_void_p)]
        L.gsdf_hip_mesh_octree_wait.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.gsdf_hip_mesh_dualcontour.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        L.gsdf_hip_mesh_flat.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        L.gsdf_hip_mesh_minecraft.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.POINTER(C.c_void_p)]
        L.gsdf_hip_mesh_stats_get.argtypes = [C.c_void_p, C.POINTER(MeshStats)]
        L.gsdf_hip_mesh_read.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
        L.gsdf_hip_mesh_dev_tris.restype = C.c_void_p
        L.gsdf_hip_mesh_dev_tris.argtypes = [C.c_void_p]
        L.gsdf_hip_mesh_stl.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.gsdf_hip_mesh_host_tris.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.gsdf_hip_mesh_host_stl.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.gsdf_hip_mesh_destroy.argtypes = [C.c_void_p]
        L.gsdf_hip_mesh_destroy.restype = None
        L.gsdf_hip_brick_owner.restype = C.c_uint32
        L.gsdf_hip_brick_owner.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.gsdf_hip_slab_range.restype = None
        L.gsdf_hip_slab_range.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        _LIB = L
    return _LIB


def _check(rc):
    if rc != 0:
        raise HipError(rc, lib().gsdf_hip_last_error().decode())


def lower(shader_or_tree):
    """Host-only lowering of a tree to the device instruction stream: (uint32 code array, lds_slots)."""
    tree = shader_or_tree.tree() if hasattr(shader_or_tree, "tree") else shader_or_tree
    n, sl = C.c_uint32(), C.c_uint32()
    _check(lib().gsdf_hip_lower(C.byref(tree), None, 0, C.byref(n), C.byref(sl)))
    code = np.zeros(n.value, np.uint32)
    _check(lib().gsdf_hip_lower(C.byref(tree), code.ctypes.data, n.value, C.byref(n), C.byref(sl)))
    return code, sl.value


def init(device=-1):
    """gleval.Init1x1GLFW analogue: make sure a HIP device is usable (raises otherwise)."""
    _check(lib().gsdf_hip_init(device))


class SDFHIP:
    """gleval.SDF3 / SDF2 implemented by the HIP interpreter kernel for one flattened tree."""

    def __init__(self, shader_or_tree, device=-1):
        L = lib()
        tree = shader_or_tree.tree() if hasattr(shader_or_tree, "tree") else shader_or_tree
        self._tree = tree
        if device >= 0:
            _check(L.gsdf_hip_init(device))
        h = C.c_void_p()
        _check(L.gsdf_hip_program_create(C.byref(tree), C.byref(h)))
        self._h = h
        self.is2d = bool(L.gsdf_hip_program_is2d(h))

    def close(self):
        if getattr(self, "_h", None):
            lib().gsdf_hip_program_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def Bounds(self):
        bb = (C.c_float * 6)()
        _check(lib().gsdf_hip_program_bounds(self._h, bb))
        return np.array(bb[:], np.float32)

    def Evaluations(self):
        return int(lib().gsdf_hip_evaluations(self._h))

    def info(self):
        a, b = C.c_uint32(), C.c_uint32()
        _check(lib().gsdf_hip_program_info(self._h, C.byref(a), C.byref(b)))
        t = C.c_double()
        sp = lib().gsdf_hip_program_is_specialized(self._h, C.byref(t))
        buf = C.create_string_buffer(512)
        _check(lib().gsdf_hip_program_kernels(self._h, buf, 512))
        kern = dict(kv.split("=") for kv in buf.value.decode().split())
        leaf_k = int(kern["leaf"].split("<")[1].split(",")[0]) if "leaf" in kern else 0  # points per lane of the leaf phase
        return {"code_words": a.value, "lds_slots": b.value, "specialized": bool(sp), "specialize_s": t.value, "kernels": kern, "leaf_k": leaf_k}

    def specialize(self):
        """Build (hiprtc) and switch to kernels specialised for this tree: same bits, no fetch/decode. Returns self."""
        _check(lib().gsdf_hip_program_specialize(self._h))
        return self

    def specialize_async(self):
        """Start the same build on a thread of the library's own and return: the handle works through the interpreter kernels until
        the specialised ones are ready and switches at its next call (gsdf_hip_program_specialize_async). Returns self."""
        _check(lib().gsdf_hip_program_specialize_async(self._h))
        return self

    def specialize_poll(self, wait=False):
        """True once the handle runs specialised kernels (wait=True: block until the background build has finished)."""
        rc = lib().gsdf_hip_program_specialize_poll(self._h, 1 if wait else 0)
        if rc < 0:
            _check(rc)
        return rc == 1

    def Evaluate(self, pos, dist=None, userData=None):
        """pos: (n,3)|(n,4)|(n,2) float32 host array (row stride = position stride); dist: (n,) float32."""
        pos = np.asarray(pos, np.float32)
        if pos.ndim != 2 or not pos.flags.c_contiguous:
            pos = np.ascontiguousarray(pos.reshape(-1, 2 if self.is2d else 3))
        n = pos.shape[0]
        if dist is None:
            dist = np.empty(n, np.float32)
        elif not (isinstance(dist, np.ndarray) and dist.dtype == np.float32 and dist.ndim == 1 and dist.flags.c_contiguous and dist.flags.writeable):
            # the C side writes n float32 values at dist's address: anything else would be silently corrupted
            raise ValueError("dist must be a writeable, C-contiguous, 1-D float32 array")
        f = lib().gsdf_hip_eval2 if self.is2d else lib().gsdf_hip_eval3
        _check(f(self._h, pos.ctypes.data, pos.strides[0], n, dist.ctypes.data, dist.shape[0]))
        return dist

    def submit(self, pos, dist):
        """Pipelined Evaluate: returns a ticket at once; wait(ticket) blocks until `dist` holds the distances. pos / dist
        must stay alive and untouched until then (C-contiguous float32 arrays)."""
        if pos.dtype != np.float32 or dist.dtype != np.float32 or not pos.flags.c_contiguous or not dist.flags.c_contiguous or dist.ndim != 1:
            raise ValueError("submit needs C-contiguous float32 arrays (pos (n,3|4), dist (n,))")
        t = C.c_int(-1)
        _check(lib().gsdf_hip_eval3_submit(self._h, pos.ctypes.data, pos.strides[0], pos.shape[0], dist.ctypes.data, dist.shape[0], C.byref(t)))
        return t.value

    def wait(self, ticket):
        _check(lib().gsdf_hip_eval_wait(self._h, ticket))

    def evaluate_dev(self, d_pos_ptr, stride_bytes, d_dist_ptr, n, stream=None):
        """Device-resident evaluation on raw device pointers (e.g. torch tensors' data_ptr())."""
        f = lib().gsdf_hip_eval2_dev if self.is2d else lib().gsdf_hip_eval3_dev
        _check(f(self._h, d_pos_ptr, stride_bytes, d_dist_ptr, n, stream))

    def render_image(self, w, h):
        """glrender.ImageRendererSDF2.Render with the default conversion: (dist (h,w) float32, rgba (h,w,4) uint8)."""
        dist = np.empty((h, w), np.float32)
        rgba = np.empty((h, w, 4), np.uint8)
        _check(lib().gsdf_hip_image2(self._h, w, h, dist.ctypes.data, rgba.ctypes.data))
        return dist, rgba

    def normals(self, pos, step):
        pos = np.ascontiguousarray(pos, np.float32)
        out = np.empty_like(pos)
        _check(lib().gsdf_hip_normals3(self._h, pos.ctypes.data, out.ctypes.data, pos.shape[0], step))
        return out


def host_array(shape, dtype=np.float32):
    """numpy array over pinned, device-mapped host memory (gsdf_hip_host_alloc): Evaluate calls on such arrays make no
    staging copy -- the kernel reads / writes them across PCIe. Freed when the array (and its views) are collected."""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    ptr = lib().gsdf_hip_host_alloc(max(n, 1))
    if not ptr:
        raise MemoryError("gsdf_hip_host_alloc failed")

    class _Owner:
        def __init__(self, p): self.p = p
        def __del__(self):
            try: lib().gsdf_hip_host_release(self.p)
            except Exception: pass
    raw = (C.c_ubyte * max(n, 1)).from_address(ptr)
    raw._owner = _Owner(ptr)
    return np.frombuffer(raw, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


SDF3HIP = SDFHIP
SDF2HIP = SDFHIP


class BlockCachedSDF3HIP:
    """gleval.BlockCachedSDF3 (gleval/gleval.go:110-218) in front of a HIP evaluator: Reset / Evaluate / CacheHits /
    Evaluations / Bounds with the reference's semantics (lossy per-cell cache, misses evaluated in one batch)."""

    def __init__(self, sdf, resX, resY, resZ):
        self._h = C.c_void_p()
        self.sdf = sdf
        _check(lib().gsdf_hip_blockcache_create(sdf._h, resX, resY, resZ, C.byref(self._h)))

    def Reset(self, sdf, resX, resY, resZ):
        _check(lib().gsdf_hip_blockcache_reset(self._h, sdf._h, resX, resY, resZ))
        self.sdf = sdf

    def Evaluate(self, pos, dist=None, userData=None):
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
        if dist is None:
            dist = np.empty(pos.shape[0], np.float32)
        _check(lib().gsdf_hip_blockcache_eval3(self._h, pos.ctypes.data, 12, pos.shape[0], dist.ctypes.data, dist.shape[0]))
        return dist

    def CacheHits(self):
        return int(lib().gsdf_hip_blockcache_hits(self._h))

    def Evaluations(self):
        return int(lib().gsdf_hip_blockcache_evaluations(self._h))

    def Bounds(self):
        return self.sdf.Bounds()

    def __del__(self):
        try:
            if self._h:
                lib().gsdf_hip_blockcache_destroy(self._h)
                self._h = None
        except Exception:
            pass


class OctreeHIP:
    """glrender.Octree drop-in: octree pruning + marching cubes on device, triangles drained on demand.

    NewOctreeRenderer(s, cubeResolution, evalBufferSize) -> OctreeHIP(sdf, res) ; the evaluation buffer
    argument has no meaning here (positions are generated on device)."""

    def __init__(self, sdf, res, evalBufferSize=64, prune=True, shard_rank=0, shard_count=1, max_tris=0, stream=None,
                 share_corners=False, host_output=False, assume_sdf=False, payload=PAYLOAD_TRIANGLES):
        if evalBufferSize < 64:
            raise ValueError("bad octree eval buffer size")
        self.sdf = sdf
        self._mesh = None
        self._cursor = 0
        # share_corners: False / 0 every corner of every leaf (the reference's evaluations), True / 1 the distinct lattice points
        # of a brick (leaf_dense_kernel), 2 the distinct z rows of a brick (same kernels, same triangles, fewer evaluations)
        # prune: True / False, or an int bit mask of the octree levels to centre-test (bit L = Level L >= 3);
        # assume_sdf: the reference's predicate verbatim instead of the field's bounds over the cube (gsdf_hip.h)
        self._opts = MeshOpts(int(prune) | (PRUNE_ASSUME_SDF if assume_sdf else 0), shard_rank, shard_count, max_tris, stream, int(share_corners), int(host_output), int(payload), 0)
        self.Reset(sdf, res)

    def Reset(self, sdf, res):
        self._free()
        self.sdf = sdf
        m = C.c_void_p()
        _check(lib().gsdf_hip_mesh_octree(sdf._h, np.float32(res), C.byref(self._opts), C.byref(m)))
        self._adopt(m)

    def _adopt(self, m):
        self._mesh = m
        self._cursor = 0
        st = MeshStats()
        _check(lib().gsdf_hip_mesh_stats_get(m, C.byref(st)))
        self.stats = st

    @classmethod
    def start(cls, sdf, res, **kw):
        """gsdf_hip_mesh_octree_start: enqueue the mesh and return a PendingMesh; .wait() gives the OctreeHIP. Up to three per
        program in flight -- start the next one before waiting for the previous one and the GPU never idles between meshes."""
        self = cls.__new__(cls)
        self.sdf, self._mesh, self._cursor = sdf, None, 0
        kw.setdefault("prune", True)
        self._opts = MeshOpts(int(kw["prune"]) | (PRUNE_ASSUME_SDF if kw.get("assume_sdf") else 0), kw.get("shard_rank", 0), kw.get("shard_count", 1),
                              kw.get("max_tris", 0), kw.get("stream"), int(kw.get("share_corners", False)), int(kw.get("host_output", False)),
                              int(kw.get("payload", PAYLOAD_TRIANGLES)), 0)
        j = C.c_void_p()
        _check(lib().gsdf_hip_mesh_octree_start(sdf._h, np.float32(res), C.byref(self._opts), C.byref(j)))
        return PendingMesh(self, j)

    def _free(self):
        if getattr(self, "_mesh", None):
            lib().gsdf_hip_mesh_destroy(self._mesh)
            self._mesh = None

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass

    def TotalPruned(self):
        return int(self.stats.pruned_leaves)

    def payload(self):
        """(kind, records, payload bytes): PAYLOAD_TRIANGLES, or PAYLOAD_RECORDS -- packed cut-leaf records, no triangles yet."""
        n, b = C.c_uint64(), C.c_uint64()
        k = lib().gsdf_hip_mesh_payload(self._mesh, C.byref(n), C.byref(b))
        _check(min(k, 0))
        return k, int(n.value), int(b.value)

    def stage_ms(self):
        """{kernel name: device milliseconds} of the mesher's stages, where it records them (dual contouring)."""
        ms, names, n = (C.c_double * 8)(), (C.c_char_p * 8)(), C.c_int()
        _check(lib().gsdf_hip_mesh_stage_ms(self._mesh, ms, names, 8, C.byref(n)))
        return {names[k].decode(): float(ms[k]) for k in range(n.value)}

    def march(self):
        """Marching cubes over a records mesh, in place: afterwards it reads like any mesh (gsdf_hip_mesh_march)."""
        _check(lib().gsdf_hip_mesh_march(self._mesh))
        return self

    def n_tris(self):
        return int(self.stats.n_tris)

    def dev_ptr(self):
        return lib().gsdf_hip_mesh_dev_tris(self._mesh)

    def ReadTriangles(self, dst, userData=None):
        """dst: (k,3,3) float32. Returns (n, eof) like (n, io.EOF); needs len(dst) >= 5 (io.ErrShortBuffer)."""
        if dst.shape[0] < 5:
            raise BufferError("short buffer")
        remaining = self.n_tris() - self._cursor
        n = min(remaining, dst.shape[0])
        if n:
            _check(lib().gsdf_hip_mesh_read(self._mesh, self._cursor, n, dst.ctypes.data))
        self._cursor += n
        return n, self._cursor >= self.n_tris()

    def RenderAll(self):
        """glrender.RenderAll: all triangles as (n,3,3) float32."""
        n = self.n_tris()
        out = np.empty((n, 3, 3), np.float32)
        if n:
            _check(lib().gsdf_hip_mesh_read(self._mesh, 0, n, out.ctypes.data))
        return out

    def WriteBinarySTL(self):
        """glrender.WriteBinarySTL of the device-resident triangles, records built on device."""
        return self.stl_view().tobytes()

    def _view(self, ptr, nbytes, dtype):
        # numpy array over memory the mesh owns; the ctypes object in its base chain keeps this renderer alive
        raw = (C.c_ubyte * nbytes).from_address(ptr)
        raw._owner = self
        a = np.frombuffer(raw, dtype=dtype)
        a.flags.writeable = False
        return a

    def triangles_view(self):
        """All triangles as a read-only (n,3,3) float32 array over pinned host memory owned by the mesh (one DMA, no
        copy). Valid until this renderer is Reset or freed."""
        n = self.n_tris()
        if n == 0:
            return np.empty((0, 3, 3), np.float32)
        p = C.c_void_p()
        _check(lib().gsdf_hip_mesh_host_tris(self._mesh, C.byref(p)))
        return self._view(p.value, n * 36, np.float32).reshape(n, 3, 3)

    def stl_view(self):
        """The complete binary STL file as a read-only uint8 array over pinned host memory owned by the mesh
        (f.write(view) writes it without another copy). Valid until this renderer is Reset or freed."""
        p, ln = C.c_void_p(), C.c_size_t()
        _check(lib().gsdf_hip_mesh_host_stl(self._mesh, C.byref(p), C.byref(ln)))
        return self._view(p.value, ln.value, np.uint8)


class PendingMesh:
    """A mesh whose chain of kernels is enqueued (OctreeHIP.start): wait() returns the finished OctreeHIP."""

    def __init__(self, oc, job):
        self._oc, self._j = oc, job

    def wait(self):
        j, self._j = self._j, None
        m = C.c_void_p()
        _check(lib().gsdf_hip_mesh_octree_wait(j, C.byref(m)))
        self._oc._adopt(m)
        return self._oc

    def __del__(self):
        try:
            if self._j:
                m = C.c_void_p()
                if lib().gsdf_hip_mesh_octree_wait(self._j, C.byref(m)) == 0 and m.value:
                    lib().gsdf_hip_mesh_destroy(m)
                self._j = None
        except Exception:
            pass


class CommHIP:
    """RCCL communicator of the multi-GPU mesher (one process per GPU; gsdf_hip_comm_* in include/gsdf_hip.h).
    Rank 0 draws CommHIP.unique_id() and ships the 128 bytes to the other ranks by any means; every rank then builds
    CommHIP(id, rank, world) after hip.init(device) -- a collective call."""
    ID_BYTES = 128

    @staticmethod
    def unique_id():
        buf = (C.c_ubyte * CommHIP.ID_BYTES)()
        _check(lib().gsdf_hip_comm_unique_id(buf))
        return bytes(buf)

    def __init__(self, unique_id, rank, world):
        if len(unique_id) != CommHIP.ID_BYTES:
            raise ValueError("unique id must be %d bytes" % CommHIP.ID_BYTES)
        buf = (C.c_ubyte * CommHIP.ID_BYTES).from_buffer_copy(unique_id)
        h = C.c_void_p()
        _check(lib().gsdf_hip_comm_create(buf, rank, world, C.byref(h)))
        self._h, self.rank, self.world = h, rank, world

    def transport(self):
        """"rccl" or "loopback" (GSDF_HIP_COMM=loopback at unique_id() time: ranks are threads of one process)."""
        return lib().gsdf_hip_comm_transport(self._h).decode()

    def allreduce_sum(self, values):
        """Sum of each of `values` (ints) over all ranks. Collective."""
        arr = (C.c_uint64 * len(values))(*[int(v) for v in values])
        _check(lib().gsdf_hip_comm_allreduce_sum_u64(self._h, arr, len(values)))
        return [int(v) for v in arr]

    def close(self):
        if getattr(self, "_h", None):
            lib().gsdf_hip_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GatheredMeshHIP(OctreeHIP):
    """All ranks' triangles, rank-major, device resident on every rank: the result of <renderer>.gatherv(comm)
    (gsdf_hip_mesh_gatherv: RCCL all-gather of the counts + one group of broadcasts, no padding, no staging copies).
    Reads like any renderer: n_tris / RenderAll / ReadTriangles / triangles_view / WriteBinarySTL / dev_ptr."""

    def __init__(self, mesh, counts, stats):
        self.sdf = None
        self._mesh = mesh
        self._cursor = 0
        self.counts = counts
        self.stats = stats

    def Reset(self, sdf, res):
        raise TypeError("a gathered mesh is a result, not a renderer")


def _gatherv(self, comm):
    """RCCL all-gatherv of this rank's triangles with those of the other ranks of `comm`. Collective."""
    out = C.c_void_p()
    counts = (C.c_uint64 * comm.world)()
    _check(lib().gsdf_hip_mesh_gatherv(self._mesh, comm._h, C.byref(out), counts))
    st = MeshStats()
    _check(lib().gsdf_hip_mesh_stats_get(out, C.byref(st)))
    return GatheredMeshHIP(out, [int(c) for c in counts], st)


OctreeHIP.gatherv = _gatherv


class PendingGather:
    """A gather whose payload is moving (gsdf_hip_mesh_gatherv_start): wait() returns (GatheredMeshHIP or None, counts,
    GatherStats). The source renderer may be Reset or freed meanwhile: the library keeps the buffers the gather reads until
    the payload has moved (a destroy in between is deferred)."""

    def __init__(self, src, comm, handle):
        self._src, self._comm, self._h = src, comm, handle

    def wait(self):
        out = C.c_void_p()
        counts = (C.c_uint64 * self._comm.world)()
        gs = GatherStats()
        h, self._h = self._h, None
        _check(lib().gsdf_hip_mesh_gatherv_wait(h, C.byref(out), counts, C.byref(gs)))
        self._src = None
        g = None
        if out.value:
            st = MeshStats()
            _check(lib().gsdf_hip_mesh_stats_get(out, C.byref(st)))
            g = GatheredMeshHIP(out, [int(c) for c in counts], st)
        return g, [int(c) for c in counts], gs

    def __del__(self):
        try:
            if self._h:
                lib().gsdf_hip_mesh_gatherv_wait(self._h, None, None, None)
                self._h = None
        except Exception:
            pass


def _gatherv_start(self, comm, mode=GATHER_ALL, root=0):
    """Collective. Exchanges the counts, enqueues the payload (ALL: everyone gets everything, ROOT: only `root`, NONE: counts
    only) and returns a PendingGather; mesh the next part, then .wait()."""
    h = C.c_void_p()
    _check(lib().gsdf_hip_mesh_gatherv_start(self._mesh, comm._h, int(mode), int(root), C.byref(h)))
    return PendingGather(self, comm, h)


OctreeHIP.gatherv_start = _gatherv_start


class DualContourHIP(OctreeHIP):
    """glrender.DualContourRenderer with DualContourLeastSquares on device (Reset + RenderAll)."""

    def __init__(self, sdf, res, chiseled=False, stream=None, shard_rank=0, shard_count=1):
        self._shard = (shard_rank, shard_count)
        self.sdf = sdf
        self._mesh = None
        self._cursor = 0
        self._chiseled = bool(chiseled)
        self._stream = stream
        self.Reset(sdf, res)

    def Reset(self, sdf, res):
        self._free()
        self.sdf = sdf
        m = C.c_void_p()
        _check(lib().gsdf_hip_mesh_dualcontour(sdf._h, np.float32(res), int(self._chiseled), self._shard[0], self._shard[1],
                                               self._stream, C.byref(m)))
        self._mesh = m
        self._cursor = 0
        st = MeshStats()
        _check(lib().gsdf_hip_mesh_stats_get(m, C.byref(st)))
        self.stats = st


class MinecraftHIP(OctreeHIP):
    """glrender.minecraftRender (dual_contour.go:297-403) on device: the axis-aligned faces between level-1 cubes whose origins lie on
    different sides of the surface."""

    def __init__(self, sdf, res, stream=None):
        self.sdf = sdf
        self._mesh = None
        self._cursor = 0
        m = C.c_void_p()
        _check(lib().gsdf_hip_mesh_minecraft(sdf._h, np.float32(res), stream, C.byref(m)))
        self._mesh = m
        st = MeshStats()
        _check(lib().gsdf_hip_mesh_stats_get(m, C.byref(st)))
        self.stats = st


class FlatHIP(OctreeHIP):
    """glrender.FlatRenderer drop-in (NewFlatRenderer(s, cubeResolution, evalBufferSize, numParallel)): the whole corner
    lattice evaluated into a device-resident grid, then marching cubes of every cube. evalBufferSize / numParallel keep
    the reference's argument checks (flatrenderer.go:37-45) and have no other meaning on the device."""

    def __init__(self, sdf, res, evalBufferSize=4096, numParallel=1, stream=None, shard_rank=0, shard_count=1):
        if evalBufferSize < 8:
            raise ValueError("flat renderer eval buffer size must be at least 8")
        if numParallel < 1:
            raise ValueError("flat renderer numParallel must be at least 1")
        self._shard = (shard_rank, shard_count)
        self.sdf = sdf
        self._mesh = None
        self._cursor = 0
        self._stream = stream
        self.Reset(sdf, res)

    def Reset(self, sdf, res):
        self._free()
        self.sdf = sdf
        m = C.c_void_p()
        _check(lib().gsdf_hip_mesh_flat(sdf._h, np.float32(res), self._shard[0], self._shard[1], self._stream, C.byref(m)))
        self._mesh = m
        self._cursor = 0
        st = MeshStats()
        _check(lib().gsdf_hip_mesh_stats_get(m, C.byref(st)))
        self.stats = st

    def Evaluations(self):
        return int(self.stats.evals)


def gather_plan(bytes_per_rank, rank, mode=GATHER_ALL, root=0):
    """gsdf_hip_gather_plan (host only, pure): ([(kind, peer, src_off, dst_off, bytes), ...], total_bytes) -- the transfers rank
    `rank` performs in a gather of payloads of bytes_per_rank[r] bytes, and the size of its gathered buffer."""
    world = len(bytes_per_rank)
    arr = (C.c_uint64 * world)(*[int(b) for b in bytes_per_rank])
    ops = (GatherOp * (2 * world + 1))()
    n, total = C.c_size_t(), C.c_uint64()
    _check(lib().gsdf_hip_gather_plan(arr, world, int(rank), int(mode), int(root), ops, len(ops), C.byref(n), C.byref(total)))
    return [(o.kind, o.peer, int(o.src_off), int(o.dst_off), int(o.bytes)) for o in ops[:n.value]], int(total.value)
