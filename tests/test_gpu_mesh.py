"""GPU parity of the on-device mesher (octree prune + marching cubes + STL) through the C ABI.
Bar: triangle COUNT equal to the oracle's octree renderer (north_star) -- and, stronger, the sorted
triangle SET bit-identical (emission order is the only freedom a parallel mesher has)."""
import hashlib
import json
import os
import struct

import numpy as np
import pytest

from scaffold.builder import Builder
from oracle import oracle
from oracle.oracle import OracleSDF

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mesh_digests.json")))


def _sorted(t):
    t = np.ascontiguousarray(t, np.float32).reshape(-1, 9)
    return t[np.lexsort(t.view(np.uint32).T[::-1])]


def _digest(t):
    return hashlib.sha256(_sorted(t).tobytes()).hexdigest()


def test_sphere_marching_triangles_41072(gpu):
    b = Builder()
    oc = gpu.OctreeHIP(gpu.SDF3HIP(b.NewSphere(1.0)), np.float32(1.0 / 33), 4097)  # glrender_test.go:83-99
    assert oc.n_tris() == 41072
    assert _digest(oc.RenderAll()) == GOLD["sphere_r1_res1_33"]["sha256_sorted"]


@pytest.mark.parametrize("scene,key", [("npt-flange", "npt_flange_resdiv100"), ("npt-flange", "npt_flange_resdiv400"),
                                       ("bolt", "bolt_resdiv150"), ("knurled-cylinder", "knurled_cylinder_resdiv120")])
def test_scene_mesh_identical_to_oracle(gpu, scene, key):
    b = Builder()
    s = b.Scene(scene)
    g = GOLD[key]
    res = np.uint32(g["res_bits"]).view(np.float32)
    sdf = gpu.SDF3HIP(s)
    oc = gpu.OctreeHIP(sdf, res)
    assert oc.stats.levels == g["levels"]
    assert oc.n_tris() == g["n_tris"]                       # README.md:116,130 for npt-flange@400: 423,852
    tg = _sorted(oc.RenderAll())
    assert hashlib.sha256(tg.tobytes()).hexdigest() == g["sha256_sorted"]
    ref = OracleSDF(s.tree()).render_octree(res, 4096, True)
    assert (tg.view(np.uint32) == _sorted(ref.tris).view(np.uint32)).all()
    assert oc.TotalPruned() == ref.pruned                    # Octree.TotalPruned
    shared = gpu.OctreeHIP(sdf, res, share_corners=True)   # exact corner sharing: same triangles, fewer evaluations
    assert (_sorted(shared.RenderAll()).view(np.uint32) == tg.view(np.uint32)).all()
    assert shared.stats.evals <= oc.stats.evals             # strictly fewer when the 4-points-per-lane brick kernel applies
    # distinct z rows (share_corners = 2): the default's kernels, every bitwise-distinct row of a brick evaluated once -- same
    # records, same triangles, same statistics, fewer evaluations; as triangles and as packed records marched afterwards
    rows = gpu.OctreeHIP(sdf, res, share_corners=2)
    assert (_sorted(rows.RenderAll()).view(np.uint32) == tg.view(np.uint32)).all()
    assert rows.TotalPruned() == oc.TotalPruned()
    assert [int(getattr(rows.stats, k)) for k in ("n_tris", "leaf_cubes", "active_leaves", "cut_leaves", "evals_prune")] == \
           [int(getattr(oc.stats, k)) for k in ("n_tris", "leaf_cubes", "active_leaves", "cut_leaves", "evals_prune")]
    assert 5 * int(rows.stats.leaf_cubes) <= int(rows.stats.evals_leaf) <= int(oc.stats.evals_leaf) == 8 * int(oc.stats.leaf_cubes)
    if sdf.info()["leaf_k"] == 4 and not os.environ.get("GSDF_HIP_FUSED_LEAF"):
        assert rows.stats.evals_leaf < oc.stats.evals_leaf
        # ... and exactly what the oracle counts for the surviving bricks: 64 columns x the distinct z rows of each; every distinct
        # lattice point in passes of 256 lane slots (the interpreter's build of leaf_dense_kernel has no tail passes)
        assert int(rows.stats.evals_leaf) == ref.evals_rows and int(shared.stats.evals_leaf) == ref.evals_points_256
    elif sdf.info()["leaf_k"] == 4:
        assert int(shared.stats.evals_leaf) == ref.evals_points   # (the fused leaf_brick_kernel counts points, not lane slots)   # (a brick has 5..8 distinct rows; all eight only where every plane's two floats differ)
    if not os.environ.get("GSDF_HIP_FUSED_LEAF"):   # (packed records are the two-kernel leaf phase's; the fused kernel of tools/gpu_variants*.sh has none)
        rrec = gpu.OctreeHIP(sdf, res, share_corners=2, payload=gpu.PAYLOAD_RECORDS)
        assert rrec.payload()[0] == gpu.PAYLOAD_RECORDS and rrec.payload()[1] == int(oc.stats.cut_leaves)
        rrec.march()
        assert (_sorted(rrec.RenderAll()).view(np.uint32) == tg.view(np.uint32)).all()
        assert int(rrec.stats.evals_leaf) == int(rows.stats.evals_leaf)
        prec = gpu.OctreeHIP(sdf, res, share_corners=1, payload=gpu.PAYLOAD_RECORDS)   # distinct lattice points + packed records
        assert prec.payload()[1] == int(oc.stats.cut_leaves) and int(prec.stats.evals_leaf) == int(shared.stats.evals_leaf)
        prec.march()
        assert (_sorted(prec.RenderAll()).view(np.uint32) == tg.view(np.uint32)).all()
    # pruning must not change the surface (flat renderer == octree renderer in the reference's README)
    if key != "npt_flange_resdiv400":
        assert gpu.OctreeHIP(sdf, res, prune=False).n_tris() == g["n_tris"]


def test_read_triangles_iterator_contract(gpu):
    b = Builder()
    oc = gpu.OctreeHIP(gpu.SDF3HIP(b.NewSphere(1.0)), np.float32(1 / 8))
    with pytest.raises(BufferError):
        oc.ReadTriangles(np.zeros((4, 3, 3), np.float32))   # io.ErrShortBuffer (octreerenderer.go:132-134)
    buf = np.zeros((4096, 3, 3), np.float32)                # glrender.RenderAll: 4096-triangle chunks
    chunks, eof = [], False
    while not eof:
        n, eof = oc.ReadTriangles(buf)
        chunks.append(buf[:n].copy())
    allt = np.concatenate(chunks)
    assert allt.shape[0] == oc.n_tris()
    assert (allt == oc.RenderAll()).all()


def test_stl_matches_oracle_writer(gpu):
    b = Builder()
    s = b.Scene("npt-flange")
    oc = gpu.OctreeHIP(gpu.SDF3HIP(s), np.float32(float(s.Diagonal()) / 150))
    tris = oc.RenderAll()
    blob = oc.WriteBinarySTL()
    assert blob == oracle.write_stl(tris)                    # normals + records byte-identical (stl.go:15-62)
    assert struct.unpack_from("<I", blob, 80)[0] == oc.n_tris()
    for n in (1, 63, 64, 65, 255, 256, 257):                 # ragged record counts through the LDS staging
        sub = gpu.OctreeHIP(gpu.SDF3HIP(b.NewSphere(0.3 + n * 1e-3)), np.float32(0.2))
        assert sub.WriteBinarySTL() == oracle.write_stl(sub.RenderAll())


def test_sharded_union_equals_whole(gpu):
    b = Builder()
    s = b.Scene("npt-flange")
    res = np.float32(float(s.Diagonal()) / 300)
    sdf = gpu.SDF3HIP(s)
    whole = gpu.OctreeHIP(sdf, res)
    for world in (2, 3, 8):
        parts = [gpu.OctreeHIP(sdf, res, shard_rank=r, shard_count=world) for r in range(world)]
        counts = [p.n_tris() for p in parts]
        assert sum(counts) == whole.n_tris()
        assert max(counts) < 2.5 * (sum(counts) / world)     # round-robin bricks: reasonably balanced
        u = np.concatenate([p.RenderAll() for p in parts])
        assert (_sorted(u).view(np.uint32) == _sorted(whole.RenderAll()).view(np.uint32)).all()


def test_resolution_errors(gpu):
    b = Builder()
    sdf = gpu.SDF3HIP(b.NewSphere(1))
    for res in (0.0, -1.0, float("nan"), float("inf"), 100.0):
        with pytest.raises(gpu.HipError) as e:
            gpu.OctreeHIP(sdf, np.float32(res))
        assert e.value.code == -8
    with pytest.raises(ValueError):
        gpu.OctreeHIP(sdf, np.float32(0.1), 32)              # "bad octree eval buffer size"
    with pytest.raises(gpu.HipError) as e:
        gpu.OctreeHIP(sdf, np.float32(0.05), max_tris=10)    # device buffer too small is an error, not truncation
    assert e.value.code == -10


def test_tiny_and_awkward_resolutions(gpu):
    b = Builder()
    s = b.NewSphere(1.0)
    sdf = gpu.SDF3HIP(s)
    ref = OracleSDF(s.tree())
    for div in (1.2, 2, 3.5, 4, 4.000001, 8, 13, 37):       # glrender_test.go:115 resolutions + level 2/3 octrees
        res = np.float32(1.0 / div)
        oc = gpu.OctreeHIP(sdf, res)
        r = ref.render_octree(res, 4096, True)
        assert oc.stats.levels == r.levels
        assert oc.n_tris() == r.n_tris
        assert (_sorted(oc.RenderAll()).view(np.uint32) == _sorted(r.tris).view(np.uint32)).all()


def test_full_size_npt_flange_resdiv1600(gpu):
    """BASELINE.json configs[1] at full size: count + digest of the sorted triangle set against the
    committed oracle digest, plus size-independent properties (determinism, shard union, STL size)."""
    b = Builder()
    s = b.Scene("npt-flange")
    g = GOLD["npt_flange_resdiv1600"]
    res = np.uint32(g["res_bits"]).view(np.float32)
    assert res == np.float32(float(s.Diagonal()) / 1600)   # examples/npt-flange/flange.go:76-78
    sdf = gpu.SDF3HIP(s)
    oc = gpu.OctreeHIP(sdf, res)
    assert oc.stats.levels == 12 and oc.n_tris() == g["n_tris"]
    t = oc.RenderAll()
    assert _digest(t) == g["sha256_sorted"]
    oc2 = gpu.OctreeHIP(sdf, res, share_corners=True)
    assert oc2.n_tris() == oc.n_tris() and _digest(oc2.RenderAll()) == g["sha256_sorted"]   # idempotent / deterministic set
    halves = [gpu.OctreeHIP(sdf, res, shard_rank=r, shard_count=2) for r in range(2)]
    assert sum(h.n_tris() for h in halves) == g["n_tris"]
    assert len(oc.WriteBinarySTL()) == 84 + 50 * g["n_tris"]
    # every vertex lies on a leaf-cube edge of the lattice: within res of the surface
    d = OracleSDF(s.tree()).Evaluate(t.reshape(-1, 3)[::997])
    assert np.abs(d).max() < float(res)


@pytest.mark.parametrize("scene,key,levels", [("bolt", "bolt_resdiv2000", 12), ("knurled-cylinder", "knurled_cylinder_resdiv2000", 12)])
def test_full_size_other_single_gpu_configs(gpu, scene, key, levels):
    """BASELINE.json configs[2] and [3] at full size (resdiv 2000) on one GPU: count and digest of the sorted triangle set
    against the oracle's committed digest (tests/golden/make_golden.py --full), specialised kernels (what bench.py times)
    and interpreter kernels alike, plus the union of two brick shards."""
    b = Builder()
    s = b.Scene(scene)
    g = GOLD[key]
    res = np.uint32(g["res_bits"]).view(np.float32)
    assert res == np.float32(float(s.Diagonal()) / 2000)
    sdf = gpu.SDF3HIP(s)
    oc = gpu.OctreeHIP(sdf, res)
    assert oc.stats.levels == levels and oc.n_tris() == g["n_tris"]
    assert _digest(oc.RenderAll()) == g["sha256_sorted"]
    del oc
    sdf.specialize()
    oc = gpu.OctreeHIP(sdf, res)
    assert oc.n_tris() == g["n_tris"] and _digest(oc.RenderAll()) == g["sha256_sorted"]
    del oc
    halves = [gpu.OctreeHIP(sdf, res, shard_rank=r, shard_count=2) for r in range(2)]
    assert sum(h.n_tris() for h in halves) == g["n_tris"]
    assert _digest(np.concatenate([h.RenderAll() for h in halves])) == g["sha256_sorted"]
    del halves
    # (Round 2's centre tests -- the reference's predicate |d| >= size*sqrt3/2 at every level -- lost 5,154 of knurled-cylinder's
    # 20,711,943 triangles at this size: its twisted cutters are not a distance field.)
    if scene == "knurled-cylinder":
        assert gpu.OctreeHIP(sdf, res, assume_sdf=True).n_tris() == 20706789


@pytest.mark.parametrize("scene,resdiv,lossy", [("bolt", 700, False), ("knurled-cylinder", 600, True), ("npt-flange", 800, False)])
def test_centre_tests_drop_no_surface(gpu, scene, resdiv, lossy):
    """The default octree (every Level >= 3 cube tested against the field's bounds over it) returns the triangle count of the
    octree that tests nothing and visits every leaf -- 10^8..10^9 leaves at these sizes. The reference's predicate applied to
    every level does not where the field is not a distance field (knurled-cylinder: 220 triangles short at resdiv 600). The
    flat renderer is not the yardstick at these sizes: its lattice corner i is origin + res*i where the octree's leaf has
    (origin + res*(i-1)) + res, one ulp apart now and then, and a handful of near-degenerate triangles come and go (bolt at
    resdiv 700: 776,048 against 776,044)."""
    b = Builder()
    s = b.Scene(scene)
    sdf = gpu.SDF3HIP(s)
    sdf.specialize()
    res = np.float32(float(s.Diagonal()) / resdiv)
    full = gpu.OctreeHIP(sdf, res, prune=False).n_tris()
    assert gpu.OctreeHIP(sdf, res).n_tris() == full
    assert (gpu.OctreeHIP(sdf, res, assume_sdf=True).n_tris() != full) == lossy
    assert abs(gpu.FlatHIP(sdf, res).n_tris() - full) <= 8


def test_fibonacci_showerhead_known_answer(gpu):
    """The reference's second held answer (README.md:152,166: 309,872 triangles at resdiv 350 from both of its renderers)
    on the device: the flat renderer gives it, and so does the octree mesher with its DEFAULT options -- every Level >= 3 cube
    centre-tested against the field's bounds over the cube (interval mode, dev_ops.h: D_LIP_*). The field is not a distance
    field: the buttress thread jumps across the seams of the screw's sawtooth and the knurl is a 45-degree helix, so the
    reference's own predicate |d| >= size * sqrt3/2 applied to every level (GSDF_PRUNE_ASSUME_SDF) loses 23 triangles --
    on the device exactly as in the oracle."""
    b = Builder()
    s = b.Scene("fibonacci-showerhead")
    res = np.float32(float(s.Diagonal()) / 350)
    assert f"{float(res):.7f}" == "0.2979682"
    ge4, every = GOLD["showerhead_resdiv350_prune_ge4"], GOLD["showerhead_resdiv350_prune_all"]
    assert ge4["n_tris"] == 309872 and every["n_tris"] == 309849
    ref = OracleSDF(s.tree()).render_octree(res, 4096, True)
    assert ref.n_tris == 309872 and _digest(ref.tris) == ge4["sha256_sorted"]
    for spec in (False, True):
        sdf = gpu.SDF3HIP(s)
        if spec:
            sdf.specialize()
        fl = gpu.FlatHIP(sdf, res)
        assert fl.n_tris() == 309872 and fl.Evaluations() == 1512024
        oc = gpu.OctreeHIP(sdf, res)                                     # the default
        assert oc.stats.levels == 9 and oc.n_tris() == 309872 and _digest(oc.RenderAll()) == ge4["sha256_sorted"]
        assert oc.TotalPruned() == ref.pruned                            # cube for cube the oracle's decisions
        oc = gpu.OctreeHIP(sdf, res, prune=sum(1 << l for l in range(4, 22)), assume_sdf=True)   # the reference tests the top of the tree only
        assert oc.n_tris() == 309872 and _digest(oc.RenderAll()) == ge4["sha256_sorted"]
        oc = gpu.OctreeHIP(sdf, res, assume_sdf=True)
        assert oc.n_tris() == 309849 and _digest(oc.RenderAll()) == every["sha256_sorted"]
        assert gpu.OctreeHIP(sdf, res, prune=False).n_tris() == 309872


# ---------------- dual contouring on device ----------------
@pytest.mark.parametrize("chiseled", [False, True])
def test_dualcontour_identical_to_oracle(gpu, chiseled):
    b = Builder()
    cases = [(b.NewSphere(1.0), 1.0 / 8), (b.NewBox(2, 2, 2, 0), 2.0 / 8), (b.Scene("bolt"), 0.5), (b.Scene("npt-flange"), 0.9),
             (b.Scene("knurled-cylinder"), 0.8), (b.Union(b.NewTorus(1.0, 0.3), b.NewHexagonalPrism(0.4, 0.6)), 1.0 / 6)]
    for k, (sh, res) in enumerate(cases):
        res = np.float32(res)
        dc = gpu.DualContourHIP(gpu.SDF3HIP(sh), res, chiseled=chiseled)
        ref = OracleSDF(sh.tree()).render_dualcontour(res, chiseled)
        assert dc.stats.levels == ref.levels
        assert dc.n_tris() == ref.n_tris, (sh, dc.n_tris(), ref.n_tris)
        tg, tc = _sorted(dc.RenderAll()), _sorted(ref.tris)
        assert (tg.view(np.uint32) == tc.view(np.uint32)).all(), sh   # float64 QR reproduced bit for bit
        # the reference sweeps the whole cubic lattice; the device skips the blocks of cell origins that an interval evaluation over
        # the block proves to be farther than 2*res from the surface (kernels_dc.h: dc_block_test_kernel), and, for trees whose field
        # is bounded from below outside a box, the cells outside that box by more than that -- same kept cubes, fewer evaluations
        assert dc.stats.evals <= ref.evals, (k, dc.stats.evals, ref.evals)


def test_dualcontour_reference_tolerances(gpu):
    # glrender/dual_contour_test.go:140-295 on the device output
    b = Builder()
    for sh, res in ((b.NewSphere(1.0), 1.0 / 8), (b.NewBox(2, 2, 2, 0), 2.0 / 8)):
        dc = gpu.DualContourHIP(gpu.SDF3HIP(sh), np.float32(res))
        v = np.unique(dc.RenderAll().reshape(-1, 3), axis=0)
        d = np.abs(gpu.SDF3HIP(sh).Evaluate(v))
        assert d.max() <= 1.5 * res and d.mean() <= 1.5 * res / 4
    fine = gpu.DualContourHIP(gpu.SDF3HIP(b.Scene("npt-flange")), np.float32(0.25))  # 9 levels: 16.7 M lattice cells
    assert fine.n_tris() > 100000 and fine.stats.levels == 9
    assert len(fine.WriteBinarySTL()) == 84 + 50 * fine.n_tris()
    with pytest.raises(gpu.HipError):
        gpu.DualContourHIP(gpu.SDF3HIP(b.NewSphere(1)), np.float32(0.0005))  # > 11 levels


def test_zero_copy_device_view_for_gather(gpu):
    """gsdf_amd.gather wraps the device triangle buffer without a copy (what the RCCL gather sends)."""
    import torch
    from gsdf_amd.gather import tensor_from_dev_ptr
    b = Builder()
    oc = gpu.OctreeHIP(gpu.SDF3HIP(b.NewSphere(1.0)), np.float32(1 / 16))
    t = tensor_from_dev_ptr(oc.dev_ptr(), oc.n_tris(), torch.device("cuda", 0))
    assert t.shape == (oc.n_tris(), 9) and t.data_ptr() == oc.dev_ptr()
    assert (t.cpu().numpy().reshape(-1, 3, 3) == oc.RenderAll()).all()
    assert tensor_from_dev_ptr(0, 0, torch.device("cuda", 0)).shape == (0, 9)


def test_dualcontour_sharded_union_equals_whole(gpu):
    b = Builder()
    for sh, res in ((b.Scene("bolt"), 0.25), (b.Scene("npt-flange"), 0.5)):
        sdf = gpu.SDF3HIP(sh)
        whole = _sorted(gpu.DualContourHIP(sdf, np.float32(res)).RenderAll())
        for world in (2, 3, 8):
            parts = [gpu.DualContourHIP(sdf, np.float32(res), shard_rank=r, shard_count=world) for r in range(world)]
            u = _sorted(np.concatenate([p.RenderAll() for p in parts]))
            assert u.shape == whole.shape and (u.view(np.uint32) == whole.view(np.uint32)).all(), (world, u.shape, whole.shape)


def test_empty_and_degenerate_meshes(gpu):
    b = Builder()
    # a surface-free field inside the bounds: an offset sphere whose zero set is outside its (reference) Bounds()
    far = b.Offset(b.NewSphere(1.0), 10.0)      # d = |p| - 1 + 10 > 0 everywhere
    sdf = gpu.SDF3HIP(far)
    oc = gpu.OctreeHIP(sdf, np.float32(0.5))
    assert oc.n_tris() == 0 and oc.RenderAll().shape == (0, 3, 3)
    n, eof = oc.ReadTriangles(np.zeros((8, 3, 3), np.float32))
    assert n == 0 and eof
    with pytest.raises(gpu.HipError) as e:
        oc.WriteBinarySTL()                      # "empty triangle slice" (stl.go:16-18)
    assert e.value.code == -1
    assert OracleSDF(far.tree()).render_octree(np.float32(0.5)).n_tris == 0
    dc = gpu.DualContourHIP(sdf, np.float32(0.5))
    assert dc.n_tris() == 0
    # the coarsest legal octree: 2 levels (resolution just fine enough)
    s = b.NewSphere(1.0)
    o2 = gpu.OctreeHIP(gpu.SDF3HIP(s), np.float32(1.2))
    r2 = OracleSDF(s.tree()).render_octree(np.float32(1.2))
    assert o2.stats.levels == 2 == r2.levels and o2.n_tris() == r2.n_tris


def test_repeated_resets_reuse_buffers_and_stay_exact(gpu):
    b = Builder()
    s = b.Scene("npt-flange")
    sdf = gpu.SDF3HIP(s)
    oc = gpu.OctreeHIP(sdf, np.float32(float(s.Diagonal()) / 100))
    first = _digest(oc.RenderAll())
    for div in (160, 100, 220, 100):            # Octree.Reset with a new resolution (octreerenderer.go:71)
        oc.Reset(sdf, np.float32(float(s.Diagonal()) / div))
        if div == 100:
            assert _digest(oc.RenderAll()) == first
    assert sdf.Evaluations() > 0


def test_glyph_plate_wide_union(gpu):
    """Config-5-shaped workload (wide 2D union -> extrude -> plate): octree + dual contouring vs the oracle."""
    b = Builder()
    s = b.Scene("glyph-plate")
    sdf = gpu.SDF3HIP(s)
    ref = OracleSDF(s.tree())
    res = np.float32(float(s.Diagonal()) / 150)
    oc = gpu.OctreeHIP(sdf, res)
    r = ref.render_octree(res)
    assert oc.n_tris() == r.n_tris
    assert (_sorted(oc.RenderAll()).view(np.uint32) == _sorted(r.tris).view(np.uint32)).all()
    dc = gpu.DualContourHIP(sdf, np.float32(1.5))
    rd = ref.render_dualcontour(np.float32(1.5))
    assert dc.n_tris() == rd.n_tris
    assert (_sorted(dc.RenderAll()).view(np.uint32) == _sorted(rd.tris).view(np.uint32)).all()
    fine = gpu.DualContourHIP(sdf, np.float32(0.25))   # 10 levels: 134 M lattice cells, the reference's practical limit
    assert fine.stats.levels == 10 and fine.n_tris() > 200000


def test_cube_queue_overflow_grows_and_reruns(gpu, monkeypatch):
    """A survivor queue that is too small is detected on device (nothing is dropped silently), grown and the pass
    repeated: same mesh as with ample queues. Exercises the LDS-staged block append of prune_kernel at its capacity."""
    s = Builder().Scene("npt-flange")
    g = GOLD["npt_flange_resdiv400"]
    res = np.uint32(g["res_bits"]).view(np.float32)
    monkeypatch.setenv("GSDF_HIP_QCAP_MIN", "100")           # level 3 has 17 K survivors at this resolution
    for spec in (False, True):
        sdf = gpu.SDF3HIP(s)                                 # fresh handle: arenas start empty
        if spec:
            sdf.specialize()
        oc = gpu.OctreeHIP(sdf, res)
        assert oc.n_tris() == g["n_tris"]
        assert _digest(oc.RenderAll()) == g["sha256_sorted"]


def test_dualcontour_lists_regrow(gpu):
    """The cube and edge lists are sized from the handle's previous dual-contouring mesh (a floor of 2^20 cubes, in eight parts): a
    fine mesh after a coarse one on the same handle overflows its parts, learns their exact sizes from the counters and repeats --
    the same triangles as on a fresh handle, whose first guess is ample."""
    b = Builder()
    s = b.Scene("npt-flange")
    res = np.float32(float(s.Diagonal()) / 800)
    used = gpu.SDF3HIP(s)
    assert gpu.DualContourHIP(used, np.float32(float(s.Diagonal()) / 60)).n_tris() > 0
    again = gpu.DualContourHIP(used, res)
    fresh = gpu.DualContourHIP(gpu.SDF3HIP(s), res)
    assert again.stats.leaf_cubes == fresh.stats.leaf_cubes > (1 << 20)          # (more kept cubes than the floor: the first attempt overflowed)
    assert again.n_tris() == fresh.n_tris() and again.stats.evals == fresh.stats.evals
    assert _digest(again.RenderAll()) == _digest(fresh.RenderAll())


def test_dualcontour_exact_box_early_out(gpu):
    """A long thin exact-distance part fills a few percent of the reference's cubic lattice: cells farther than 2*res
    outside the part's box are decided without evaluation -- identical mesh, a fraction of the evaluations. The same
    tree also exercises the far-child skip of wide unions (D_SKIPFAR*)."""
    b = Builder()
    s = b.Scene("glyph-plate")
    res = np.float32(float(s.Diagonal()) / 96)
    ref = OracleSDF(s.tree()).render_dualcontour(res, False)
    for spec in (False, True):
        sdf = gpu.SDF3HIP(s)
        if spec:
            sdf.specialize()
        dc = gpu.DualContourHIP(sdf, res)
        assert dc.n_tris() == ref.n_tris
        assert (_sorted(dc.RenderAll()).view(np.uint32) == _sorted(ref.tris).view(np.uint32)).all()
        assert dc.stats.evals < ref.evals // 4, (dc.stats.evals, ref.evals)


def _text_plate(b, text="gsdf MI355X"):
    """BASELINE config 5's tree shape from the reference's own font: forge/textsdf line -> extrude -> union with a plate."""
    ttf = open(os.path.join(os.path.dirname(__file__), "golden", "iso-3098.ttf"), "rb").read()
    t2 = b.TextLine(ttf, text)
    bb = t2.Bounds()
    t3 = b.Extrude(t2, 0.12)
    w, h = float(bb[3] - bb[0]), float(bb[4] - bb[1])
    plate = b.Translate(b.NewBox(w + 0.3, h + 0.3, 0.06, 0.01), float(bb[0] + bb[3]) / 2, float(bb[1] + bb[4]) / 2, -0.08)
    return b.Union(t3, plate)


def test_dualcontour_twelve_levels(gpu):
    """The largest lattice dual contouring takes: 12 octree levels, a 2048^3 lattice, a 34 GB int32 index grid (sized for one
    288 GB device; 11 levels until round 2). Too large for the oracle: the unit sphere at res 2/1500 is checked on its own terms --
    vertices within a cell of the sphere, quads in pairs of triangles, and the union of three z-slabs bit-identical to the whole."""
    b = Builder()
    sdf = gpu.SDF3HIP(b.NewSphere(1.0))
    res = np.float32(2.0 / 1500)
    dc = gpu.DualContourHIP(sdf, res)
    assert dc.stats.levels == 12 and dc.n_tris() > 10_000_000 and dc.n_tris() % 2 == 0
    t = dc.RenderAll()
    r = np.linalg.norm(t.reshape(-1, 3).astype(np.float64), axis=1)
    assert np.abs(r - 1.0).max() < 2 * float(res)
    whole = _digest(t)
    del t
    parts = np.concatenate([gpu.DualContourHIP(sdf, res, shard_rank=k, shard_count=3).RenderAll() for k in range(3)])
    assert _digest(parts) == whole
    with pytest.raises(gpu.HipError):
        gpu.DualContourHIP(sdf, np.float32(2.0 / 2100))                        # 13 levels: refused, not attempted


def test_text_plate_from_reference_font(gpu):
    """forge/textsdf (host mirror) -> HIP evaluator / octree mesher / dual contouring, bit-identical to the oracle on
    the same tree, interpreter and specialised kernels. Wide 2-D union of translated glyph polygons (far-child skip)."""
    b = Builder()
    s = _text_plate(b)
    ref = OracleSDF(s.tree())
    rng = np.random.default_rng(23)
    bb = s.Bounds()
    pos = (bb[:3] + rng.random((40000, 3), np.float32) * (bb[3:] - bb[:3])).astype(np.float32)
    dref = ref.Evaluate(pos)
    res = np.float32(float(s.Diagonal()) / 160)
    mo = ref.render_octree(res, 4096, True)
    md = ref.render_dualcontour(np.float32(float(s.Diagonal()) / 96), False)
    assert mo.n_tris > 20000 and md.n_tris > 5000
    for spec in (False, True):
        sdf = gpu.SDF3HIP(s)
        if spec:
            sdf.specialize()
        d = sdf.Evaluate(pos)
        assert int((d.view(np.uint32) != dref.view(np.uint32)).sum()) == 0
        oc = gpu.OctreeHIP(sdf, res)
        assert oc.n_tris() == mo.n_tris
        assert (_sorted(oc.RenderAll()).view(np.uint32) == _sorted(mo.tris).view(np.uint32)).all()
        dc = gpu.DualContourHIP(sdf, np.float32(float(s.Diagonal()) / 96))
        assert dc.n_tris() == md.n_tris
        assert (_sorted(dc.RenderAll()).view(np.uint32) == _sorted(md.tris).view(np.uint32)).all()
        assert dc.stats.evals < md.evals // 4                # the exact-box early-out applies: text + box are exact fields


def test_zero_copy_host_views(gpu):
    """gsdf_hip_mesh_host_tris / gsdf_hip_mesh_host_stl: pinned host memory owned by the mesh, same bytes as the copying
    calls, for all three meshers; stable across repeated calls; kept alive by the arrays that look at it."""
    import ctypes as C
    import gc
    b = Builder()
    s = b.Scene("npt-flange")
    sdf = gpu.SDF3HIP(s)
    res = np.float32(float(s.Diagonal()) / 120)
    for mk in (lambda: gpu.OctreeHIP(sdf, res), lambda: gpu.FlatHIP(sdf, res), lambda: gpu.DualContourHIP(sdf, np.float32(res * 2))):
        m = mk()
        tv = m.triangles_view()
        assert tv.shape == (m.n_tris(), 3, 3) and not tv.flags.writeable
        assert (tv.view(np.uint32) == m.RenderAll().view(np.uint32)).all()
        assert m.triangles_view().ctypes.data == tv.ctypes.data           # same memory on the second call
        sv = m.stl_view()
        copy = np.empty(84 + 50 * m.n_tris(), np.uint8)                   # the copying entry point
        assert gpu.lib().gsdf_hip_mesh_stl(m._mesh, copy.ctypes.data, copy.size) == 0
        assert sv.size == copy.size and (sv == copy).all()
        assert sv.tobytes() == oracle.write_stl(np.array(tv))
        # the views keep the renderer (and with it the mesh's host memory) alive
        del m
        gc.collect()
        assert int(tv.view(np.uint32).sum(dtype=np.uint64)) >= 0 and sv[80:84].view(np.uint32)[0] == tv.shape[0]
    # host_output: the mesher writes into pinned host memory itself; same triangles, the view IS the output buffer.
    # A fresh handle at 1.7 M triangles starts from the default 1 M-triangle buffer: overflow -> exact rerun included.
    fresh = gpu.SDF3HIP(s)
    res8 = np.float32(float(s.Diagonal()) / 800)
    ho = gpu.OctreeHIP(fresh, res8, host_output=True)
    dv = gpu.OctreeHIP(fresh, res8)
    assert ho.n_tris() == dv.n_tris() > 1 << 20
    assert (_sorted(ho.triangles_view()).view(np.uint32) == _sorted(dv.RenderAll()).view(np.uint32)).all()
    assert (_sorted(ho.RenderAll()).view(np.uint32) == _sorted(dv.RenderAll()).view(np.uint32)).all()   # copying read of host memory
    assert ho.stl_view().tobytes() == oracle.write_stl(np.array(ho.triangles_view()))               # device STL build reads it back
    # empty mesh: no triangles to look at, and the reference refuses to write an empty STL (stl.go:19-21)
    empty = gpu.OctreeHIP(gpu.SDF3HIP(b.Offset(b.NewSphere(1.0), 10.0)), np.float32(0.5))   # d > 0 everywhere
    assert empty.n_tris() == 0 and empty.triangles_view().shape == (0, 3, 3)
    with pytest.raises(gpu.HipError, match="empty triangle slice"):
        empty.stl_view()


def test_dual_contouring_two_handles_of_one_tree_at_once(gpu):
    """What bench.py --renderer dualcontour does at N = 1: two handles of the same tree, a host thread and one blocking dual-contouring
    mesh in flight on each (a mesh is one chain with host round trips between its stages). Every mesh made that way is the mesh of a
    sequential call, bit for bit; interpreter and specialised handles side by side."""
    import threading
    b = Builder()
    s = b.Scene("npt-flange")
    res = np.float32(float(s.Diagonal()) / 160)
    want = _digest(gpu.DualContourHIP(gpu.SDF3HIP(s), res).RenderAll())
    handles = [gpu.SDF3HIP(s), gpu.SDF3HIP(s).specialize()]
    got, errs = [[], []], []

    def work(t):
        try:
            for _ in range(6):
                got[t].append(_digest(gpu.DualContourHIP(handles[t], res).RenderAll()))
        except Exception as e:  # noqa: BLE001 - reported below, in the main thread
            errs.append((t, repr(e)))
    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    assert got == [[want] * 6, [want] * 6]


def test_concurrent_meshing_from_host_threads(gpu):
    """Handles are independent (own stream, own workspace; buffer pools are locked): four host threads meshing four
    different scenes at once get the results of the sequential runs."""
    import threading
    b = Builder()
    jobs = [("npt-flange", 300), ("bolt", 260), ("knurled-cylinder", 220), ("npt-flange", 411)]
    want = []
    for name, rd in jobs:
        s = b.Scene(name)
        oc = gpu.OctreeHIP(gpu.SDF3HIP(s), np.float32(float(s.Diagonal()) / rd))
        want.append(_digest(oc.RenderAll()))
    got = [None] * len(jobs)
    errs = []

    def work(i):
        try:
            name, rd = jobs[i]
            s = Builder().Scene(name)
            sdf = gpu.SDF3HIP(s)
            if i % 2 == 0:
                sdf.specialize()            # hiprtc builds in parallel too
            res = np.float32(float(s.Diagonal()) / rd)
            for _ in range(3):
                oc = gpu.OctreeHIP(sdf, res)
                d = _digest(oc.triangles_view())
                assert got[i] in (None, d)
                got[i] = d
            fl = gpu.FlatHIP(sdf, np.float32(res * 2))
            assert fl.n_tris() > 0 and len(fl.stl_view()) == 84 + 50 * fl.n_tris()
        except Exception as e:  # noqa: BLE001 - reported below, in the main thread
            errs.append((i, repr(e)))

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    assert got == want


@pytest.mark.parametrize("nv,inner,offset", [(40, 0.55, 0.0), (64, 0.8, 0.0), (24, 0.3, 75.0), (12, 0.97, -3.0), (30, 0.6, 3000.0)])
def test_polygon_edge_culling_is_exact(gpu, nv, inner, offset):
    """poly_cull (interp.h): the leaf kernel skips polygon edges that cannot hold the minimum / cannot be crossed for any
    point of a brick. Stars and gears with fine teeth, meshed finely enough for the culling to bite, also far from the
    origin (large coordinates, same absolute feature size: the margin scales with the magnitudes involved)."""
    import math
    b = Builder()
    verts = []
    for i in range(nv):
        a = 2 * math.pi * i / nv
        r = 1.0 if i % 2 == 0 else inner
        verts.append((offset + r * math.cos(a), offset * 0.5 + r * math.sin(a)))
    part = b.Extrude(b.NewPolygon(verts), 0.6)
    if nv == 64:
        part = b.Difference(part, b.Translate(b.NewCylinder(0.3, 2.0, 0.0), offset, offset * 0.5, 0.0))
    res = np.float32(float(part.Diagonal()) / 260)
    ref = OracleSDF(part.tree())
    want = _sorted(ref.render_octree(res, 4096, True).tris)
    sdf = gpu.SDF3HIP(part)
    for spec in (False, True):
        if spec:
            sdf.specialize()
        oc = gpu.OctreeHIP(sdf, res)
        got = _sorted(oc.RenderAll())
        assert got.shape == want.shape and (got.view(np.uint32) == want.view(np.uint32)).all(), (nv, spec)
    rng = np.random.default_rng(nv)
    bb = part.Bounds()
    pos = (bb[:3] + rng.random((30000, 3), np.float32) * (bb[3:] - bb[:3])).astype(np.float32)
    assert (sdf.Evaluate(pos).view(np.uint32) == ref.Evaluate(pos).view(np.uint32)).all()


@pytest.mark.parametrize("turn,ncopies,twist", [(0.5, 9, 0.0), (np.pi / 4, 16, 0.15), (0.0, 7, 0.0)])
def test_circular_array_sector_gate_is_exact(gpu, turn, ncopies, twist):
    """D_CIRC_ORDER / D_GATEOB (compile.cpp: the circular array's sector gate): the wave evaluates the nearer of the array's two
    sector copies first and skips the other where its turned-box bound exceeds the first one's value. Gears of extruded star
    teeth (an expensive child with a turned box for a region), plain, twisted and with an axis-aligned tooth: distances and
    meshes bit-identical to the oracle, interpreter and specialised kernels (that the lowering contains the gate for such a
    child: tests/test_lowering.py::test_sector_gate_of_circular_arrays)."""
    b = Builder()
    star = b.NewPolygon([(1.2 * np.cos(t) * (1 if i % 2 else 0.5), 0.8 * np.sin(t) * (1 if i % 2 else 0.5)) for i, t in enumerate(np.linspace(0, 2 * np.pi, 12, endpoint=False))])
    tooth = b.Extrude(star, 3.0)
    if turn:
        tooth = b.Rotate(tooth, turn, (0, 0, 1))
    gear = b.CircularArray(b.Translate(tooth, 6.0, 0, 0), ncopies, ncopies)
    if twist:
        gear = b.Twist(gear, twist)
    part = b.SmoothUnion(0.2, b.NewCylinder(5.6, 2.0, 0.1), gear)
    ref = OracleSDF(part.tree())
    sdf = gpu.SDF3HIP(part)
    rng = np.random.default_rng(ncopies)
    bb = part.Bounds()
    pos = (bb[:3] + rng.random((40000, 3), np.float32) * (bb[3:] - bb[:3])).astype(np.float32)
    pos[:2000, :2] *= np.float32(0.02)                      # near the axis, where every sector is close
    dref = ref.Evaluate(pos)
    res = np.float32(float(part.Diagonal()) / 300)
    want = _sorted(ref.render_octree(res, 4096, True).tris)
    for spec in (False, True):
        if spec:
            sdf.specialize()
        assert (sdf.Evaluate(pos).view(np.uint32) == dref.view(np.uint32)).all(), spec
        got = _sorted(gpu.OctreeHIP(sdf, res).RenderAll())
        assert got.shape == want.shape and (got.view(np.uint32) == want.view(np.uint32)).all(), spec
    assert len(want) > 20000


def test_example_render_stl(gpu, tmp_path):
    """examples/render_stl.py: the reference's example flow (part -> mesh -> binary STL file) end to end."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("render_stl", os.path.join(os.path.dirname(os.path.dirname(__file__)), "examples", "render_stl.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = tmp_path / "flange.stl"
    assert mod.main(["npt-flange", "--resdiv", "400", "-o", str(out)]) == 0
    blob = out.read_bytes()
    assert struct.unpack_from("<I", blob, 80)[0] == 423852 and len(blob) == 84 + 50 * 423852   # README.md:116
    b = Builder()
    s = b.Scene("npt-flange")
    ref = OracleSDF(s.tree()).render_octree(np.float32(float(s.Diagonal()) / 400), 4096, True)
    # same triangles as the oracle's octree renderer, records in device emission order
    rec = np.frombuffer(blob, np.uint8, offset=84).reshape(-1, 50)
    tris = np.ascontiguousarray(rec[:, 12:48]).view(np.float32).reshape(-1, 9)
    assert (_sorted(tris).view(np.uint32) == _sorted(ref.tris).view(np.uint32)).all()


def test_octree_start_wait_three_in_flight(gpu):
    """gsdf_hip_mesh_octree_start / _wait: the next meshes' chains of kernels are enqueued while the previous mesh runs (three per
    program, a workspace and a stream each). Same meshes as the blocking call, at mixed resolutions and payloads, through capacity
    reruns (first meshes of a handle), interpreter and specialised; a fourth job and the other meshers are refused while jobs are in
    flight."""
    b = Builder()
    sh = b.Scene("npt-flange")
    for spec in (False, True):
        sdf = gpu.SDF3HIP(sh)
        if spec:
            sdf.specialize()
        rds = [90, 260, 140, 260, 400, 90, 140]
        ress = [np.float32(float(sh.Diagonal()) / rd) for rd in rds]
        want = [gpu.OctreeHIP(gpu.SDF3HIP(sh), r) for r in ress[:3]]
        want = {float(r): (w.n_tris(), w.TotalPruned(), int(w.stats.evals), _digest(w.RenderAll())) for r, w in zip(ress[:3], want)}
        want[float(ress[4])] = (423852, None, None, None)
        pend, got = [], []
        for k, r in enumerate(ress):
            pend.append(gpu.OctreeHIP.start(sdf, r, payload=gpu.PAYLOAD_RECORDS if k == 3 else gpu.PAYLOAD_TRIANGLES))
            if len(pend) == 3:
                if k == 2:                                            # three in flight: a fourth is refused, so are the other meshers
                    with pytest.raises(gpu.HipError):
                        gpu.OctreeHIP.start(sdf, r)
                    with pytest.raises(gpu.HipError):
                        gpu.FlatHIP(sdf, r)
                    with pytest.raises(gpu.HipError):
                        gpu.DualContourHIP(sdf, r)
                got.append(pend.pop(0).wait())
        while len(pend) > 1:
            got.append(pend.pop(0).wait())
        pend = pend[0]
        got.append(pend.wait())
        assert got[3].payload()[0] == gpu.PAYLOAD_RECORDS
        got[3].march()
        for r, oc in zip(ress, got):
            n, pr, ev, dg = want[float(r)]
            assert oc.n_tris() == n
            if dg is not None:
                assert oc.TotalPruned() == pr and int(oc.stats.evals) == ev and _digest(oc.RenderAll()) == dg
        gpu.OctreeHIP.start(sdf, ress[0])                             # dropped without wait(): completed and released by its finaliser
        assert gpu.FlatHIP(sdf, ress[0]).n_tris() > 0                 # ... after which the handle is free again


def test_minecraft_render_identical_to_oracle(gpu):
    """gsdf_hip_mesh_minecraft (glrender.minecraftRender, dual_contour.go:297-403): the reference's own case (unit sphere, res 1/4,
    glrender_test.go:55-81), a CSG part and a tree with negative coordinates throughout -- triangle sets bit-identical to the oracle's
    (which tests/test_oracle_golden.py holds to an independent statement), evaluation counts equal, interpreter and specialised
    evaluation kernels; errors like the other renderers'."""
    b = Builder()
    cases = [(b.NewSphere(1.0), np.float32(0.25)), (b.Scene("npt-flange"), None), (b.Translate(b.NewBox(1.0, 0.7, 0.5, 0.1), -3.0, -2.0, -1.5), np.float32(0.04))]
    for sh, res in cases:
        res = res if res is not None else np.float32(float(sh.Diagonal()) / 70)
        want = OracleSDF(sh.tree()).render_minecraft(res)
        for spec in (False, True):
            sdf = gpu.SDF3HIP(sh)
            if spec:
                sdf.specialize()
            m = gpu.MinecraftHIP(sdf, res)
            assert m.n_tris() == want.n_tris > 0 and int(m.stats.evals) == want.evals and int(m.stats.levels) == want.levels
            assert (_sorted(m.RenderAll()).view(np.uint32) == _sorted(want.tris).view(np.uint32)).all()
            assert sdf.Evaluations() >= want.evals
            assert len(m.WriteBinarySTL()) == 84 + 50 * want.n_tris
    sdf = gpu.SDF3HIP(b.NewSphere(1.0))
    with pytest.raises(gpu.HipError):
        gpu.MinecraftHIP(sdf, np.float32(0))            # "invalid renderer cube resolution"
    with pytest.raises(gpu.HipError):
        gpu.MinecraftHIP(sdf, np.float32(8.0))          # "resolution not fine enough ..."
    with pytest.raises(gpu.HipError):
        gpu.MinecraftHIP(sdf, np.float32(1e-3))         # more than 9 levels: every cube of the lattice would be evaluated
