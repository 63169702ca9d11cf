"""GPU parity of the run-time specialised kernels (gsdf_hip_program_specialize): bit-identical to the oracle and
to the interpreter kernels on every node type, on the golden vectors and on the meshes."""
import hashlib
import json
import os

import numpy as np
import pytest

import corpus
from par import pmap
from scaffold.builder import Builder
from oracle.oracle import OracleSDF

pytestmark = pytest.mark.gpu
GOLDDIR = os.path.join(os.path.dirname(__file__), "golden")
GOLD = json.load(open(os.path.join(GOLDDIR, "mesh_digests.json")))


def _mismatch(a, b):
    return int(((a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))).sum())


def _sorted(t):
    t = np.ascontiguousarray(t, np.float32).reshape(-1, 9)
    return t[np.lexsort(t.view(np.uint32).T[::-1])]


@pytest.mark.parametrize("which", ["3d", "2d"])
def test_corpus_specialised_bit_exact(gpu, which):
    """All 53 node types: specialised Evaluate == oracle == committed golden distances, bit for bit."""
    gold = np.load(os.path.join(GOLDDIR, "corpus_distances.npz"))
    gold = {k: gold[k] for k in gold.files}
    _, shapes = (corpus.shapes3d if which == "3d" else corpus.shapes2d)()

    def check(item):                                          # (one build per shape: side by side, tests/par.py)
        name, sh = item
        sdf = gpu.SDFHIP(sh).specialize()
        assert sdf.info()["specialized"]
        pos = corpus.sample_points(sh)
        assert _mismatch(sdf.Evaluate(pos), OracleSDF(sh.tree()).Evaluate(pos)) == 0, name
        assert _mismatch(sdf.Evaluate(gold["pos_" + name]), gold["dist_" + name]) == 0, name
    pmap(check, shapes, workers=12)


@pytest.mark.parametrize("scene,key", [("npt-flange", "npt_flange_resdiv400"), ("bolt", "bolt_resdiv150"),
                                       ("knurled-cylinder", "knurled_cylinder_resdiv120")])
def test_specialised_mesh_identical(gpu, scene, key):
    s = Builder().Scene(scene)
    g = GOLD[key]
    res = np.uint32(g["res_bits"]).view(np.float32)
    sdf = gpu.SDF3HIP(s).specialize()
    oc = gpu.OctreeHIP(sdf, res)
    assert oc.n_tris() == g["n_tris"]                       # npt-flange@400: 423,852 (README.md:116,130)
    tg = _sorted(oc.RenderAll())
    assert hashlib.sha256(tg.tobytes()).hexdigest() == g["sha256_sorted"]
    ref = OracleSDF(s.tree()).render_octree(res, 4096, True)
    assert oc.TotalPruned() == ref.pruned
    # sharded and unpruned runs go through the same specialised kernels
    parts = [gpu.OctreeHIP(sdf, res, shard_rank=r, shard_count=3).RenderAll() for r in range(3)]
    assert (_sorted(np.concatenate(parts)).view(np.uint32) == tg.view(np.uint32)).all()
    if key != "npt_flange_resdiv400":
        assert gpu.OctreeHIP(sdf, res, prune=False).n_tris() == g["n_tris"]
    # the corner-sharing kernel (interpreter build) still works on a specialised handle
    assert (_sorted(gpu.OctreeHIP(sdf, res, share_corners=True).RenderAll()).view(np.uint32) == tg.view(np.uint32)).all()
    # distinct z rows through the specialised kernels (built on first use): same set, fewer evaluations, whole and in shards
    rows = gpu.OctreeHIP(sdf, res, share_corners=2)
    assert (_sorted(rows.RenderAll()).view(np.uint32) == tg.view(np.uint32)).all()
    assert rows.stats.evals <= oc.stats.evals and rows.stats.active_leaves == oc.stats.active_leaves
    if sdf.info()["kernels"].get("leaf_rows"):   # (a tree whose distinct-rows kernel does not build without scratch keeps every row)
        assert rows.stats.evals < oc.stats.evals and int(rows.stats.evals_leaf) == ref.evals_rows   # the oracle's count for the surviving bricks
    dense = gpu.OctreeHIP(sdf, res, share_corners=1)
    auto = gpu.OctreeHIP(sdf, res, share_corners=3)   # the library's choice between the two, by the tree: threads and knurls take the points
    assert (_sorted(auto.RenderAll()).view(np.uint32) == tg.view(np.uint32)).all()
    assert int(auto.stats.evals_leaf) == int((rows if scene == "npt-flange" else dense).stats.evals_leaf)
    if sdf.info()["leaf_k"] == 4:   # lane slots: passes of 256 + a tail of 64 / 128 / 256 in the specialised build, passes of 256 in the interpreter's
        assert int(dense.stats.evals_leaf) == (ref.evals_points_tails if sdf.info()["kernels"].get("leaf_dense") else ref.evals_points_256)
    parts = [gpu.OctreeHIP(sdf, res, shard_rank=r, shard_count=2, share_corners=2) for r in range(2)]
    assert (_sorted(np.concatenate([q.RenderAll() for q in parts])).view(np.uint32) == tg.view(np.uint32)).all()
    assert sum(int(q.stats.evals_leaf) for q in parts) == int(rows.stats.evals_leaf)


def test_specialised_full_size_npt_flange(gpu):
    """BASELINE.json configs[1] through the specialised kernels: count + digest of the sorted triangle set."""
    s = Builder().Scene("npt-flange")
    g = GOLD["npt_flange_resdiv1600"]
    res = np.uint32(g["res_bits"]).view(np.float32)
    sdf = gpu.SDF3HIP(s).specialize()
    oc = gpu.OctreeHIP(sdf, res)
    assert oc.n_tris() == g["n_tris"]
    assert hashlib.sha256(_sorted(oc.RenderAll()).tobytes()).hexdigest() == g["sha256_sorted"]
    rows = gpu.OctreeHIP(sdf, res, share_corners=2)   # distinct z rows of every brick once: the same set from fewer evaluations
    assert rows.n_tris() == g["n_tris"] and rows.stats.evals < oc.stats.evals

    def bag(t):   # order-independent fingerprint of a triangle multiset (a sort of 6.8 M triangles is 15 s; the default mesh above is sorted and hashed)
        w = np.ascontiguousarray(t, np.float32).reshape(-1, 9).view(np.uint32).astype(np.uint64)
        h = np.zeros(len(w), np.uint64)
        for c in range(9):
            h = (h * np.uint64(0x9E3779B97F4A7C15) + w[:, c]) ^ (h >> np.uint64(29))
        return int(h.sum(dtype=np.uint64)), int(np.bitwise_xor.reduce(h))
    assert bag(rows.RenderAll()) == bag(oc.RenderAll())


def test_specialise_is_idempotent_and_counts_evaluations(gpu):
    s = Builder().Scene("npt-flange")
    sdf = gpu.SDF3HIP(s)
    assert not sdf.info()["specialized"]
    rng = np.random.default_rng(3)
    bb = s.Bounds()
    pos = (bb[:3] + rng.random((5000, 3), np.float32) * (bb[3:] - bb[:3])).astype(np.float32)
    d0 = sdf.Evaluate(pos)
    sdf.specialize().specialize()
    info = sdf.info()
    assert info["specialized"] and info["specialize_s"] > 0
    assert _mismatch(sdf.Evaluate(pos), d0) == 0
    assert sdf.Evaluations() == 10000


def test_specialised_dualcontour_normals_image(gpu):
    """The second kernel group of a specialised handle (built on first use): dual contouring, central-difference
    normals and the 2-D image renderer give the interpreter's bits."""
    b = Builder()
    s = b.Scene("npt-flange")
    spec, plain = gpu.SDF3HIP(s).specialize(), gpu.SDF3HIP(s)
    res = np.float32(float(s.Diagonal()) / 60)
    for chis in (False, True):
        a = gpu.DualContourHIP(spec, res, chiseled=chis).RenderAll()
        c = gpu.DualContourHIP(plain, res, chiseled=chis).RenderAll()
        assert a.shape == c.shape and (_sorted(a).view(np.uint32) == _sorted(c).view(np.uint32)).all()
    ref = OracleSDF(s.tree()).render_dualcontour(res, False)
    assert (_sorted(gpu.DualContourHIP(spec, res).RenderAll()).view(np.uint32) == _sorted(ref.tris).view(np.uint32)).all()
    rng = np.random.default_rng(11)
    bb = s.Bounds()
    pos = (bb[:3] + rng.random((4000, 3), np.float32) * (bb[3:] - bb[:3])).astype(np.float32)
    assert _mismatch(spec.normals(pos, 1e-3), plain.normals(pos, 1e-3)) == 0
    sh2 = b.Union2D(b.NewCircle(1.0), b.Translate2D(b.NewRectangle(1.0, 2.0), 0.5, 0.25))
    s2, p2 = gpu.SDF2HIP(sh2).specialize(), gpu.SDF2HIP(sh2)
    ia, ib = s2.render_image(96, 64), p2.render_image(96, 64)
    assert all((np.asarray(x).view(np.uint8) == np.asarray(y).view(np.uint8)).all() for x, y in zip(ia, ib))


def test_background_specialisation(gpu, tmp_path, monkeypatch):
    """gsdf_hip_program_specialize_async: the handle meshes through the interpreter kernels while the build runs and through the
    specialised ones once it is adopted -- the same triangles, bit for bit, at every moment; a warm code-object cache makes the
    second handle's build a file read; destroying a handle waits for a build under way."""
    monkeypatch.setenv("GSDF_HIP_CACHE_DIR", str(tmp_path))
    s = Builder().Scene("bolt")
    g = GOLD["bolt_resdiv150"]
    res = np.uint32(g["res_bits"]).view(np.float32)
    sdf = gpu.SDF3HIP(s)
    assert not sdf.info()["specialized"]
    sdf.specialize_async()
    sdf.specialize_async()                                   # idempotent while a build is under way
    first = gpu.OctreeHIP(sdf, res)                          # at once: whichever kernels are ready (with a cold cache, the interpreter's)
    assert first.n_tris() == g["n_tris"] and hashlib.sha256(_sorted(first.RenderAll()).tobytes()).hexdigest() == g["sha256_sorted"]
    pos = corpus.sample_points(s)
    d0 = sdf.Evaluate(pos)
    assert sdf.specialize_poll(wait=True) and sdf.info()["specialized"]
    assert "specialised" in sdf.info()["kernels"]["leaf"]
    later = gpu.OctreeHIP(sdf, res)
    assert later.n_tris() == g["n_tris"] and hashlib.sha256(_sorted(later.RenderAll()).tobytes()).hexdigest() == g["sha256_sorted"]
    assert _mismatch(sdf.Evaluate(pos), d0) == 0 and _mismatch(d0, OracleSDF(s.tree()).Evaluate(pos)) == 0
    sdf.specialize_async()                                   # nothing left to do
    assert sdf.specialize_poll()
    # a second handle of the same tree: the cache holds the code object now
    import time
    t0 = time.perf_counter()
    sdf2 = gpu.SDF3HIP(s).specialize_async()
    assert sdf2.specialize_poll(wait=True)
    assert time.perf_counter() - t0 < 1.0 and sdf2.info()["kernels"].get("compiler") == "cache"
    # a handle closed while its build runs (a fresh tree, nothing cached): close() returns after the build, nothing leaks or crashes
    other = gpu.SDF3HIP(Builder().Scene("knurled-cylinder")).specialize_async()
    other.close()
    # poll on a handle that never asked for a build
    assert gpu.SDF3HIP(s).specialize_poll() is False


def test_rotated_screws_through_the_short_atan2_route(gpu):
    """A screw under a general rotation leaves no two points of a lane with equal x, y in its frame: every point takes math32.Atan2
    on its own, which the specialised kernels evaluate by the short float64 route (dm::atan2_fast, accepted only where its float32 is
    decided, the reference's sequence for the wave otherwise). Twelve seeded trees -- ISO, NPT (tapered) and buttress screws, rotated
    about random axes, translated, some unioned with a second screw -- on 150 000 points each, a third of them on the lattice of a
    renderer (shared coordinates, points on the screw's own axes): bit-identical to the oracle, octree meshes identical."""
    rng = np.random.default_rng(61)

    def make(seed):
        r = np.random.default_rng(seed)
        b = Builder()
        kind = seed % 3
        sc = (b.ScrewISO(1.0 + r.random(), 0.15 + 0.1 * r.random(), bool(seed & 1), 2.0) if kind == 0 else
              b.ScrewNPT(0.5, 1.5) if kind == 1 else b.ScrewPlasticButtress(1.2, 0.25, 2.0))
        axis = r.standard_normal(3)
        s = b.Translate(b.Rotate(sc, float(r.uniform(0.2, 2.9)), tuple(float(a) for a in axis / np.linalg.norm(axis))),
                        float(r.uniform(-1, 1)), float(r.uniform(-1, 1)), float(r.uniform(-1, 1)))
        if seed % 4 == 0:
            s = b.Union(s, b.Rotate(b.ScrewISO(0.8, 0.2, True, 1.5), 1.1, (1.0, 0.0, 0.0)))
        return b, s

    def check(seed):
        b, s = make(seed)
        sdf = gpu.SDF3HIP(s).specialize()
        assert sdf.info()["specialized"]
        bb = s.Bounds().astype(np.float64)
        c, h = (bb[:3] + bb[3:]) / 2, (bb[3:] - bb[:3]) / 2 * 1.1
        r = np.random.default_rng(seed + 1000)
        pos = (c + (r.random((150000, 3)) * 2 - 1) * h).astype(np.float32)
        res = np.float32(float(s.Diagonal()) / 257)
        lat = (np.floor(pos[:50000] / res) * res).astype(np.float32)            # lattice-like: shared x, y, z values
        lat[::7, 0] = 0.0
        lat[::11, 1] = -0.0
        pos[:50000] = lat
        ref = OracleSDF(s.tree())
        assert _mismatch(sdf.Evaluate(pos), ref.Evaluate(pos)) == 0, seed
        oc = gpu.OctreeHIP(sdf, np.float32(float(s.Diagonal()) / 90))
        want = ref.render_octree(np.float32(float(s.Diagonal()) / 90), 4096, True)
        assert oc.n_tris() == want.n_tris and (_sorted(oc.RenderAll()).view(np.uint32) == _sorted(want.tris).view(np.uint32)).all(), seed
    pmap(check, [int(x) for x in rng.integers(0, 10 ** 6, 12)], workers=12)
