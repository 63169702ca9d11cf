"""Randomised compositions (tests/fuzz_trees.py): distances and octree meshes of seeded random CSG trees are
bit-identical between the HIP backend (interpreter kernels, run-time specialised kernels) and the oracle."""
import numpy as np
import pytest

import fuzz_trees
from oracle.oracle import OracleSDF
from par import pmap

pytestmark = pytest.mark.gpu


def _mismatch(a, b):
    return int(((a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))).sum())


def _sorted(t):
    t = np.ascontiguousarray(t, np.float32).reshape(-1, 9)
    return t[np.lexsort(t.view(np.uint32).T[::-1])]


def _points(sh, rng, n=6000):
    bb = sh.Bounds().astype(np.float32)
    c, h = (bb[:3] + bb[3:]) / 2, (bb[3:] - bb[:3]) / 2 * np.float32(1.1)
    p = (c + (rng.random((n, 3), np.float32) * 2 - 1) * h).astype(np.float32)
    # lattice-like points too: exact coordinate repeats, zeros, axis planes
    g = np.float32(0.125) * rng.integers(-12, 13, (n // 4, 3)).astype(np.float32)
    return np.concatenate([p, g]).astype(np.float32)


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_trees_distances_and_meshes(gpu, seed):
    _, shapes = fuzz_trees.random_shapes(seed, 14, depth=4)
    assert len(shapes) == 14
    rng = np.random.default_rng(100 + seed)
    points = [_points(sh, rng) for sh in shapes]              # (drawn in tree order, whatever order the trees are checked in)

    def check(k):                                             # one tree: its builds are compiler runs of seconds -- side by side (tests/par.py)
        sh, pos, meshed = shapes[k], points[k], 0
        ref = OracleSDF(sh.tree())
        sdf = gpu.SDF3HIP(sh)
        dref = ref.Evaluate(pos)
        assert not np.isnan(dref).any(), (seed, k)   # trees the constructors accept, finite positions: no NaN, so _mismatch masks nothing (the NaN relation: tests/test_gpu_nan.py)
        assert _mismatch(sdf.Evaluate(pos), dref) == 0, (seed, k, "interpreter")
        if k % 3 == 0:
            sdf.specialize()
            assert _mismatch(sdf.Evaluate(pos), dref) == 0, (seed, k, "specialised")
        # octree mesh at a coarse resolution: the leaf kernels' corner-pair sharing and slot reuse in action
        res = np.float32(float(sh.Diagonal()) / 48)
        oc = gpu.OctreeHIP(sdf, res)
        m = ref.render_octree(res, 4096, True)
        assert oc.n_tris() == m.n_tris, (seed, k)
        if m.n_tris:
            want = _sorted(m.tris).view(np.uint32)
            assert (_sorted(oc.RenderAll()).view(np.uint32) == want).all(), (seed, k)
            meshed += 1
            # the evaluations the reference repeats left out: distinct lattice points (1) / distinct z rows (2) of a brick once each
            for sc in ((1, 2) if k % 2 else ()):   # (every other tree: each option is a mesh and a sort of its own)
                sh_ = gpu.OctreeHIP(sdf, res, share_corners=sc)
                assert sh_.n_tris() == m.n_tris and sh_.stats.evals <= oc.stats.evals, (seed, k, sc)
                assert (_sorted(sh_.RenderAll()).view(np.uint32) == want).all(), (seed, k, sc)
        # flat renderer: the PAIRED lattice pass (two planes per lane) against the oracle's FlatRenderer
        if k % 2 == 0:
            fl = gpu.FlatHIP(sdf, res)
            mf = ref.render_flat(res, 4096, 2)
            assert fl.Evaluations() == mf.evals and fl.n_tris() == mf.n_tris, (seed, k, "flat")
            if mf.n_tris:
                assert (_sorted(fl.RenderAll()).view(np.uint32) == _sorted(mf.tris).view(np.uint32)).all(), (seed, k, "flat")
        return meshed
    meshed = sum(pmap(check, range(len(shapes)), workers=7))
    assert meshed >= 8


def test_random_2d_trees(gpu):
    _, shapes = fuzz_trees.random_shapes2d(7, 24, depth=3)
    assert len(shapes) == 24
    rng = np.random.default_rng(70)
    points = []
    for sh in shapes:
        bb = np.asarray(sh.Bounds(), np.float32)
        lo, hi = bb[[0, 1]], bb[[3, 4]]
        c, h = (lo + hi) / 2, (hi - lo) / 2 * np.float32(1.2)
        pos = (c + (rng.random((5000, 2), np.float32) * 2 - 1) * h).astype(np.float32)
        points.append(np.concatenate([pos, np.float32(0.125) * rng.integers(-12, 13, (1000, 2)).astype(np.float32)]))

    def check(k):
        sh, pos = shapes[k], points[k]
        dref = OracleSDF(sh.tree()).Evaluate(pos)
        sdf = gpu.SDF2HIP(sh)
        assert _mismatch(sdf.Evaluate(pos), dref) == 0, (k, "interpreter")
        if k % 4 == 0:
            assert _mismatch(sdf.specialize().Evaluate(pos), dref) == 0, (k, "specialised")
    pmap(check, range(len(shapes)), workers=6)


def test_random_trees_dual_contouring_and_normals(gpu):
    """Dual contouring (all five stages incl. the fp64 QR) and central-difference normals of random trees."""
    _, shapes = fuzz_trees.random_shapes(11, 8, depth=3)
    rng = np.random.default_rng(110)
    points = [_points(sh, rng, 2000) for sh in shapes]

    def check(k):
        sh = shapes[k]
        ref = OracleSDF(sh.tree())
        sdf = gpu.SDF3HIP(sh)
        if k % 2 == 0:
            sdf.specialize()
        res = np.float32(float(sh.Diagonal()) / 40)
        try:
            m = ref.render_dualcontour(res, False)
        except Exception:
            return 0                                  # lattice too large / degenerate for the oracle's renderer
        done = 0
        a = gpu.DualContourHIP(sdf, res).RenderAll()
        assert a.shape[0] == m.n_tris, (k, a.shape[0], m.n_tris)
        if m.n_tris:
            assert (_sorted(a).view(np.uint32) == _sorted(m.tris).view(np.uint32)).all(), k
            done = 1
        pos = points[k]
        assert _mismatch(sdf.normals(pos, 1e-3), ref.normals_central_diff(pos, 1e-3)) == 0, k
        return done
    done = sum(pmap(check, range(len(shapes)), workers=4))
    assert done >= 4


def test_ragged_last_tile_of_a_large_specialised_2d_program(gpu):
    """Regression (2-D fuzz tree 708/1): under the 4-workgroup register budget its specialised eval kernel needs scratch, and
    that build wrote the results of the points sharing a lane with the padding of a ragged last tile to wrong addresses.
    Kernels that need scratch are no longer used; whatever kernel the handle ends up with must give the oracle's bits
    for ragged and full batch sizes alike."""
    _, shapes = fuzz_trees.random_shapes2d(708, 6, depth=3)
    sh = shapes[1]
    ref = OracleSDF(sh.tree())
    rng = np.random.default_rng(1)
    bb = np.asarray(sh.Bounds(), np.float32)
    lo, hi = bb[[0, 1]], bb[[3, 4]]
    allpos = ((lo + hi) / 2 + (rng.random((5000, 2), np.float32) * 2 - 1) * (hi - lo) * np.float32(0.6)).astype(np.float32)
    want = ref.Evaluate(allpos)
    for spec in (False, True):
        sdf = gpu.SDF2HIP(sh)
        if spec:
            sdf.specialize()
        for n in (3000, 4097, 1024, 216, 257, 1, 5000):
            got = sdf.Evaluate(allpos[:n].copy())
            assert _mismatch(got, want[:n]) == 0, (spec, n)
