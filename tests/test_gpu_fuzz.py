"""Randomised compositions (tests/fuzz_trees.py): distances and octree meshes of seeded random CSG trees are
bit-identical between the HIP backend (interpreter kernels, run-time specialised kernels) and the oracle."""
import numpy as np
import pytest

import fuzz_trees
from oracle.oracle import OracleSDF

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


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_trees_distances_and_meshes(gpu, seed):
    _, shapes = fuzz_trees.random_shapes(seed, 14, depth=4)
    assert len(shapes) == 14
    rng = np.random.default_rng(100 + seed)
    meshed = 0
    for k, sh in enumerate(shapes):
        ref = OracleSDF(sh.tree())
        sdf = gpu.SDF3HIP(sh)
        pos = _points(sh, rng)
        dref = ref.Evaluate(pos)
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
            assert (_sorted(oc.RenderAll()).view(np.uint32) == _sorted(m.tris).view(np.uint32)).all(), (seed, k)
            meshed += 1
    assert meshed >= 8
