"""GPU: the octree's centre tests in interval mode (dev_ops.h: D_LIP_*, interp.h: LIP) against the oracle's
(orc_eval3_bounds) on fields that are not distance fields -- twists, high-lead and tapered screws, buttress threads, knurls,
non-rigid transforms, under scales, shells and every boolean. Cube for cube the same decisions (TotalPruned), the same
triangle set, and the surface of the unpruned octree, through the interpreter and the specialised kernels."""
import hashlib

import numpy as np
import pytest

from scaffold.builder import Builder
from nonlip_trees import nonlip_shapes
from oracle.oracle import OracleSDF

pytestmark = pytest.mark.gpu


def _sorted(t):
    t = np.ascontiguousarray(t, np.float32).reshape(-1, 9)
    return t[np.lexsort(t.view(np.uint32).T[::-1])]


@pytest.mark.parametrize("seed,spec", [(21, False), (22, True), (23, False)])
def test_non_lipschitz_trees_prune_like_the_oracle(gpu, seed, spec):
    _, shapes = nonlip_shapes(seed, 12)
    lost = 0
    for i, sh in enumerate(shapes):
        res = np.float32(float(sh.Diagonal()) / 70)
        ref = OracleSDF(sh.tree()).render_octree(res, 4096, True)
        sdf = gpu.SDF3HIP(sh)
        if spec:
            sdf.specialize()
        oc = gpu.OctreeHIP(sdf, res)
        assert oc.n_tris() == ref.n_tris and oc.TotalPruned() == ref.pruned, (seed, i)
        assert (_sorted(oc.RenderAll()).view(np.uint32) == _sorted(ref.tris).view(np.uint32)).all(), (seed, i)
        full = gpu.OctreeHIP(sdf, res, prune=False).n_tris()
        assert oc.n_tris() == full, (seed, i)                                    # no cube that holds surface was dropped
        plain = gpu.OctreeHIP(sdf, res, assume_sdf=True)                         # the reference's predicate at every level
        assert plain.TotalPruned() == OracleSDF(sh.tree()).render_octree(res, 4096, True, assume_sdf=True).pruned
        lost += plain.n_tris() != full
    assert lost >= 1   # ... which this family defeats


def test_buttress_cap_known_counts(gpu):
    """tests/test_prune_bounds.py's small showerhead cap on the device: 54,752 triangles by default and from the flat
    renderer, 54,293 with the reference's predicate applied to every level."""
    b = Builder()
    cap = b.Difference(b.KnurledHead(1.5, 1.0, 0.2), b.ScrewPlasticButtress(2.2, 0.3, 1.2))
    res = np.float32(float(cap.Diagonal()) / 120)
    for spec in (False, True):
        sdf = gpu.SDF3HIP(cap)
        if spec:
            sdf.specialize()
        assert gpu.OctreeHIP(sdf, res).n_tris() == 54752 == gpu.FlatHIP(sdf, res).n_tris()
        assert gpu.OctreeHIP(sdf, res, assume_sdf=True).n_tris() == 54293


def test_negative_scale_factors_prune_like_the_oracle(gpu):
    """Scale by a negative factor around a twist, a screw, a non-rigid transform (tests/tree_edit.py): the device's interval
    radius scales by the magnitude like the oracle's -- same decisions, same triangles, nothing lost against the unpruned octree."""
    from tree_edit import negative_scale_trees
    for k, (name, t) in enumerate(negative_scale_trees()):
        bb = np.array(t.bb[:], np.float32)
        res = np.float32(float(np.linalg.norm(bb[3:] - bb[:3])) / 70)
        ref = OracleSDF(t).render_octree(res, 4096, True)
        sdf = gpu.SDFHIP(t)
        if k % 2:
            sdf.specialize()
        oc = gpu.OctreeHIP(sdf, res)
        assert oc.n_tris() == ref.n_tris and oc.TotalPruned() == ref.pruned, name
        assert (_sorted(oc.RenderAll()).view(np.uint32) == _sorted(ref.tris).view(np.uint32)).all(), name
        assert gpu.OctreeHIP(sdf, res, prune=False).n_tris() == oc.n_tris(), name
