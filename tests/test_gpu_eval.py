"""GPU parity: the HIP evaluator (through the C ABI) against the oracle and the committed golden
vectors. Bar: BIT-EXACT float32 distances (stricter than north_star's 1e-5 relative, which the tests
also state), because exact triangle counts need exact corner signs."""
import os

import numpy as np
import pytest

from par import pmap

import corpus
from scaffold.builder import Builder
from oracle.oracle import OracleSDF

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
REL_TOL = 1e-5  # north_star: "within 1e-5 relative"; eps for the denominator = 1e-3


def _mismatch(a, b):
    return int(((a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))).sum())


@pytest.mark.parametrize("which", ["3d", "2d"])
def test_corpus_bit_exact_vs_oracle(gpu, which):
    _, shapes = (corpus.shapes3d if which == "3d" else corpus.shapes2d)()
    for name, sh in shapes:
        pos = corpus.sample_points(sh)
        dg = gpu.SDFHIP(sh).Evaluate(pos)
        dc = OracleSDF(sh.tree()).Evaluate(pos)
        rel = np.abs(dg - dc) / np.maximum(np.abs(dc), 1e-3)
        assert np.nanmax(rel) <= REL_TOL, (name, np.nanmax(rel))
        assert _mismatch(dg, dc) == 0, (name, _mismatch(dg, dc), pos.shape[0])


def test_quadbezier_within_tolerance(gpu):
    # math32.Pow is float32-native with amd64 assembly Exp/Log upstream: restated via exp/log, so the
    # device (ocml) and the oracle (libm) may differ in the last ulp. Outside every benchmark config.
    _, shapes = corpus.bezier2d()
    for name, sh in shapes:
        pos = corpus.sample_points(sh)
        dg = gpu.SDFHIP(sh).Evaluate(pos)
        dc = OracleSDF(sh.tree()).Evaluate(pos)
        assert np.max(np.abs(dg - dc) / np.maximum(np.abs(dc), 1e-3)) <= REL_TOL


def test_golden_vectors(gpu):
    gold = np.load(os.path.join(GOLD, "corpus_distances.npz"))
    for fn in (corpus.shapes3d, corpus.shapes2d):
        _, shapes = fn()
        for name, sh in shapes:
            d = gpu.SDFHIP(sh).Evaluate(gold["pos_" + name])
            assert _mismatch(d, gold["dist_" + name]) == 0, name


def test_strides_ragged_sizes_and_counter(gpu):
    b = Builder()
    s = b.Scene("npt-flange")
    sdf = gpu.SDF3HIP(s)
    ref = OracleSDF(s.tree())
    rng = np.random.default_rng(5)
    bb = s.Bounds()
    total = 0
    for n in (1, 2, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4096, 32768, 100003):
        pos = (bb[:3] + rng.random((n, 3), np.float32) * (bb[3:] - bb[:3])).astype(np.float32)
        d12 = sdf.Evaluate(pos)
        pos16 = np.zeros((n, 4), np.float32)  # std140 vec3 / 16-byte stride (glbuild.go:195-197)
        pos16[:, :3] = pos
        pos16[:, 3] = 123.0
        d16 = sdf.Evaluate(pos16)
        assert _mismatch(d12, ref.Evaluate(pos)) == 0 and _mismatch(d12, d16) == 0, n
        total += 2 * n
    assert sdf.Evaluations() == total  # (*SDF3Compute).Evaluations, gpu.go:80


def test_error_behaviour_like_reference(gpu):
    b = Builder()
    sdf = gpu.SDF3HIP(b.NewSphere(1))
    with pytest.raises(gpu.HipError) as e:
        sdf.Evaluate(np.zeros((0, 3), np.float32))
    assert e.value.msg == "empty buffers"                                          # errEmptyBuffers
    with pytest.raises(gpu.HipError) as e:
        sdf.Evaluate(np.zeros((4, 3), np.float32), np.zeros(3, np.float32))
    assert e.value.msg == "position and distance buffer length mismatch"            # errMismatchBufferLength
    sdf2 = gpu.SDF2HIP(b.NewCircle(1))
    assert sdf2.is2d
    with pytest.raises(gpu.HipError):
        gpu.lib().gsdf_hip_eval3  # noqa: B018
        from gsdf_amd.hip import _check
        p = np.zeros((4, 3), np.float32)
        d = np.zeros(4, np.float32)
        _check(gpu.lib().gsdf_hip_eval3(sdf2._h, p.ctypes.data, 12, 4, d.ctypes.data, 4))  # 2D program, 3D call
    before = sdf.Evaluations()
    with pytest.raises(gpu.HipError):
        sdf.Evaluate(np.zeros((0, 3), np.float32))
    assert sdf.Evaluations() == before  # failed calls do not count


def test_positions_not_modified_and_dist_fully_overwritten(gpu):
    b = Builder()
    sdf = gpu.SDF3HIP(b.Scene("bolt"))
    pos = np.random.default_rng(2).standard_normal((5000, 3)).astype(np.float32) * 4
    keep = pos.copy()
    dist = np.full(5000, np.nan, np.float32)
    sdf.Evaluate(pos, dist)
    assert (pos == keep).all() and not np.isnan(dist).any()


def test_device_resident_eval(gpu):
    import torch
    b = Builder()
    s = b.Scene("knurled-cylinder")
    sdf = gpu.SDF3HIP(s)
    n = 200001
    tp = (torch.rand((n, 3), device="cuda") * 60 - 30).contiguous()
    td = torch.empty(n, device="cuda")
    sdf.evaluate_dev(tp.data_ptr(), 12, td.data_ptr(), n)
    torch.cuda.synchronize()
    dc = OracleSDF(s.tree()).Evaluate(tp.cpu().numpy())
    assert _mismatch(td.cpu().numpy(), dc) == 0


def test_normals_central_diff(gpu):
    b = Builder()
    s = b.Scene("npt-flange")
    pos = corpus.sample_points(s, n_grid=4, n_rand=500)
    ng = gpu.SDF3HIP(s).normals(pos, 1e-3)
    nc = OracleSDF(s.tree()).normals_central_diff(pos, 1e-3)
    assert _mismatch(ng.ravel(), nc.ravel()) == 0


def test_many_handles_and_big_union(gpu):
    b = Builder()
    rng = np.random.default_rng(9)
    parts = [b.Translate(b.NewSphere(0.2 + 0.1 * i / 32), *(rng.random(3) * 4 - 2)) for i in range(32)]
    u = b.Union(*parts)  # examples/test/glsdf3test.go: 32-sphere union CPU vs GPU
    pos = corpus.sample_points(u)
    assert _mismatch(gpu.SDF3HIP(u).Evaluate(pos), OracleSDF(u.tree()).Evaluate(pos)) == 0
    hs = [gpu.SDF3HIP(p) for p in parts[:8]]
    for h, p in zip(hs, parts):
        assert _mismatch(h.Evaluate(pos), OracleSDF(p.tree()).Evaluate(pos)) == 0


def test_exact_division_by_uniform_divisor_exhaustive(gpu):
    """dm::div_by_uniform (interp.h polygon loop) == IEEE division for EVERY float32 numerator the fast path
    accepts, for every polygon divisor |e|^2 of the benchmark scenes and a few adversarial divisors."""
    import ctypes as C
    from gsdf_amd._ctypes_common import OP
    b = Builder()
    divisors = set()
    for sc in ("npt-flange", "bolt", "knurled-cylinder"):
        t = b.Scene(sc).tree()
        for i in range(t.n_nodes):
            nd = t.nodes[i]
            if nd.op == OP["POLY2D"]:
                v = np.array([t.aux[nd.aux_off + k] for k in range(nd.aux_len)], np.float32).reshape(-1, 2)
                e = np.roll(v, 1, axis=0) - v
                divisors.update(np.float32(e[:, 0] * e[:, 0] + e[:, 1] * e[:, 1]).tolist())
            if nd.op == OP["SCREW"]:
                divisors.add(float(nd.p[0]))                      # sawTooth: x / pitch
            if nd.op in (OP["SMOOTH_UNION"], OP["SMOOTH_DIFF"], OP["SMOOTH_INTERSECT"]):
                divisors.add(float(nd.p[0]))                      # (0.5*(b -+ a)) / k
    divisors.update([1.0, 3.0, 0.1, 7.0, 1.9999999, 1.0000001, 6.2831855, 1e-9, 1e9, float(np.float32(2.0) ** -30)])
    divisors.add(float(np.float32(6.2831853071795862)))          # screw: lead*theta / 2pi
    assert len(divisors) > 20
    checked = 0
    for d in sorted(divisors):
        bad, nfast, r = C.c_uint64(), C.c_uint64(), C.c_float()
        rc = gpu.lib().gsdf_hip_selftest_div(np.float32(d), C.byref(bad), C.byref(nfast), C.byref(r))
        assert rc == 0
        if r.value == 0.0:
            continue  # not eligible: interpreter uses the IEEE expansion
        assert nfast.value > 2_000_000_000 and bad.value == 0, (d, bad.value)
        checked += 1
    assert checked >= 20


def test_image_renderer_sdf2(gpu):
    """glrender.ImageRendererSDF2 (image.go:76-118): pixel lattice, row 0 on top, default black/white conversion."""
    _, shapes = corpus.shapes2d()
    for name, sh in shapes[:12]:
        for (w, h) in ((64, 48), (257, 131)):
            dg, cg = gpu.SDF2HIP(sh).render_image(w, h)
            dc, cc = OracleSDF(sh.tree()).render_image(w, h)
            assert _mismatch(dg.ravel(), dc.ravel()) == 0, (name, w, h)
            assert (cg == cc).all()
    b = Builder()
    with pytest.raises(gpu.HipError):
        gpu.SDF3HIP(b.NewSphere(1)).render_image(8, 8)  # 3D program


def test_large_batch_and_special_values(gpu):
    b = Builder()
    s = b.Scene("npt-flange")
    sdf = gpu.SDF3HIP(s)
    n = (1 << 22) + 3
    rng = np.random.default_rng(11)
    pos = (rng.standard_normal((n, 3)) * 20).astype(np.float32)
    pos[:64] = 0.0                                   # the axis: atan2(0,0), hypot(0,0)
    pos[64:128, :2] = 0.0                            # on the screw axis at various z
    pos[128:192, 1] = 0.0                            # y == 0: atan2(+-0, x)
    pos[192:256, 1] = -0.0
    pos[256:320, 0] = 0.0
    d = sdf.Evaluate(pos)
    ref = OracleSDF(s.tree())
    idx = np.concatenate([np.arange(4096), rng.integers(0, n, 200000)])
    assert _mismatch(d[idx], ref.Evaluate(pos[idx])) == 0
    assert np.isfinite(d).all()


def test_circular_array_sector_index_without_the_angle(gpu):
    """dm::circ_sector_fast decides floor(atan2(y, x) / angle) from a float32 estimate only where that is certain: 2^32 points
    per sector count (all magnitudes, signed zeros, points within 1e-9 rad of the boundaries), no mismatch allowed."""
    import ctypes as C
    for ncirc in (24, 3, 7, 12, 100, 1000):
        bad, nfast = C.c_uint64(1), C.c_uint64(0)
        assert gpu.lib().gsdf_hip_selftest_circ(np.float32(ncirc), C.byref(bad), C.byref(nfast)) == 0
        assert bad.value == 0 and nfast.value > 1_500_000_000, (ncirc, bad.value, nfast.value)


def test_atan2_short_route_rounds_like_the_reference(gpu):
    """dm::atan2_fast -- float32(math.Atan2) from a 16-FMA float64 polynomial instead of Go's Cephes sequence, accepted only where
    every float64 within 2^-40 of its result rounds to the same float32 -- against the reference's sequence (dm::atan2_ref): 2^32
    hashed pairs of every sign and magnitude, 2^30 pairs steered towards float32 rounding boundaries, 2^30 lattice-shaped pairs
    (equal magnitudes, axes, signed zeros, the old routine's range boundaries). No accepted result may differ; nearly all are accepted."""
    import ctypes as C
    for mode, log2n in ((0, 32), (1, 30), (2, 30)):
        bad, nfast = C.c_uint64(1), C.c_uint64(0)
        assert gpu.lib().gsdf_hip_selftest_atan2(mode, log2n, C.byref(bad), C.byref(nfast)) == 0
        assert bad.value == 0, (mode, bad.value, nfast.value)
        assert nfast.value > 0.95 * 2.0 ** log2n, (mode, nfast.value / 2.0 ** log2n)   # (zeros are one pair in 32: they take the reference's route)


def test_cossin_short_route_rounds_like_the_reference_for_every_float(gpu):
    """dm::cossin_fast -- float32(math.Cos), float32(math.Sin) by FMAs on a 32-bit octant, accepted only where every float64 within 2^-44
    of its results rounds to one float32 -- against the reference's sequence (dm::cossinf_) for EVERY float32 argument (2^32 bit patterns:
    subnormals, both zeros, |x| up to the route's 2^20 limit; beyond it, infinities and NaN it declines). No accepted result may differ."""
    import ctypes as C
    bad, nfast = C.c_uint64(1), C.c_uint64(0)
    assert gpu.lib().gsdf_hip_selftest_cossin(C.byref(bad), C.byref(nfast)) == 0
    assert bad.value == 0, (bad.value, nfast.value)
    # |x| < 2^20: exponents 0 .. 146 of 256, both signs: 2 * 147 * 2^23 arguments less the rejected one in ~2^19
    assert 0.999 * 2 * 147 * 2 ** 23 < nfast.value <= 2 * 147 * 2 ** 23, nfast.value


def test_circular_array_points_on_sector_boundaries(gpu):
    """The sector index comes from a float32 angle estimate unless some point of the wave is too close to a sector boundary:
    points ON the boundaries (angle k * 2 pi / n to the last bit, the axes, +-0, the origin) mixed with ordinary ones, so that
    waves take both paths and the hand-over is exact; interpreter and specialised kernels, 3-D and 2-D arrays."""
    b = Builder()
    rng = np.random.default_rng(5)
    cases = []
    for div in (3, 4, 7, 24, 45):
        n = max(1, div - 1)
        sh3 = b.CircularArray(b.Translate(b.NewBox(0.6, 0.4, 0.5, 0.05), 1.5, 0, 0), n, div)
        sh2 = b.CircularArray2D(b.Translate2D(b.NewRectangle(0.6, 0.4), 1.5, 0), n, div)
        k = rng.integers(-2 * div, 2 * div, 40000)
        th = (k * (2 * np.pi / div)).astype(np.float64)
        th[::3] += rng.standard_normal(len(th[::3])) * 10.0 ** rng.uniform(-9, -1, len(th[::3]))   # a third near, not on
        rad = 10.0 ** rng.uniform(-1.5, 1.0, len(th))
        xy = np.stack([rad * np.cos(th), rad * np.sin(th)], 1).astype(np.float32)
        xy[:64] = 0.0
        xy[64:128, 1] = 0.0
        xy[128:192, 1] = -0.0
        xy[192:256, 0] = 0.0
        xy[256:320, 0] = -0.0
        far = (rng.standard_normal((20000, 2)) * 3).astype(np.float32)                                 # waves that never leave the fast path
        xy = np.concatenate([xy, far])
        z = (rng.standard_normal(len(xy)) * 0.4).astype(np.float32)
        for shape, pos in ((sh3, np.concatenate([xy, z[:, None]], 1)), (sh2, xy)):
            cases.append((div, shape, np.ascontiguousarray(pos, np.float32)))

    def check(case):                                          # (ten builds: side by side, tests/par.py)
        div, shape, pos = case
        ref = OracleSDF(shape.tree()).Evaluate(pos)
        sdf = gpu.SDFHIP(shape)
        assert _mismatch(sdf.Evaluate(pos), ref) == 0, (div, pos.shape)
        sdf.specialize()
        assert _mismatch(sdf.Evaluate(pos), ref) == 0, (div, pos.shape, "specialised")
    pmap(check, cases, workers=10)


def test_sqrt_unit_range_exhaustive(gpu):
    import ctypes as C
    bad = C.c_uint64(1)
    assert gpu.lib().gsdf_hip_selftest_sqrt(C.byref(bad)) == 0
    assert bad.value == 0


def test_block_cached_sdf3_like_reference(gpu):
    """gleval.BlockCachedSDF3 (gleval.go:110-218) in front of the HIP evaluator: distances, CacheHits and Evaluations
    equal the restated reference wrapper over the oracle, including the lossy-cache behaviour (a cell answers with the
    distance of the LAST position evaluated in it) and Reset clearing the statistics."""
    from oracle.oracle import OracleBlockCachedSDF3
    b = Builder()
    s = b.Scene("npt-flange")
    sdf = gpu.SDF3HIP(s)
    ref = OracleSDF(s.tree())
    rx, ry, rz = 0.5, 0.75, 0.25
    cg = gpu.BlockCachedSDF3HIP(sdf, rx, ry, rz)
    co = OracleBlockCachedSDF3(ref, rx, ry, rz)
    rng = np.random.default_rng(17)
    bb = s.Bounds()
    for rnd in range(4):
        n = 3000
        pos = (bb[:3] + rng.random((n, 3), np.float32) * (bb[3:] - bb[:3]) * np.float32(0.3)).astype(np.float32)
        pos[::7] = pos[1::7][: pos[::7].shape[0]]           # exact repeats inside one batch: both are misses, both evaluated
        if rnd == 2:
            pos[:50] -= np.float32(5.0)                       # outside the bounds: negative cell indices (truncation toward zero)
        dg, do = cg.Evaluate(pos), co.Evaluate(pos)
        assert _mismatch(dg, do) == 0, rnd
        assert cg.CacheHits() == co.hits and cg.Evaluations() == co.evals, rnd
    assert cg.CacheHits() > 0
    cg.Reset(sdf, 1.0, 1.0, 1.0)
    assert cg.CacheHits() == 0 and cg.Evaluations() == 0
    assert np.array_equal(cg.Bounds(), sdf.Bounds())
    with pytest.raises(gpu.HipError) as e:
        cg.Reset(sdf, 0.0, 1.0, 1.0)
    assert e.value.msg == "invalid resolution for BlockCachedSDF3"
    with pytest.raises(gpu.HipError) as e:
        cg.Evaluate(np.zeros((0, 3), np.float32))
    assert e.value.msg == "empty buffers"


def test_host_buffer_paths_agree_across_the_small_call_threshold(gpu):
    """gsdf_hip_eval3 serves calls of up to 1 MiB of positions / 2^18 points from pinned, device-mapped host memory (the
    kernel reads across PCIe) and larger ones through DMA staging: both sides of each limit, both strides, interleaved
    on one handle, must give the oracle's bits."""
    b = Builder()
    s = b.Scene("bolt")
    sdf = gpu.SDF3HIP(s)
    ref = OracleSDF(s.tree())
    rng = np.random.default_rng(5)
    bb = s.Bounds()
    for n, width in ((87381, 3), (87382, 3), (65536, 4), (65537, 4), (1, 3), (262144, 3), (40000, 4), (262145, 3), (5, 4)):
        pos = np.zeros((n, width), np.float32)
        pos[:, :3] = bb[:3] + rng.random((n, 3), np.float32) * (bb[3:] - bb[:3])
        keep = pos.copy()
        d = sdf.Evaluate(pos)
        assert d.shape == (n,) and (pos == keep).all()
        step = max(1, n // 3000)   # the oracle on a sample, plus the first and last points
        idx = np.unique(np.concatenate([np.arange(0, n, step), [0, n - 1]]))
        dr = ref.Evaluate(np.ascontiguousarray(pos[idx, :3]))
        assert (d[idx].view(np.uint32) == dr.view(np.uint32)).all(), (n, width)


def test_registered_buffers_pipelined_and_concurrent_host_calls(gpu):
    """The host-buffer drop-in beyond one blocking call: caller buffers the GPU reaches directly (gsdf_hip_host_alloc: no
    staging copy), the submit / wait pair with several batches in flight, and concurrent callers on one program -- what
    glrender.FlatRenderer's goroutines do to an evaluator (flatrenderer.go:120-129). Same bits on every path; the
    evaluation counter adds up."""
    import threading
    b = Builder()
    s = b.Scene("npt-flange")
    ref = OracleSDF(s.tree())
    bb = s.Bounds()
    rng = np.random.default_rng(9)
    for spec in (False, True):
        sdf = gpu.SDF3HIP(s)
        if spec:
            sdf.specialize()
        total = 0
        for n in (1, 257, 4096, 32768, 300001):                        # the last one is beyond the staging slots' size
            src = (bb[:3] + rng.random((n, 3), np.float32) * (bb[3:] - bb[:3])).astype(np.float32)
            want = ref.Evaluate(src)
            ppos, pdist = gpu.host_array((n, 3)), gpu.host_array((n,))
            ppos[:] = src
            pdist[:] = np.nan
            sdf.Evaluate(ppos, pdist)                                  # zero copy: the kernel reads / writes these arrays
            assert _mismatch(np.asarray(pdist), want) == 0 and (np.asarray(ppos) == src).all(), n
            total += n
            mixed = np.full(n, np.nan, np.float32)                     # registered positions, pageable distances: staged
            sdf.Evaluate(ppos, mixed)
            assert _mismatch(mixed, want) == 0
            total += n
        # pipelined: four batches in flight, waited in another order than submitted
        n = 20000
        batches = []
        for k in range(4):
            src = (bb[:3] + rng.random((n, 3), np.float32) * (bb[3:] - bb[:3])).astype(np.float32)
            batches.append((src, np.full(n, np.nan, np.float32), ref.Evaluate(src)))
        tickets = [sdf.submit(p_, d_) for p_, d_, _ in batches]
        assert sorted(t & 0xff for t in tickets) == [0, 1, 2, 3]       # a ticket = staging slot | generation << 8
        for k in (2, 0, 3, 1):
            sdf.wait(tickets[k])
            assert _mismatch(batches[k][1], batches[k][2]) == 0
        total += 4 * n
        with pytest.raises(gpu.HipError):
            sdf.wait(tickets[0])                                       # already waited for
        # a stale ticket must not release (or copy into) the call that has since taken its slot
        d_new = np.full(n, np.nan, np.float32)
        t_new = sdf.submit(batches[0][0], d_new)
        stale = [t for t in tickets if (t & 0xff) == (t_new & 0xff)][0]
        assert stale != t_new
        with pytest.raises(gpu.HipError):
            sdf.wait(stale)
        sdf.wait(t_new)
        assert _mismatch(d_new, batches[0][2]) == 0
        total += n
        with pytest.raises(gpu.HipError):
            sdf.submit(np.zeros((1 << 19, 3), np.float32), np.zeros(1 << 19, np.float32))   # too large for a staging slot
        # concurrent callers
        errs = []

        done = []

        def worker2(seed):
            r = np.random.default_rng(seed)
            tot = 0
            myref = OracleSDF(s.tree())                                # the oracle's evaluator is not shared between threads
            for _ in range(25):
                m = int(r.integers(1, 40000))
                p_ = (bb[:3] + r.random((m, 3), np.float32) * (bb[3:] - bb[:3])).astype(np.float32)
                if _mismatch(sdf.Evaluate(p_), myref.Evaluate(p_)):
                    errs.append(seed)
                tot += m
            done.append(tot)

        th = [threading.Thread(target=worker2, args=(100 + k,)) for k in range(6)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs and len(done) == 6
        total += sum(done)
        assert sdf.Evaluations() == total
    with pytest.raises(ValueError):
        gpu.SDF3HIP(b.NewSphere(1)).Evaluate(np.zeros((4, 3), np.float32), np.zeros(4, np.float64))   # wrong dist dtype: refused

