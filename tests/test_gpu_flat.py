"""GPU parity of the device FlatRenderer (gsdf_hip_mesh_flat) with the oracle's restatement of
glrender/flatrenderer.go: same lattice, same evaluation count, bit-identical triangle set."""
import numpy as np
import pytest

from scaffold.builder import Builder
from oracle.oracle import OracleSDF

pytestmark = pytest.mark.gpu


def _sorted(t):
    t = np.ascontiguousarray(t, np.float32).reshape(-1, 9)
    return t[np.lexsort(t.view(np.uint32).T[::-1])]


def _same(a, b):
    a, b = _sorted(a), _sorted(b)
    return a.shape == b.shape and bool((a.view(np.uint32) == b.view(np.uint32)).all())


@pytest.mark.parametrize("scene,resdiv", [("npt-flange", 100), ("npt-flange", 233), ("bolt", 120), ("knurled-cylinder", 90)])
def test_flat_identical_to_oracle(gpu, scene, resdiv):
    b = Builder()
    s = b.Scene(scene)
    res = np.float32(float(s.Diagonal()) / resdiv)
    sdf = gpu.SDF3HIP(s)
    fl = gpu.FlatHIP(sdf, res)
    ref = OracleSDF(s.tree()).render_flat(res, 4096, 4)
    assert fl.Evaluations() == ref.evals                      # (nx+1)(ny+1)(nz+1): FlatRenderer.Evaluations
    assert fl.stats.leaf_cubes == ref.grid[0] * ref.grid[1] * ref.grid[2]
    assert fl.n_tris() == ref.n_tris
    assert _same(fl.RenderAll(), ref.tris)
    # kernels specialised for the tree: identical bits
    sdf.specialize()
    fs = gpu.FlatHIP(sdf, res)
    assert fs.n_tris() == ref.n_tris and _same(fs.RenderAll(), ref.tris)


def test_flat_readme_counts_resdiv400(gpu):
    b = Builder()
    s = b.Scene("npt-flange")
    res = np.float32(float(s.Diagonal()) / 400)
    fl = gpu.FlatHIP(gpu.SDF3HIP(s), res)
    assert fl.Evaluations() == 6711685                        # README.md:130 reports 6,711,686 (lattice + 1 probe evaluation)
    assert fl.n_tris() == 423852                              # README.md:116,130


def test_flat_sphere_41072(gpu):
    b = Builder()
    sdf = gpu.SDF3HIP(b.NewSphere(1.0))
    res = np.float32(1.0 / 33)
    fl = gpu.FlatHIP(sdf, res)
    ref = OracleSDF(b.NewSphere(1.0).tree()).render_flat(res, 4096, 1)
    assert fl.n_tris() == ref.n_tris and _same(fl.RenderAll(), ref.tris)


@pytest.mark.parametrize("nx", [3, 4, 31, 62, 63, 64, 65, 127, 128, 191])
def test_flat_row_lengths_around_the_bit_words(gpu, nx):
    """The marching pass reads the lattice as bit planes in words of 64 corners, rows running on (flat_cut_scan_kernel): row
    lengths of nx + 1 corners below, at and around multiples of 64 (a row further on = a whole number of words, or 63 bits
    more), and rows shorter than a word."""
    b = Builder()
    sdf = gpu.SDF3HIP(b.NewSphere(1.0))
    res = np.float32(2.02 / (nx - 0.5))
    fl = gpu.FlatHIP(sdf, res)
    ref = OracleSDF(b.NewSphere(1.0).tree()).render_flat(res, 4096, 1)
    assert ref.grid[0] == nx and fl.stats.leaf_cubes == ref.grid[0] * ref.grid[1] * ref.grid[2]
    assert fl.n_tris() == ref.n_tris and _same(fl.RenderAll(), ref.tris)
    for count in (2, 5):
        parts = [gpu.FlatHIP(sdf, res, shard_rank=r, shard_count=count).RenderAll().reshape(-1, 9) for r in range(count)]
        assert _same(np.concatenate(parts), ref.tris)


def test_flat_float_stream_pass_gives_the_same_triangles(gpu):
    """GSDF_HIP_FLAT_STREAM=1 selects the marching pass of rounds 1-2 (flat_march_kernel, which streams the float grid) in
    place of the bit-plane pass; read once per process, hence the subprocess."""
    import hashlib, os, subprocess, sys
    code = ("import hashlib, numpy as np\n"
            "from scaffold.builder import Builder\n"
            "from gsdf_amd import hip\n"
            "hip.init(0)\n"
            "s = Builder().Scene('npt-flange')\n"
            "t = hip.FlatHIP(hip.SDF3HIP(s), np.float32(float(s.Diagonal()) / 233)).RenderAll().reshape(-1, 9)\n"
            "t = t[np.lexsort(t.view(np.uint32).T[::-1])]\n"
            "print(len(t), hashlib.sha256(t.tobytes()).hexdigest())\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GSDF_HIP_FLAT_STREAM="1", PYTHONPATH=root)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    n, digest = out.stdout.split()[-2:]
    s = Builder().Scene("npt-flange")
    t = _sorted(gpu.FlatHIP(gpu.SDF3HIP(s), np.float32(float(s.Diagonal()) / 233)).RenderAll())
    assert int(n) == len(t) and digest == hashlib.sha256(t.tobytes()).hexdigest()


def test_flat_sharded_union_equals_whole(gpu):
    b = Builder()
    s = b.Scene("npt-flange")
    res = np.float32(float(s.Diagonal()) / 150)
    sdf = gpu.SDF3HIP(s)
    whole = gpu.FlatHIP(sdf, res)
    for count in (2, 3, 8):
        parts = [gpu.FlatHIP(sdf, res, shard_rank=r, shard_count=count) for r in range(count)]
        assert sum(p.n_tris() for p in parts) == whole.n_tris()
        assert sum(int(p.stats.leaf_cubes) for p in parts) == int(whole.stats.leaf_cubes)
        assert _same(np.concatenate([p.RenderAll().reshape(-1, 9) for p in parts]), whole.RenderAll())
    # more ranks than cube planes: the surplus ranks get nothing
    tiny = np.float32(float(s.Diagonal()) / 8)
    parts = [gpu.FlatHIP(sdf, tiny, shard_rank=r, shard_count=16) for r in range(16)]
    assert _same(np.concatenate([p.RenderAll().reshape(-1, 9) for p in parts]), gpu.FlatHIP(sdf, tiny).RenderAll())


def test_flat_argument_errors(gpu):
    b = Builder()
    sdf = gpu.SDF3HIP(b.NewSphere(1.0))
    with pytest.raises(gpu.HipError, match="invalid renderer cube resolution"):
        gpu.FlatHIP(sdf, np.float32(0))
    with pytest.raises(gpu.HipError, match="invalid renderer cube resolution"):
        gpu.FlatHIP(sdf, np.float32(-1))
    with pytest.raises(ValueError, match="eval buffer size must be at least 8"):
        gpu.FlatHIP(sdf, np.float32(0.1), evalBufferSize=4)
    with pytest.raises(ValueError, match="numParallel must be at least 1"):
        gpu.FlatHIP(sdf, np.float32(0.1), numParallel=0)
    with pytest.raises(gpu.HipError, match="too fine"):
        gpu.FlatHIP(sdf, np.float32(1e-5))
    # a lattice whose grid cannot fit the device (npt-flange at resdiv 12000: ~700 GB) is refused, not attempted
    fl = b.Scene("npt-flange")
    with pytest.raises(gpu.HipError, match="does not fit the device memory"):
        gpu.FlatHIP(gpu.SDF3HIP(fl), np.float32(float(fl.Diagonal()) / 12000))
    # one cube per axis is a legal lattice
    one = gpu.FlatHIP(sdf, np.float32(3.0))
    assert one.stats.leaf_cubes == 1 and one.Evaluations() == 8
    with pytest.raises(gpu.HipError, match="2D"):
        gpu.FlatHIP(gpu.SDF2HIP(b.NewCircle(1.0)), np.float32(0.1))


def test_flat_full_size_resdiv1600(gpu):
    """BASELINE config size: 420,224,000 lattice corners (SURVEY 8 header). The oracle needs minutes here, so parity
    rests on properties: specialised == interpreter kernels, union of z-slabs == whole, and the octree mesher's count
    (the two renderers sample shared corners at coordinates that may differ in the last bit -- x0+res vs o+(i+1)*res --
    so only the counts are comparable, and those only up to corners within an ulp of the surface)."""
    b = Builder()
    s = b.Scene("npt-flange")
    res = np.float32(float(s.Diagonal()) / 1600)
    sdf = gpu.SDF3HIP(s)
    fi = gpu.FlatHIP(sdf, res)
    assert fi.Evaluations() == 420224000
    ti = _sorted(fi.RenderAll())
    del fi
    sdf.specialize()
    fl = gpu.FlatHIP(sdf, res)
    assert fl.Evaluations() == 420224000
    tw = _sorted(fl.RenderAll())
    assert tw.shape == ti.shape and (tw.view(np.uint32) == ti.view(np.uint32)).all()
    parts = [gpu.FlatHIP(sdf, res, shard_rank=r, shard_count=3).RenderAll().reshape(-1, 9) for r in range(3)]
    assert _same(np.concatenate(parts), tw)
    oc = gpu.OctreeHIP(sdf, res)
    assert abs(oc.n_tris() - fl.n_tris()) <= 64
