"""Multi-GPU exchange through the C ABI on RCCL (gsdf_hip_comm_* / gsdf_hip_mesh_gatherv, include/gsdf_hip.h), exercised
at world size 1 -- what one GPU allows: the communicator, the count all-gather, the grouped broadcast of the payload and
the reduction all run through librccl, so the `nccl` path executes at least once per round. The rank-major ordering of a
real multi-rank gather is covered on CPU by tests/test_gather_gloo.py (same layout, gloo)."""
import numpy as np
import pytest

from scaffold.builder import Builder

pytestmark = pytest.mark.gpu


def test_rccl_gatherv_world_size_one(gpu):
    b = Builder()
    comm = gpu.CommHIP(gpu.CommHIP.unique_id(), 0, 1)
    assert comm.allreduce_sum([5, 7, 2**40 + 3]) == [5, 7, 2**40 + 3]
    for sh, res in ((b.NewSphere(1.0), 1.0 / 16), (b.Scene("npt-flange"), None)):
        res = np.float32(res if res else float(sh.Diagonal()) / 200)
        sdf = gpu.SDF3HIP(sh)
        for mesh in (gpu.OctreeHIP(sdf, res), gpu.FlatHIP(sdf, res), gpu.DualContourHIP(sdf, np.float32(4 * res))):
            g = mesh.gatherv(comm)
            assert g.counts == [mesh.n_tris()] and g.n_tris() == mesh.n_tris() > 0
            assert g.dev_ptr() != mesh.dev_ptr()                       # its own device buffer
            assert (g.RenderAll().view(np.uint32) == mesh.RenderAll().view(np.uint32)).all()   # same triangles, same order
            assert g.WriteBinarySTL() == mesh.WriteBinarySTL()
    # shard union through the gather: with one rank per shard the gathered mesh is the concatenation; here the two
    # shards of a 2-way split are gathered one after the other on the same communicator
    sdf = gpu.SDF3HIP(b.Scene("bolt"))
    res = np.float32(float(b.Scene("bolt").Diagonal()) / 150)
    whole = gpu.OctreeHIP(sdf, res)
    parts = [gpu.OctreeHIP(sdf, res, shard_rank=r, shard_count=2).gatherv(comm) for r in range(2)]
    assert sum(p.n_tris() for p in parts) == whole.n_tris()
    # an empty contribution is legal (a rank may own no surface bricks)
    small = gpu.SDF3HIP(b.NewSphere(1.0))
    shards = [gpu.OctreeHIP(small, np.float32(0.25), shard_rank=r, shard_count=64) for r in range(64)]
    empties = [m for m in shards if m.n_tris() == 0]
    assert empties and sum(m.n_tris() for m in shards) == gpu.OctreeHIP(small, np.float32(0.25)).n_tris()
    g = empties[0].gatherv(comm)
    assert g.n_tris() == 0 and g.counts == [0] and g.RenderAll().shape == (0, 3, 3)
    comm.close()
    with pytest.raises(ValueError):
        gpu.CommHIP(b"short", 0, 1)
    with pytest.raises(gpu.HipError):
        gpu.CommHIP(gpu.CommHIP.unique_id(), 3, 2)                     # rank out of range


def test_gather_modes_and_pipelining_world_size_one(gpu):
    """gsdf_hip_mesh_gatherv_start / _wait: ALL, ROOT (rank 0 copies its own share, nobody else sends) and NONE (counts only) at
    world size 1, and two gathers in flight while the next mesh is made -- the loop bench.py --gpus N runs."""
    b = Builder()
    comm = gpu.CommHIP(gpu.CommHIP.unique_id(), 0, 1)
    sh = b.Scene("npt-flange")
    sdf = gpu.SDF3HIP(sh)
    sdf.specialize()
    res = np.float32(float(sh.Diagonal()) / 300)
    want = gpu.OctreeHIP(sdf, res)
    ref = want.RenderAll().copy()
    srt = lambda t: t.reshape(-1, 9)[np.lexsort(t.reshape(-1, 9).view(np.uint32).T[::-1])]
    for mode in (gpu.GATHER_ALL, gpu.GATHER_ROOT):
        g, counts, gs = want.gatherv_start(comm, mode, 0).wait()
        assert counts == [want.n_tris()] and g.n_tris() == want.n_tris()
        assert (g.RenderAll().view(np.uint32) == ref.view(np.uint32)).all()
        assert gs.bytes_received == 0 and gs.bytes_sent == 0 and gs.ms_payload >= 0 and gs.ms_counts > 0
    g, counts, gs = want.gatherv_start(comm, gpu.GATHER_NONE, 0).wait()
    assert g is None and counts == [want.n_tris()]
    with pytest.raises(gpu.HipError):
        want.gatherv_start(comm, 7, 0)
    with pytest.raises(gpu.HipError):
        want.gatherv_start(comm, gpu.GATHER_ROOT, 3)
    # pipelined: the payload of mesh i moves while mesh i+1 is made; results identical to the unpipelined ones
    pend, got = None, []
    for i in range(4):
        oc = gpu.OctreeHIP(sdf, res)
        nxt = oc.gatherv_start(comm, gpu.GATHER_ALL, 0)
        if pend is not None:
            got.append(pend.wait()[0])
        pend = nxt
    got.append(pend.wait()[0])
    for g in got:
        assert g.n_tris() == want.n_tris() and (srt(g.RenderAll()).view(np.uint32) == srt(ref).view(np.uint32)).all()
    # a pending gather that is dropped without wait() is completed and released by its finaliser
    gpu.OctreeHIP(sdf, res).gatherv_start(comm, gpu.GATHER_ALL, 0)
    comm.close()


def test_mesh_outlives_its_program(gpu):
    """A mesh owns what it needs to be read: destroying the program handle first (Go finalisers / Python GC run in any
    order) must not break later reads or the on-device STL build."""
    b = Builder()
    sh = b.NewSphere(1.0)
    sdf = gpu.SDF3HIP(sh)
    oc = gpu.OctreeHIP(sdf, np.float32(1.0 / 20))
    want = oc.RenderAll().copy()
    oc2 = gpu.OctreeHIP(sdf, np.float32(1.0 / 20))                     # a second mesh, read only after the program is gone
    sdf.close()                                                        # gsdf_hip_program_destroy: its stream is gone
    srt = lambda t: t.reshape(-1, 9)[np.lexsort(t.reshape(-1, 9).view(np.uint32).T[::-1])]
    assert (srt(oc2.RenderAll()) == srt(want)).all()                   # first read of oc2: DMA on the mesh's own stream
    assert len(oc2.WriteBinarySTL()) == 84 + 50 * oc2.n_tris()      # stl_kernel on the mesh's own stream
    assert oc2.triangles_view().shape == want.shape
