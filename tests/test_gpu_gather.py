"""Multi-GPU exchange through the C ABI (gsdf_hip_comm_* / gsdf_hip_mesh_gatherv*, include/gsdf_hip.h).
  * on RCCL at world size 1 -- what one GPU allows of librccl: the communicator, the count all-gather, the plan's copy, the
    reduction, so the `nccl` path executes at least once per round;
  * on the in-process loopback transport (GSDF_HIP_COMM=loopback: the ranks are threads, a transfer is a device copy ordered by
    events) at world sizes 2, 3 and 8: the SHIPPING code of gsdf_hip_mesh_gatherv_start -- counts, plan, one group of sends and
    receives, marching cubes over gathered records -- with every rank meshing its own shard, all modes, both payloads, pipelined;
  * the records payload against the triangle payload: same triangle set, bit for bit.
The plan itself is checked as data and executed over gloo on CPU (tests/test_gather_gloo.py)."""
import os
import threading

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


def _srt(t):
    t = np.ascontiguousarray(t, np.float32).reshape(-1, 9)
    return t[np.lexsort(t.view(np.uint32).T[::-1])]


def test_records_payload_gives_the_same_triangles(gpu):
    """gsdf_mesh_opts.payload = records: the mesher stops at the packed cut-leaf records (scan_groups_kernel + pack_records_kernel);
    gsdf_hip_mesh_march (march_dense_kernel) makes the triangles afterwards -- the same set as the default path's, and the
    accessors refuse a mesh that has none yet."""
    b = Builder()
    for name, rd, spec in (("npt-flange", 300, True), ("bolt", 150, False), ("knurled-cylinder", 120, True)):
        sh = b.Scene(name)
        sdf = gpu.SDF3HIP(sh)
        if spec:
            sdf.specialize()
        res = np.float32(float(sh.Diagonal()) / rd)
        want = gpu.OctreeHIP(sdf, res)
        for shard in ((0, 1), (1, 3)):
            ref = gpu.OctreeHIP(sdf, res, shard_rank=shard[0], shard_count=shard[1])
            rec = gpu.OctreeHIP(sdf, res, shard_rank=shard[0], shard_count=shard[1], payload=gpu.PAYLOAD_RECORDS)
            kind, nrec, nbytes = rec.payload()
            assert kind == gpu.PAYLOAD_RECORDS and nrec == ref.stats.cut_leaves == rec.stats.cut_leaves
            assert nbytes == nrec * 40 + (((nrec + 255) // 256 * 4 + 7) & ~7)
            assert rec.n_tris() == ref.n_tris() and rec.stats.evals == ref.stats.evals and rec.TotalPruned() == ref.TotalPruned()
            with pytest.raises(gpu.HipError):
                rec.RenderAll()
            with pytest.raises(gpu.HipError):
                rec.triangles_view()
            rec.march()
            assert rec.payload() == (gpu.PAYLOAD_TRIANGLES, 0, 0)
            assert (_srt(rec.RenderAll()).view(np.uint32) == _srt(ref.RenderAll()).view(np.uint32)).all(), (name, shard)
            assert rec.march() is rec                                   # no-op on triangles
        assert want.n_tris() > 10000
    # a mesh without surface: no records, no triangles, still a mesh
    small = gpu.SDF3HIP(b.NewSphere(1.0))
    shards = [gpu.OctreeHIP(small, np.float32(0.25), shard_rank=r, shard_count=64, payload=gpu.PAYLOAD_RECORDS) for r in range(64)]
    empty = next(m for m in shards if m.n_tris() == 0)
    assert empty.payload() == (gpu.PAYLOAD_RECORDS, 0, 0) and empty.march().RenderAll().shape == (0, 3, 3)
    # small record buffers overflow once and are repeated with the exact size (the handle remembers the last mesh's count)
    sdf = gpu.SDF3HIP(b.Scene("npt-flange"))
    r1 = gpu.OctreeHIP(sdf, np.float32(float(b.Scene("npt-flange").Diagonal()) / 60), payload=gpu.PAYLOAD_RECORDS)
    r2 = gpu.OctreeHIP(sdf, np.float32(float(b.Scene("npt-flange").Diagonal()) / 400), payload=gpu.PAYLOAD_RECORDS)
    assert r2.stats.cut_leaves > 17 * r1.stats.cut_leaves / 16 + 1024
    assert r2.march().n_tris() == 423852


def _loopback_world(gpu, world, work):
    """Run work(rank, comm) on `world` threads, each a rank of one loopback communicator; returns the per-rank results."""
    old = os.environ.get("GSDF_HIP_COMM")
    os.environ["GSDF_HIP_COMM"] = "loopback"
    try:
        uid = gpu.CommHIP.unique_id()
    finally:
        if old is None:
            del os.environ["GSDF_HIP_COMM"]
        else:
            os.environ["GSDF_HIP_COMM"] = old
    out, err = [None] * world, [None] * world

    def run(r):
        try:
            gpu.init(0)
            comm = gpu.CommHIP(uid, r, world)
            assert comm.transport() == "loopback"
            out[r] = work(r, comm)
            comm.close()
        except BaseException as e:  # noqa: BLE001 -- reported by the main thread
            err[r] = e
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
    assert not any(t.is_alive() for t in th), "a rank hangs"
    for e in err:
        if e is not None:
            raise e
    return out


@pytest.mark.parametrize("world", [2, 3, 8])
def test_gatherv_across_ranks_on_the_loopback_transport(gpu, world):
    """Every rank meshes its own shard (brick_owner partition) and the ranks gather: the gathered mesh is the whole mesh --
    bit-identical triangle set -- on every receiving rank, rank-major with the ranks' own counts, for both payloads and all
    three modes; the statistics add up through the all-reduce."""
    b = Builder()
    sh = b.Scene("npt-flange")
    res = np.float32(float(sh.Diagonal()) / 260)
    whole = gpu.OctreeHIP(gpu.SDF3HIP(sh), res)
    want = _srt(whole.RenderAll())
    want_evals_leaf, want_tris = int(whole.stats.evals_leaf), whole.n_tris()

    def work(r, comm):
        sdf = gpu.SDF3HIP(sh)
        if r % 2:
            sdf.specialize()
        res_r = {}
        for payload in (gpu.PAYLOAD_TRIANGLES, gpu.PAYLOAD_RECORDS):
            mine = gpu.OctreeHIP(sdf, res, shard_rank=r, shard_count=world, payload=payload)
            tot = comm.allreduce_sum([mine.n_tris(), int(mine.stats.evals_leaf)])
            for mode, root in ((gpu.GATHER_ALL, 0), (gpu.GATHER_ROOT, world - 1), (gpu.GATHER_NONE, 0)):
                g, counts, gs = mine.gatherv_start(comm, mode, root).wait()
                recv = mode == gpu.GATHER_ALL or (mode == gpu.GATHER_ROOT and r == root)
                assert (g is not None) == recv
                assert counts[r] == mine.n_tris() and sum(counts) == want_tris
                unit = None if payload == gpu.PAYLOAD_RECORDS else 36
                if recv:
                    t = g.RenderAll()
                    assert (_srt(t).view(np.uint32) == want.view(np.uint32)).all()
                    if unit:                                           # triangles: rank-major, each rank's block in its own order
                        o = sum(counts[:r])
                        assert (t.reshape(-1, 9)[o:o + counts[r]].view(np.uint32) == mine.RenderAll().reshape(-1, 9).view(np.uint32)).all()
                        assert gs.bytes_received == 36 * (want_tris - counts[r])
                    else:
                        assert 0 < gs.bytes_received < 36 * (want_tris - counts[r]) * 0.62 and gs.ms_march > 0
                if mode == gpu.GATHER_ALL and unit:
                    assert gs.bytes_sent == 36 * counts[r] * (world - 1)
                if mode == gpu.GATHER_NONE:
                    assert gs.bytes_sent == 0 and gs.bytes_received == 0
            res_r[payload] = tot
        return res_r

    for tot in _loopback_world(gpu, world, work):
        for payload in (gpu.PAYLOAD_TRIANGLES, gpu.PAYLOAD_RECORDS):
            assert tot[payload] == [want_tris, want_evals_leaf]        # the shards partition the leaf work exactly


def test_pipelined_gathers_and_early_release_on_the_loopback_transport(gpu):
    """bench.py's loop at world size 3: the payload of mesh i moves while mesh i+1 is made, on the SAME renderer handle
    (Reset under a gather in flight: the library keeps the gathered buffers until the payload has moved), records payload."""
    b = Builder()
    sh = b.Scene("bolt")
    res = np.float32(float(sh.Diagonal()) / 170)
    want = _srt(gpu.OctreeHIP(gpu.SDF3HIP(sh), res).RenderAll())

    def work(r, comm):
        sdf = gpu.SDF3HIP(sh)
        sdf.specialize()
        oc = gpu.OctreeHIP(sdf, res, shard_rank=r, shard_count=3, payload=gpu.PAYLOAD_RECORDS)
        pend, got = None, []
        for i in range(5):
            nxt = oc.gatherv_start(comm, gpu.GATHER_ALL, 0)
            oc.Reset(sdf, res)                                        # the next mesh, on the same handle, under the gather
            if pend is not None:
                got.append(pend.wait()[0])
            pend = nxt
        got.append(pend.wait()[0])
        return [bool((_srt(g.RenderAll()).view(np.uint32) == want.view(np.uint32)).all()) for g in got]

    for oks in _loopback_world(gpu, 3, work):
        assert oks == [True] * 5


@pytest.mark.parametrize("payload_name", ["records", "triangles"])
def test_gathers_under_three_meshes_in_flight_on_the_loopback_transport(gpu, payload_name):
    """bench.py's N > 1 loop since round 6 (run_meshes: rank_pipeline) at world size 3: every rank keeps three meshes of its shard in flight
    on ONE handle (gsdf_hip_mesh_octree_start / _wait), starts the gather of mesh k when mesh k is done and waits for the gather of
    mesh k - 1 -- eight pooled buffers circulate per rank. Every gathered mesh is the whole mesh, bit for bit."""
    payload = gpu.PAYLOAD_RECORDS if payload_name == "records" else gpu.PAYLOAD_TRIANGLES
    b = Builder()
    sh = b.Scene("npt-flange")
    res = np.float32(float(sh.Diagonal()) / 300)
    want = _srt(gpu.OctreeHIP(gpu.SDF3HIP(sh), res).RenderAll())
    n = 9

    def work(r, comm):
        sdf = gpu.SDF3HIP(sh)
        if r != 1:
            sdf.specialize()
        inflight, started, pend, got, keep = [], 0, None, [], None
        for k in range(n):
            while started < n and len(inflight) < 3:
                inflight.append(gpu.OctreeHIP.start(sdf, res, shard_rank=r, shard_count=3, payload=payload))
                started += 1
            oc = inflight.pop(0).wait()
            nxt = oc.gatherv_start(comm, gpu.GATHER_ALL, 0)
            if pend is not None:
                got.append(pend.wait()[0])
            pend, keep = nxt, oc                                       # (the mesh before `oc` is released here, its gather waited for)
        got.append(pend.wait()[0])
        return [bool((_srt(g.RenderAll()).view(np.uint32) == want.view(np.uint32)).all()) for g in got]

    for oks in _loopback_world(gpu, 3, work):
        assert oks == [True] * n


def test_gather_rejects_mixed_payloads(gpu):
    b = Builder()
    sh = b.NewSphere(1.0)

    def work(r, comm):
        sdf = gpu.SDF3HIP(sh)
        m = gpu.OctreeHIP(sdf, np.float32(0.05), shard_rank=r, shard_count=2, payload=gpu.PAYLOAD_RECORDS if r else gpu.PAYLOAD_TRIANGLES)
        with pytest.raises(gpu.HipError):
            m.gatherv_start(comm, gpu.GATHER_ALL, 0)
        return True

    assert _loopback_world(gpu, 2, work) == [True, True]


@pytest.mark.parametrize("world", [2, 3])
def test_gatherv_across_processes_sharing_a_device(gpu, world, tmp_path):
    """The N > 1 path with the ranks as PROCESSES -- each its own HIP context on device 0 -- over the library's inter-process
    transport (GSDF_HIP_COMM=ipc: RCCL refuses two ranks on one GPU): the id travels through a file, every rank meshes its shard and
    the ranks gather in all three modes with both payloads, then pipelined as bench.py does. The gathered mesh is the whole mesh, bit
    for bit, on every receiving rank (tests/ipc_rank.py is one rank)."""
    import hashlib
    import json
    import subprocess
    import sys
    b = Builder()
    sh = b.Scene("npt-flange")
    res = np.float32(float(sh.Diagonal()) / 260)
    whole = gpu.OctreeHIP(gpu.SDF3HIP(sh), res)
    want = hashlib.sha256(_srt(whole.RenderAll()).tobytes()).hexdigest()
    want_tris, want_evals = whole.n_tris(), int(whole.stats.evals_leaf)
    env = dict(os.environ, GSDF_HIP_COMM="ipc", HSA_ENABLE_IPC_MODE_LEGACY="0", GSDF_HIP_IPC_TIMEOUT_S="90")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ipc_rank.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = []
    for pr in procs:
        try:
            o, _ = pr.communicate(timeout=400)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("a rank hangs")
        logs.append(o.decode(errors="replace")[-3000:])
    assert all(pr.returncode == 0 for pr in procs), "\n----\n".join(logs)
    for r in range(world):
        j = json.load(open(tmp_path / f"rank{r}.json"))
        assert len(j["modes"]) == 6
        for row in j["modes"]:
            assert row["total"] == [want_tris, want_evals] and sum(row["counts"]) == want_tris and row["counts"][r] == row["own"]
            recv = row["mode"] == gpu.GATHER_ALL or (row["mode"] == gpu.GATHER_ROOT and r == row["root"])
            assert row["received"] == recv
            if recv:
                assert row["n_tris"] == want_tris and row["sha256_sorted"] == want
                if row["payload"] == gpu.PAYLOAD_TRIANGLES:
                    assert row["bytes_received"] == 36 * (want_tris - row["own"])
            if row["mode"] == gpu.GATHER_NONE:
                assert row["bytes_sent"] == 0 and row["bytes_received"] == 0
        assert j["pipelined"] == [want] * 3


def test_bench_self_launches_two_ranks_on_one_gpu(gpu):
    """`GSDF_HIP_COMM=ipc python3 bench.py --gpus 2` with NO launcher -- the form the driver starts the N = 1 line in -- must start its
    own ranks (torch.distributed.run, one process per rank) and print the one JSON line: VALU roofline like the N = 1 line, the three
    gather modes timed in the same run, the gathered count equal to the whole mesh's."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(GSDF_HIP_COMM="ipc", HSA_ENABLE_IPC_MODE_LEGACY="0", GSDF_HIP_IPC_TIMEOUT_S="120")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--preheat", "3", "--resdiv", "400"]
    pr = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert pr.returncode == 0, pr.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in pr.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and d["schema"] == 6
    assert d["triangles_per_step"] == 423852                         # README.md:116,130 -- the two shards together are the whole mesh
    rf = d["roofline"]
    assert rf["bound"] == "valu" and rf["kernel_ms"] > 0 and "leaf_eval_kernel" in rf["kernel"]
    assert rf["frac"] is None or 0 < rf["frac"] <= 1
    assert "frac" not in rf["hbm_notional"]
    gm = d["gather_modes"]
    assert set(gm) >= {"all", "root", "none"}
    for mode in ("all", "root", "none"):
        assert gm[mode]["ms_per_step"] > 0 and gm[mode]["mode"] == mode
    assert gm["none"]["bytes_received_per_rank"] == 0 and gm["all"]["bytes_received_per_rank"] > 0
    assert d["gather"]["mode"] == "all" and "ranks share a device" in d["config"]["devices"]
