"""N>1 path on CPU: the gather schedule the library runs on RCCL -- gsdf_hip_gather_plan, the pure function
gsdf_hip_mesh_gatherv_start executes as one group of sends / receives -- checked as data for world sizes 2, 3 and 8, and
EXECUTED at world sizes 2 and 3 over gloo point-to-point transfers (gsdf_amd/gather.py: run_plan) with ragged and empty
ranks, in all three modes, against the rank-major layout. (The partition itself -- gsdf_hip_brick_owner, gsdf_hip_slab_range
-- is tested in tests/test_capi_load.py; the same plan through the library's own executor runs on the GPU at world sizes
2, 3 and 8 over the in-process loopback transport: tests/test_gpu_gather.py.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from gsdf_amd import hip  # noqa: E402

ALL, ROOT_MODE, NONE = hip.GATHER_ALL, hip.GATHER_ROOT, hip.GATHER_NONE


# ---- the plan as data -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("sizes", [(180, 108), (0, 144), (36, 0, 72), (0, 0, 0), (40, 44, 0, 4, 400, 8, 0, 36), (7,), (1 << 33, 5, 1 << 32)])
@pytest.mark.parametrize("mode", [ALL, ROOT_MODE, NONE])
def test_plan_is_a_consistent_schedule(sizes, mode):
    world = len(sizes)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(object)
    for root in ((0, world - 1) if mode == ROOT_MODE else (0,)):
        plans = [hip.gather_plan(sizes, r, mode, root) for r in range(world)]
        sends, recvs = [], []
        for r, (ops, total) in enumerate(plans):
            receives = mode == ALL or (mode == ROOT_MODE and r == root)
            assert total == (int(off[-1]) if receives else 0)
            covered = []
            for kind, peer, src_off, dst_off, nbytes in ops:
                assert nbytes > 0                                   # empty ranks appear in nobody's list
                if kind == hip.GOP_COPY:
                    assert peer == r and src_off == 0 and nbytes == sizes[r] and dst_off == off[r]
                    covered.append((dst_off, nbytes))
                elif kind == hip.GOP_SEND:
                    assert peer != r and src_off == 0 and nbytes == sizes[r]   # a rank only ever sends its whole payload
                    sends.append((r, peer, nbytes))
                else:
                    assert kind == hip.GOP_RECV and peer != r and nbytes == sizes[peer] and dst_off == off[peer]
                    recvs.append((peer, r, nbytes))
                    covered.append((dst_off, nbytes))
            if receives:                                            # the gathered buffer is tiled exactly once
                covered.sort()
                pos = 0
                for o, n in covered:
                    assert o == pos
                    pos += n
                assert pos == off[-1]
            else:
                assert not covered
        assert sorted(sends) == sorted(recvs)                       # every send has its receive
        if mode == NONE:
            assert not sends and all(not ops for ops, _ in plans)
        if mode == ALL:                                             # a rank with payload sends it to every other rank
            assert len(sends) == sum(world - 1 for s in sizes if s)
            # rotated peer order: at step d everybody sends to rank + d, so the d-th sends of the world hit distinct receivers
            for r, (ops, _) in enumerate(plans):
                peers = [p for k, p, *_ in ops if k == hip.GOP_SEND]
                assert peers == [(r + d) % world for d in range(1, world)][:len(peers)] or not sizes[r]


def test_plan_argument_checks():
    L = hip.lib()
    import ctypes as C
    sizes = (C.c_uint64 * 2)(36, 72)
    n, total = C.c_size_t(), C.c_uint64()
    assert L.gsdf_hip_gather_plan(sizes, 2, 2, ALL, 0, None, 0, C.byref(n), C.byref(total)) == -3      # rank out of range
    assert L.gsdf_hip_gather_plan(sizes, 2, 0, 7, 0, None, 0, C.byref(n), C.byref(total)) == -3        # bad mode
    assert L.gsdf_hip_gather_plan(sizes, 2, 0, ROOT_MODE, 2, None, 0, C.byref(n), C.byref(total)) == -3  # bad root
    assert L.gsdf_hip_gather_plan(sizes, 2, 0, ALL, 0, None, 0, C.byref(n), C.byref(total)) == 0 and n.value == 3 and total.value == 108
    ops = (hip.GatherOp * 2)()
    assert L.gsdf_hip_gather_plan(sizes, 2, 0, ALL, 0, ops, 2, C.byref(n), C.byref(total)) == -9       # short buffer, count still reported
    assert n.value == 3


# ---- the plan executed over gloo ----------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _payload(rank, n):
    return ((np.arange(n, dtype=np.int64) * 7 + 31 * rank) % 251).astype(np.uint8)


def _worker(rank, world, port, cases, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gsdf_amd.gather import run_plan, all_gatherv
    res = []
    for sizes, mode, root in cases:
        out, got_sizes, ops = run_plan(torch.from_numpy(_payload(rank, sizes[rank])), mode, root)
        res.append((out.numpy().copy(), got_sizes, len(ops)))
    n = cases[0][0][rank] // 36                                  # and the triangle form on the first case
    tris, counts = all_gatherv(torch.arange(n * 9, dtype=torch.float32).view(n, 9) + 1000.0 * rank)
    q.put((rank, res, tris.numpy().copy(), counts))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, cases):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, cases, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, outs, tris, counts in res:
        for (sizes, mode, root), (out, got_sizes, n_ops) in zip(cases, outs):
            assert got_sizes == list(sizes)
            receives = mode == ALL or (mode == ROOT_MODE and rank == root)
            want = np.concatenate([_payload(r, n) for r, n in enumerate(sizes)]) if receives else np.zeros(0, np.uint8)
            np.testing.assert_array_equal(out, want)                 # rank-major, identical on every receiving rank
        c0 = [s // 36 for s in cases[0][0]]
        assert counts == c0
        np.testing.assert_array_equal(tris, np.concatenate([np.arange(n * 9, dtype=np.float32).reshape(n, 9) + 1000.0 * r for r, n in enumerate(c0)]))


def test_plan_over_gloo_world2():
    _run(2, [((180, 108), ALL, 0), ((0, 144), ALL, 0), ((72, 40), ROOT_MODE, 1), ((72, 40), ROOT_MODE, 0), ((36, 36), NONE, 0), ((0, 0), ALL, 0)])


def test_plan_over_gloo_world3():
    _run(3, [((360, 0, 108), ALL, 0), ((5, 1000, 17), ALL, 0), ((0, 44, 80), ROOT_MODE, 0), ((40, 0, 0), ROOT_MODE, 2), ((8, 16, 24), NONE, 1)])
