"""N>1 path on CPU: world_size-2 gloo run of the variable-length triangle gather (gsdf_amd/gather.py)
with oracle-free synthetic triangles of ragged per-rank counts (the partition itself -- gsdf_hip_brick_owner,
gsdf_hip_slab_range -- is tested in tests/test_capi_load.py)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, counts, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gsdf_amd.gather import all_gatherv
    n = counts[rank]
    local = (torch.arange(n * 9, dtype=torch.float32).view(n, 9) + 1000.0 * rank)
    out, got_counts = all_gatherv(local)
    q.put((rank, out.numpy().copy(), got_counts))
    dist.barrier()
    dist.destroy_process_group()


def _run(counts):
    world = len(counts)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, counts, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    want = np.concatenate([np.arange(n * 9, dtype=np.float32).reshape(n, 9) + 1000.0 * r for r, n in enumerate(counts)])
    for rank, out, got_counts in res:
        assert got_counts == list(counts)
        np.testing.assert_array_equal(out, want)  # rank-major, identical on every rank


def test_all_gatherv_world2_ragged():
    _run((5, 3))


def test_all_gatherv_world2_one_rank_empty():
    _run((0, 4))
