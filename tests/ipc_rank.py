"""One rank of the multi-PROCESS gather test (tests/test_gpu_gather.py::test_gatherv_across_processes_sharing_a_device): a process of
its own with its own HIP context on device 0, the library's inter-process transport (GSDF_HIP_COMM=ipc), the communicator id handed
over through a file. Meshes its shard, gathers in every mode with both payloads, and writes what it saw as JSON.
    python tests/ipc_rank.py <rank> <world> <dir>"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gsdf_amd import hip as gpu  # noqa: E402
from scaffold.builder import Builder  # noqa: E402


def srt(t):
    t = np.ascontiguousarray(t, np.float32).reshape(-1, 9)
    return t[np.lexsort(t.view(np.uint32).T[::-1])]


def main():
    rank, world, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    gpu.init(0)
    idf = os.path.join(d, "id.bin")
    if rank == 0:
        uid = gpu.CommHIP.unique_id()
        with open(idf + ".tmp", "wb") as f:
            f.write(bytes(uid))
        os.rename(idf + ".tmp", idf)
    else:
        t0 = time.time()
        while not os.path.exists(idf):
            assert time.time() - t0 < 120, "rank 0 never wrote the communicator id"
            time.sleep(0.01)
        uid = open(idf, "rb").read()
    comm = gpu.CommHIP(uid, rank, world)
    assert comm.transport() == "ipc", comm.transport()
    sh = Builder().Scene("npt-flange")
    res = np.float32(float(sh.Diagonal()) / 260)
    sdf = gpu.SDF3HIP(sh)
    if rank % 2:
        sdf.specialize()
    out = {"rank": rank, "modes": []}
    for payload in (gpu.PAYLOAD_TRIANGLES, gpu.PAYLOAD_RECORDS):
        mine = gpu.OctreeHIP(sdf, res, shard_rank=rank, shard_count=world, payload=payload)
        tot = comm.allreduce_sum([mine.n_tris(), int(mine.stats.evals_leaf)])
        for mode, root in ((gpu.GATHER_ALL, 0), (gpu.GATHER_ROOT, world - 1), (gpu.GATHER_NONE, 0)):
            g, counts, gs = mine.gatherv_start(comm, mode, root).wait()
            row = {"payload": int(payload), "mode": int(mode), "root": root, "counts": [int(c) for c in counts], "own": int(mine.n_tris()),
                   "total": [int(x) for x in tot], "bytes_received": int(gs.bytes_received), "bytes_sent": int(gs.bytes_sent), "received": g is not None}
            if g is not None:
                row["sha256_sorted"] = hashlib.sha256(srt(g.RenderAll()).tobytes()).hexdigest()
                row["n_tris"] = int(g.n_tris())
            out["modes"].append(row)
    # bench.py's loop: the payload of mesh i is gathered while mesh i + 1 is made on the same handle
    oc = gpu.OctreeHIP(sdf, res, shard_rank=rank, shard_count=world, payload=gpu.PAYLOAD_RECORDS)
    pend, digests = None, []
    for _ in range(3):
        nxt = oc.gatherv_start(comm, gpu.GATHER_ALL, 0)
        oc.Reset(sdf, res)
        if pend is not None:
            digests.append(hashlib.sha256(srt(pend.wait()[0].RenderAll()).tobytes()).hexdigest())
        pend = nxt
    digests.append(hashlib.sha256(srt(pend.wait()[0].RenderAll()).tobytes()).hexdigest())
    out["pipelined"] = digests
    comm.close()
    with open(os.path.join(d, f"rank{rank}.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
