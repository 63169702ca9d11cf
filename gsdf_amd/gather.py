"""The library's gather plan (gsdf_hip_gather_plan, include/gsdf_hip.h) executed over torch.distributed point-to-point
transfers instead of the library's own RCCL communicator.

The product's multi-GPU exchange is gsdf_hip_mesh_gatherv_start / _wait: RCCL called directly inside libgsdfhip.so, which
runs exactly the list of copies / sends / receives that gsdf_hip_gather_plan returns for the ranks' payload sizes. This module
runs THE SAME LIST as one dist.batch_isend_irecv -- on gloo for the CPU tests of the schedule and the rank-major layout
(tests/test_gather_gloo.py, world sizes 2 and 3, ragged and empty ranks, all three modes), and on torch's RCCL process group
as bench.py's fallback when the library's communicator cannot be had. Nothing is padded: the counts are exchanged (one
all_gather of world int64) and every transfer has its exact size.

The mesher shards octree bricks across ranks with no data-path communication; the only exchange is this final gather
(SURVEY.md 8(e)).
"""
import torch
import torch.distributed as dist

from . import hip


class _DevArray:
    """Zero-copy view of a raw device pointer for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr, nfloats):
        self.__cuda_array_interface__ = {"shape": (int(nfloats),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def tensor_from_dev_ptr(ptr, n_tris, device):
    if n_tris == 0 or not ptr:
        return torch.empty((0, 9), dtype=torch.float32, device=device)
    return torch.as_tensor(_DevArray(ptr, n_tris * 9), device=device).view(n_tris, 9)


def run_plan(payload, mode=hip.GATHER_ALL, root=0, group=None):
    """payload: 1-D uint8 tensor (this rank's bytes). Exchanges the sizes, asks the library for this rank's plan and runs it.
    Returns (gathered uint8 tensor -- empty on ranks that receive nothing --, bytes per rank, the plan)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = payload.device
    n = torch.tensor([payload.numel()], dtype=torch.int64, device=dev)
    sizes = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, n, group=group)
    sizes_l = [int(c) for c in sizes.tolist()]
    ops, total = hip.gather_plan(sizes_l, rank, mode, root)
    out = torch.empty(total, dtype=torch.uint8, device=dev)
    # The sends and receives run as ONE batch (dist.batch_isend_irecv): on torch's RCCL process group that is one
    # ncclGroupStart / End around all of them, like the library's own executor. Issued one by one, both ranks of a pair would
    # enqueue their send in front of their receive on the pair's communicator, and a payload beyond the eager limit (a mesh
    # is tens of MB) then waits for a receive that sits behind the peer's own send. gloo takes the same batch.
    p2p = []
    for kind, peer, src_off, dst_off, nbytes in ops:
        if kind == hip.GOP_COPY:
            out[dst_off:dst_off + nbytes] = payload[src_off:src_off + nbytes]
        elif kind == hip.GOP_SEND:
            p2p.append(dist.P2POp(dist.isend, payload[src_off:src_off + nbytes].contiguous(), peer, group))
        else:
            p2p.append(dist.P2POp(dist.irecv, out[dst_off:dst_off + nbytes], peer, group))
    if p2p:
        for r in dist.batch_isend_irecv(p2p):
            r.wait()
    return out, sizes_l, ops


def all_gatherv(local, group=None):
    """local: (n_i, 9) float32 tensor on this rank's device. Returns ((sum n_i, 9) tensor, counts list), rank-major order,
    identical on every rank: mode ALL of the library's plan."""
    # (the transfers work on torch-owned memory: `local` may be a view of a buffer that another HIP runtime instance
    # allocated -- libgsdfhip.so links the system runtime, PyTorch bundles its own)
    payload = local.contiguous().view(torch.uint8).reshape(-1).clone()
    out, sizes, _ = run_plan(payload, hip.GATHER_ALL, 0, group)
    return out.view(torch.float32).view(-1, 9), [s // 36 for s in sizes]


def all_gatherv_triangles(dev_ptr, n_tris, device, group=None):
    """Gather the device-resident triangle buffers of all ranks (OctreeHIP.dev_ptr()/n_tris())."""
    return all_gatherv(tensor_from_dev_ptr(dev_ptr, n_tris, device), group)
