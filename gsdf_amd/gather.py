"""Variable-length gather of per-rank triangle buffers over torch.distributed -- the gloo-testable REFERENCE of the
rank-major layout. The product's multi-GPU exchange is gsdf_hip_mesh_gatherv (include/gsdf_hip.h): RCCL called directly
inside libgsdfhip.so, no padding and no staging copies; bench.py --gpus N uses that. This module remains for CPU tests
of the ordering (tests/test_gather_gloo.py) and for callers that already hold torch tensors.

The mesher shards octree bricks across ranks with no data-path communication; the only exchange is
this final gather. RCCL has no all-gatherv: exchange the counts (one all_gather of
world int64), pad every rank's payload to the maximum count and run ONE all_gather_into_tensor (one
large collective; bricks are dealt by a coordinate hash so counts are balanced and padding is small), then
compact on device.
"""
import torch
import torch.distributed as dist


class _DevArray:
    """Zero-copy view of a raw device pointer for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr, nfloats):
        self.__cuda_array_interface__ = {"shape": (int(nfloats),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def tensor_from_dev_ptr(ptr, n_tris, device):
    if n_tris == 0 or not ptr:
        return torch.empty((0, 9), dtype=torch.float32, device=device)
    return torch.as_tensor(_DevArray(ptr, n_tris * 9), device=device).view(n_tris, 9)


def all_gatherv(local, group=None):
    """local: (n_i, 9) float32 tensor on this rank's device. Returns ((sum n_i, 9) tensor, counts list),
    rank-major order, identical on every rank."""
    world = dist.get_world_size(group)
    dev = local.device
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    counts = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, n, group=group)
    counts_l = [int(c) for c in counts.tolist()]
    mx = max(counts_l)
    if mx == 0:
        return torch.empty((0, 9), dtype=torch.float32, device=dev), counts_l
    # The collective always works on torch-owned memory: `local` may be a view of a buffer that another HIP runtime
    # instance allocated (libgsdfhip.so links the system runtime, PyTorch bundles its own), and one device copy of a
    # rank's share (30 MB at 8 ranks) is cheap insurance against RCCL's pointer bookkeeping.
    send = torch.empty((mx, 9), dtype=torch.float32, device=dev)   # the padding rows are never read back
    send[: local.shape[0]] = local
    recv = torch.empty((world, mx, 9), dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=group)
    out = torch.cat([recv[r, : counts_l[r]] for r in range(world)], dim=0)
    return out, counts_l


def all_gatherv_triangles(dev_ptr, n_tris, device, group=None):
    """Gather the device-resident triangle buffers of all ranks (OctreeHIP.dev_ptr()/n_tris())."""
    return all_gatherv(tensor_from_dev_ptr(dev_ptr, n_tris, device), group)
