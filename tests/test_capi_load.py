"""The C-ABI library loads without a GPU and exports every symbol include/gsdf_hip.h declares; host-only
entry points behave (no compute calls here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from gsdf_amd import hip
from gsdf_amd._ctypes_common import GsdfNode, GsdfTree, OP
from gsdf_amd.builder import Builder

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported():
    hdr = open(os.path.join(ROOT, "include", "gsdf_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(gsdf_hip_\w+)\s*\(", hdr)))
    assert len(declared) >= 19
    L = hip.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert sorted(declared) == sorted(hip.SYMBOLS)


def test_enum_matches_header():
    hdr = open(os.path.join(ROOT, "include", "gsdf_program.h")).read()
    body = hdr[hdr.index("enum gsdf_op {"):hdr.index("GSDF_OP_COUNT")]
    names = re.findall(r"\bGSDF_(?:OP_)?([A-Z0-9_]+)\b(?=\s*(?:=\s*0)?,)", body)
    from gsdf_amd._ctypes_common import OPS
    assert names == OPS


def test_malformed_trees_are_rejected_on_host():
    L = hip.lib()
    nodes = (GsdfNode * 1)()
    nodes[0].op = 999
    t = GsdfTree(nodes, 1, None, 0, None, 0, 0)
    h = C.c_void_p()
    assert L.gsdf_hip_program_create(C.byref(t), C.byref(h)) == -4  # GSDF_ERR_BAD_TREE
    assert b"bad op" in L.gsdf_hip_last_error()
    nodes[0].op = OP["UNION"]
    nodes[0].nchild = 1
    links = (C.c_uint32 * 1)(0)
    t = GsdfTree(nodes, 1, links, 1, None, 0, 0)
    assert L.gsdf_hip_program_create(C.byref(t), C.byref(h)) == -4
    assert L.gsdf_hip_program_create(None, C.byref(h)) == -3         # GSDF_ERR_BAD_ARGUMENT


def test_no_gpu_is_reported_not_hidden():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    b = Builder()
    with pytest.raises(hip.HipError) as e:
        hip.SDF3HIP(b.NewSphere(1))
    assert e.value.code in (-6, -5)  # GSDF_ERR_NO_DEVICE / HIP error: never a silent CPU fallback


def test_brick_owner_partition_is_balanced_and_total():
    """Multi-GPU partition (host-visible twin of the device function): every brick has exactly one owner,
    ownership depends only on the coordinates, and the deal is balanced."""
    L = hip.lib()
    coords = [(x, y, z) for x in range(24) for y in range(24) for z in range(8)]
    for world in (1, 2, 3, 4, 8):
        owners = np.array([L.gsdf_hip_brick_owner(x, y, z, world) for x, y, z in coords])
        assert owners.min() >= 0 and owners.max() < world
        counts = np.bincount(owners, minlength=world)
        assert counts.sum() == len(coords)
        assert counts.max() < 1.25 * len(coords) / world and counts.min() > 0.75 * len(coords) / world
        again = np.array([L.gsdf_hip_brick_owner(x, y, z, world) for x, y, z in coords])
        assert (owners == again).all()
    assert L.gsdf_hip_brick_owner(1, 2, 3, 0) == 0
