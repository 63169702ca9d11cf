"""The C-ABI library loads without a GPU and exports every symbol include/gsdf_hip.h declares; host-only
entry points behave (no compute calls here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from gsdf_amd import hip
from gsdf_amd._ctypes_common import GsdfNode, GsdfTree, OP
from scaffold.builder import Builder

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


def test_headers_are_plain_c_and_the_library_links_from_c(tmp_path):
    """cgo sees include/*.h as C: they must compile as C11, and a C program must be able to link the library and drive a
    host-only entry point (a one-node tree through gsdf_hip_lower) -- no C++ types, no name mangling at the boundary."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = tmp_path / "abi.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "gsdf_hip.h"
int main(void) {
  gsdf_node n;
  memset(&n, 0, sizeof n);
  n.op = GSDF_SPHERE;
  n.p[0] = 1.5f;
  gsdf_tree t;
  memset(&t, 0, sizeof t);
  t.nodes = &n; t.n_nodes = 1; t.root = 0;
  t.bb[0] = t.bb[1] = t.bb[2] = -1.5f; t.bb[3] = t.bb[4] = t.bb[5] = 1.5f;
  uint32_t code[64], words = 0, slots = 0;
  int rc = gsdf_hip_lower(&t, code, 64, &words, &slots);
  if (rc != GSDF_OK) { printf("error %d: %s\n", rc, gsdf_hip_last_error()); return 1; }
  printf("words %u slots %u owner %u\n", words, slots, gsdf_hip_brick_owner(1, 2, 3, 8));
  return 0;
}
''')
    inc = os.path.join(ROOT, "include")
    libdir = os.path.join(ROOT, "gsdf_amd", "csrc")
    for h in ("gsdf_hip.h", "gsdf_program.h"):
        subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(inc, h)])
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-I", inc, str(src), "-L", libdir, "-lgsdfhip", "-Wl,-rpath," + libdir, "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True)
    m = re.match(r"words (\d+) slots (\d+) owner (\d+)", out)
    assert m and int(m.group(1)) >= 2 and int(m.group(3)) == hip.lib().gsdf_hip_brick_owner(1, 2, 3, 8)


def test_slab_partition_tiles_the_lattice():
    """The z-slab split of the flat renderer and of dual contouring (multi-GPU, no data-path collective): for any plane
    count and world size the ranks' slabs are contiguous, disjoint, in order, cover [0, n) and differ by at most one."""
    L = hip.lib()
    lo, hi = C.c_uint32(), C.c_uint32()
    for n in (0, 1, 2, 7, 8, 335, 1024, 2048, 4095, 2**31 + 5):
        for world in (1, 2, 3, 4, 8, 16, 64):
            edges, sizes = [], []
            for r in range(world):
                L.gsdf_hip_slab_range(n, r, world, C.byref(lo), C.byref(hi))
                assert lo.value <= hi.value <= n
                edges.append((lo.value, hi.value))
                sizes.append(hi.value - lo.value)
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            assert max(sizes) - min(sizes) <= 1
    L.gsdf_hip_slab_range(10, 5, 4, C.byref(lo), C.byref(hi))     # rank out of range: empty
    assert (lo.value, hi.value) == (0, 0)
    L.gsdf_hip_slab_range(10, 0, 0, C.byref(lo), C.byref(hi))     # no ranks: empty
    assert (lo.value, hi.value) == (0, 0)
