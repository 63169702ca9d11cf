"""Run-time specialisation, host side (no GPU): the generated evaluator source and a gfx950 hiprtc build of the
specialised kernels (gsdf_hip_specialize_source / gsdf_hip_specialize_check)."""
import ctypes as C
import re

import pytest

from gsdf_amd import hip
from scaffold.builder import Builder


def _source(shader):
    t = shader.tree()
    n = C.c_size_t()
    assert hip.lib().gsdf_hip_specialize_source(C.byref(t), None, 0, C.byref(n)) == 0
    buf = C.create_string_buffer(n.value + 1)
    assert hip.lib().gsdf_hip_specialize_source(C.byref(t), buf, n.value + 1, C.byref(n)) == 0
    return buf.value.decode()


def test_generated_source_follows_the_program():
    b = Builder()
    sh = b.Scene("npt-flange")
    src = _source(sh)
    code, _ = hip.lower(sh)
    # the program words are embedded as literals, in order
    tab = re.search(r"kSpecCode\[(\d+)\] = \{(.*?)\};", src, re.S)
    assert int(tab.group(1)) == len(code)
    words = [int(x.rstrip("u"), 16) for x in re.findall(r"0x[0-9a-f]+u", tab.group(2))]
    assert words == [int(w) for w in code]
    # one block per instruction, in program order, named after the interpreter's cases; no D_END block, no dispatch loop
    blocks = re.findall(r"\{  // (\d+) (D_[A-Z0-9_]+)", src)
    # (brick masks, dev_ops.h: every operand subtree of the four combine frames carries a number -- a D_SKIP in front of it, a
    # D_LIP_PUSH at the frame's entry and a D_LIP_DOM in front of the combine, both for the interval mode of the centre tests)
    assert [n for _, n in blocks] == ["D_LIP_PUSH", "D_SCALE_PRE", "D_LIP_PUSH", "D_SKIP", "D_CYL0", "D_SAVER", "D_SKIP", "D_SAVEP3", "D_LIP_PUSH", "D_SKIP",
                                      "D_TRANSLATE", "D_CYLR", "D_SAVER", "D_LOADP3", "D_SKIP", "D_GATE3D", "D_LIP_PUSH", "D_SKIP", "D_CYL0", "D_SAVER",
                                      "D_SKIP", "D_GATEZC", "D_LIP_PUSH", "D_SCREW_PRE", "D_LIP_WRAP", "D_POLY2D", "D_LIP_POP", "D_MAXR_SLOT",
                                      "D_LIP_DOM", "D_COMBINE_DIFF", "D_LIP_DOM", "D_COMBINE_SUNION", "D_LIP_DOM", "D_COMBINE_DIFF", "D_MULR", "D_LIP_POP"]
    # the two gates are structured ifs around their children: `if (far_) R = L; else { child }`, closed before the combine -- the
    # interpreter's own test (interp.h: GSDF_GATE_TEST); the six numbered subtrees likewise: `if (mask bit) R = subst; else { subtree }`
    assert src.count("GSDF_GATE_TEST((region_lb_") == 2 and src.count("if (far_) GSDF_GATE_TAKEN else {") == 2
    assert src.count("if (!LIP && ((bmask >> (PU(0) & 31u)) & 1u) != 0u) {") == 6 and src.count("}}  // end of gated child") == 8
    assert src.count("GSDF_LIP_DOM_BODY") == 3
    assert src.index("// end of gated child") < src.index("D_COMBINE_DIFF")
    assert [int(p) for p, _ in blocks] == sorted(int(p) for p, _ in blocks)
    assert "switch (op)" not in src and "readfirstlane" not in src
    # the statements are the interpreter's own: a characteristic line of the cylinder case appears verbatim
    assert "hypot_k<K>(ax, ay, h);" in src and "in[kp] = minf(0.f, maxf(dx, dy));" in src
    # short buffer is reported, not overrun
    t = sh.tree()
    n = C.c_size_t()
    small = C.create_string_buffer(16)
    assert hip.lib().gsdf_hip_specialize_source(C.byref(t), small, 16, C.byref(n)) == hip.lib().gsdf_hip_specialize_source(C.byref(t), small, 16, None) != 0


@pytest.mark.parametrize("scene", ["npt-flange", "knurled-cylinder"])
def test_specialised_kernels_build_for_gfx950(scene):
    """hiprtc needs no GPU: the eval / prune / leaf kernels of the specialised build compile to a gfx950 code object."""
    t = Builder().Scene(scene).tree()
    n = C.c_size_t()
    rc = hip.lib().gsdf_hip_specialize_check(C.byref(t), C.byref(n))
    assert rc == 0, hip.lib().gsdf_hip_last_error().decode()[:2000]
    assert n.value > 10000


def test_tree_that_broke_the_backend_builds():
    """tools/gpu_fuzz_long.sh seed 5009, tree 7 (round 2): with the triangle-count table of leaf_eval_kernel chosen between LDS
    and global memory by a RUN-TIME flag, the compiler merged the two loads into one through a flat pointer and ROCm's backend
    stopped with "Illegal instruction detected ... V_CMP_NE_U32_e32 0, $src_shared_base" -- for 4 of 146 random trees. The
    choice is a template argument now; this tree must build."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fuzz_trees
    _, shapes = fuzz_trees.random_shapes(5009, 10, depth=4)
    t = shapes[7].tree()
    n = C.c_size_t()
    assert hip.lib().gsdf_hip_specialize_check(C.byref(t), C.byref(n)) == 0, hip.lib().gsdf_hip_last_error().decode()[:2000]


def test_specialised_2d_program_builds():
    b = Builder()
    t = b.Union2D(b.NewCircle(1.0), b.Translate2D(b.NewRectangle(1.0, 2.0), 0.5, 0.25)).tree()
    n = C.c_size_t()
    assert hip.lib().gsdf_hip_specialize_check(C.byref(t), C.byref(n)) == 0, hip.lib().gsdf_hip_last_error().decode()[:2000]


def test_code_object_cache_on_disk(tmp_path, monkeypatch):
    """GSDF_HIP_CACHE_DIR: the second build of the same tree is a file read; a damaged file is ignored and rebuilt."""
    import time
    monkeypatch.setenv("GSDF_HIP_CACHE_DIR", str(tmp_path))
    t = Builder().Scene("bolt").tree()
    n1, n2, n3 = C.c_size_t(), C.c_size_t(), C.c_size_t()
    t0 = time.perf_counter()
    assert hip.lib().gsdf_hip_specialize_check(C.byref(t), C.byref(n1)) == 0, hip.lib().gsdf_hip_last_error().decode()[:2000]
    t_build = time.perf_counter() - t0
    files = list(tmp_path.glob("gsdf_*.co"))
    assert len(files) == 1 and not list(tmp_path.glob("*.tmp.*"))
    assert files[0].stat().st_size > n1.value
    t0 = time.perf_counter()
    assert hip.lib().gsdf_hip_specialize_check(C.byref(t), C.byref(n2)) == 0
    t_hit = time.perf_counter() - t0
    assert n2.value == n1.value and t_hit < t_build / 4, (t_build, t_hit)
    # another tree gets another key
    t2 = Builder().Scene("npt-flange").tree()
    assert hip.lib().gsdf_hip_specialize_check(C.byref(t2), C.byref(n3)) == 0
    assert len(list(tmp_path.glob("gsdf_*.co"))) == 2
    # flip a byte in the middle of the first file: checksum mismatch -> rebuilt and rewritten
    raw = bytearray(files[0].read_bytes())
    raw[len(raw) // 2] ^= 0x40
    files[0].write_bytes(bytes(raw))
    assert hip.lib().gsdf_hip_specialize_check(C.byref(t), C.byref(n2)) == 0
    # (a rebuild by the out-of-process compiler is not byte-for-byte reproducible: temporary paths end up in the object)
    assert abs(n2.value - n1.value) < 4096 and files[0].read_bytes() != bytes(raw)
    # truncated file: ignored as well
    files[0].write_bytes(files[0].read_bytes()[:100])
    assert hip.lib().gsdf_hip_specialize_check(C.byref(t), C.byref(n2)) == 0 and n2.value == n1.value
