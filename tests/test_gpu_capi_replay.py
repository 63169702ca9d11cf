"""The cgo shim of INTEGRATION.md replayed from plain C on the GPU (tests/capi_shim_replay.c): create -> Evaluate with the
reference's error values -> octree mesh -> the 4096-triangle ReadTriangles loop of glrender.RenderAll -> the STL file ->
RCCL communicator and gatherv at world size 1. No Python between the program and the C ABI."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_replay(out_dir):
    inc, libdir = os.path.join(ROOT, "include"), os.path.join(ROOT, "gsdf_amd", "csrc")
    exe = os.path.join(str(out_dir), "capi_shim_replay")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-Werror", "-I", inc, os.path.join(ROOT, "tests", "capi_shim_replay.c"),
                           "-L", libdir, "-lgsdfhip", "-lm", "-Wl,-rpath," + libdir, "-o", exe])
    return exe


@pytest.mark.gpu
def test_cgo_shim_call_sequence_from_c(gpu, tmp_path):
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    exe = build_replay(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "replay ok: 41072 triangles in 11 ReadTriangles calls, STL 2053684 bytes" in r.stdout   # (RCCL prints its banner first)


def test_replay_program_compiles_and_links():
    """CPU suite: the replay program builds against include/*.h with -Werror and links every symbol it uses."""
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        assert os.path.exists(build_replay(d))
