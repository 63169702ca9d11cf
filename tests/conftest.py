import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_libs():
    """Build the native pieces once if a fresh checkout has none (CPU-only: hipcc cross-compiles)."""
    need = [os.path.join(ROOT, "scaffold", "libgsdfhost.so"), os.path.join(ROOT, "gsdf_amd", "csrc", "libgsdfhip.so"),
            os.path.join(ROOT, "oracle", "liborc.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "__graft_entry__.py")])


@pytest.fixture(scope="session", autouse=True)
def _code_object_cache(tmp_path_factory):
    """One code-object cache for the whole run (GSDF_HIP_CACHE_DIR, specialize.cpp): the example scenes are specialised by many
    tests, and each build is an out-of-process compiler run of seconds -- after the first, a file read. Tests of the cache itself set
    their own directory."""
    if os.environ.get("GSDF_HIP_CACHE_DIR"):
        yield
        return
    d = tmp_path_factory.mktemp("gsdf_code_objects")
    os.environ["GSDF_HIP_CACHE_DIR"] = str(d)
    yield
    os.environ.pop("GSDF_HIP_CACHE_DIR", None)


@pytest.fixture(scope="session")
def gpu():
    # Tests that hand device buffers to torch (the RCCL gather) need torch's bundled HIP runtime initialised BEFORE the
    # system one that libgsdfhip.so links: in the other order torch reports "No HIP GPUs are available". Test order must
    # not matter, so do it here.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    from gsdf_amd import hip
    hip.init(0)  # raises loudly when no device / no library: GPU tests must never fall back
    return hip
