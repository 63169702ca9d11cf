"""Trees side by side. A specialised build is an out-of-process compiler run of seconds (gsdf_hip_program_specialize: the installed
hipcc), and the GPU suite builds a few hundred of them: one after the other they were two thirds of its 900+ seconds. Handles are
independent (own streams and workspaces, locked pools: tests/test_gpu_mesh.py::test_concurrent_meshing_from_host_threads), the
calls release the GIL, so the per-tree bodies of the build-heavy tests run on a small pool of threads."""
from concurrent.futures import ThreadPoolExecutor


def pmap(fn, items, workers=8):
    """[fn(x) for x in items] on `workers` threads, results in order. The first failure is re-raised (with the item's index)."""
    items = list(items)
    if not items:
        return []
    with ThreadPoolExecutor(max_workers=min(workers, len(items))) as ex:
        futs = [ex.submit(fn, x) for x in items]
        out = []
        for i, f in enumerate(futs):
            try:
                out.append(f.result())
            except BaseException as e:  # noqa: BLE001 -- re-raised with the item
                raise type(e)(f"item {i}: {e}") if isinstance(e, AssertionError) else e
        return out
