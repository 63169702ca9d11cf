#!/usr/bin/env python3
"""Mesh one of the reference's example parts on the GPU and write a binary STL, like the reference's
examples/<part>/main.go does through gsdfaux.RenderShader3D (gsdfaux/gsdfaux.go:63-241):

    python examples/render_stl.py npt-flange --resdiv 400 -o npt-flange.stl [--renderer octree|flat|dualcontour]

The tree is built with the host mirror of gsdf.Builder, lowered and (unless --interpreter) compiled into kernels
specialised for it; the mesh stays on the device until the STL records have been built there, and the file arrives in
pinned host memory by one DMA (gsdf_hip_mesh_host_stl)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("scene", choices=["npt-flange", "bolt", "knurled-cylinder", "glyph-plate"])
    ap.add_argument("--resdiv", type=int, default=400, help="resolution = bounding-box diagonal / resdiv (the examples' -resdiv)")
    ap.add_argument("-o", "--output", default=None)
    ap.add_argument("--renderer", choices=["octree", "flat", "dualcontour"], default="octree")
    ap.add_argument("--interpreter", action="store_true", help="skip the per-tree kernel build")
    args = ap.parse_args(argv)

    import numpy as np
    from gsdf_amd import hip
    from scaffold.builder import Builder

    hip.init(0)
    t0 = time.perf_counter()
    shape = Builder().Scene(args.scene)
    sdf = hip.SDF3HIP(shape)
    if not args.interpreter:
        sdf.specialize()
    t1 = time.perf_counter()
    res = np.float32(float(shape.Diagonal()) / args.resdiv)
    mesh = {"octree": hip.OctreeHIP, "flat": hip.FlatHIP, "dualcontour": hip.DualContourHIP}[args.renderer](sdf, res)
    t2 = time.perf_counter()
    out = args.output or f"{args.scene}.stl"
    stl = mesh.stl_view()
    with open(out, "wb") as f:
        f.write(stl)
    t3 = time.perf_counter()
    st = mesh.stats
    print(f"{args.scene} resdiv {args.resdiv} ({args.renderer}): {st.evals} evaluations, {mesh.n_tris()} triangles; "
          f"setup {t1 - t0:.2f} s, mesh {(t2 - t1) * 1e3:.2f} ms (device {st.ms_total:.2f} ms), "
          f"STL {len(stl) / 1e6:.1f} MB written to {out} in {(t3 - t2) * 1e3:.1f} ms")
    return 0


if __name__ == "__main__":
    sys.exit(main())
