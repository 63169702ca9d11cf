"""Scaffolding, not product: a C++ / Python mirror of the reference's gsdf.Builder, forge/threads and forge/textsdf that
produces gsdf_tree blobs for the tests, the benchmark and the examples (the reference's own builder is Go and cannot run
here). libgsdfhip.so does not depend on anything in this directory."""
from .builder import Builder, Shader, ShapeError  # noqa: F401
