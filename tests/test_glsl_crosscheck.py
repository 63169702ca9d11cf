"""The reference holds a SECOND statement of every node's formula: the GLSL its GPU path compiles (the strings its AppendShaderBody
methods emit + glbuild/glsllib/*.glsl), which its own tests hold to 5e-3 of the CPU evaluators (gsdf_test.go:527-543). No Go runs
here, but that text can be evaluated: tests/glslref renders, for a flattened tree, the GLSL program from the reference's own method
bodies (translated statement by statement, strings verbatim) and interprets it. This test compares it with the oracle on the whole
parity corpus -- every node type, the forge/threads parts and the four BASELINE scenes (npt-flange, bolt, knurled-cylinder, glyph
plate), three of which have no reference-held triangle count to pin them. Supplementary evidence: it shows restatement == the
reference's formulas where no answer is held; it does not replace a Go run (evaluation order, float32 rounding and math32's own
routines are not exercised by it). Build container only: it reads /root/reference (skipped where that does not exist)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import corpus  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference tree (build container)")

REF_TOL = 5e-3    # the reference's own CPU <-> GPU tolerance (gsdf_test.go:529)
TIGHT_TOL = 1e-4  # what the two statements actually agree to on these parts (float32 noise of trees a few units across)

# The two places where the reference's CPU and GLSL statements differ BY CONSTRUCTION (each is checked only where they coincide):
#   transform   GLSL multiplies vec4(p, 0.0): the translation column is dropped (operations.go:385); the CPU path applies it
#               (cpu_evaluators.go:497). The corpus transforms are rotations: no translation, both agree.
#   screw       GLSL adds p.z * atan(taper), the CPU path z * tan(taper) (threads.go:125-127 vs :157): equal at taper 0, 2e-5 apart
#               per unit of z at NPT's 1.79 degrees.


def _reachable(tree):
    seen, todo = set(), [tree.root]
    while todo:
        i = todo.pop()
        if i in seen:
            continue
        seen.add(i)
        n = tree.nodes[i]
        todo += [tree.links[n.link_off + k] for k in range(n.nchild)]
    return seen


def test_reference_glsl_agrees_with_the_oracle_on_the_corpus():
    from glslref.glsl import Interp
    from glslref.goshader import Reference, STRUCT
    from gsdf_amd._ctypes_common import OPS
    from oracle.oracle import OracleSDF
    ref = Reference()
    assert len(ref.body) == 55                      # every shader node struct of the reference has a body to render
    _, s3 = corpus.shapes3d()
    _, s2 = corpus.shapes2d()
    _, sz = corpus.bezier2d()
    rng = np.random.default_rng(11)
    covered, rows = set(), []
    for name, s, dim in [(n, s, 3) for n, s in s3] + [(n, s, 2) for n, s in s2 + sz]:
        tree = s.tree()
        covered |= {STRUCT[OPS[tree.nodes[i].op]] for i in _reachable(tree)}
        src, root, _ = ref.program(tree)
        prog = Interp(src)
        bb = s.Bounds().astype(np.float64)
        lo, hi = (bb[:3], bb[3:]) if dim == 3 else (bb[[0, 1]], bb[[3, 4]])
        c, h = (lo + hi) / 2, (hi - lo) / 2 * 1.25 + 1e-3
        n = 60 if tree.n_nodes and name == "scene_glyph_plate" else 120
        pos = (c + (rng.random((n, dim)) * 2 - 1) * h).astype(np.float32)
        want = OracleSDF(tree).Evaluate(pos)
        got = np.array([prog.call(root, [p.astype(np.float64)]) for p in pos])
        ok = np.isfinite(got)
        assert ok.mean() > 0.98, (name, int((~ok).sum()))   # (a degenerate branch of a GLSL body may be undefined at a point: pow of a negative, 0/0)
        err = float(np.abs(got[ok] - want[ok]).max())
        rows.append((name, len(_reachable(tree)), err))
        assert err <= REF_TOL, (name, err)
        assert err <= TIGHT_TOL, (name, err)
    # every node type is reached by some shape, but the two SSBO variants (no Builder flag of the corpus makes them: their bodies are
    # the plain ones behind a #define, primitives2d.go:151-164,541-546)
    assert set(STRUCT.values()) - covered == set(), set(STRUCT.values()) - covered
    out = os.environ.get("GSDF_GLSL_CROSSCHECK_LOG")
    if out:
        with open(out, "w") as f:
            f.write("# max |reference GLSL (interpreted) - oracle| over 120 points of 1.25 x the bounding box per shape; tests/test_glsl_crosscheck.py\n")
            for name, nn, err in rows:
                f.write("%-28s nodes %3d  max abs diff %.3e\n" % (name, nn, err))
            f.write("# %d shapes, %d node types, worst %.3e (reference's own CPU<->GPU tolerance: 5e-3)\n" % (len(rows), len(covered), max(r[2] for r in rows)))


def test_glsl_interpreter_on_hand_checked_values():
    """The interpreter itself, on values worked out by hand (so that an agreement above is not two wrongs)."""
    from glslref.glsl import Interp
    src = """
    float f(vec3 p){ vec3 q = abs(p) - vec3(0.5,0.5,0.5); return length(max(q,0.0)) + min(max(q.x,max(q.y,q.z)),0.0); }
    float g(vec2 p){ p.x = -p.x; vec2 a[2] = vec2[](vec2(1.,2.),vec2(3.,4.)); float s = 0.; for (int i=0, j=1; i<a.length(); j=i, i++) { s += a[i].y * float(j+1); } return s + p.x; }
    float h(vec2 p){ mat2 m = mat2(0.,1.,-1.,0.); vec2 r = m * p; return (r.x > 0. ? 1. : -1.) * r.y; }
    float k(float x){ float y = x; y *= 2.; y -= 1.; if (y >= 3.) { return y; } else if (y < 0.) return -y; return 0.5; }
    """
    it = Interp(src)
    assert abs(it.call("f", [np.array([1.5, 0.0, 0.0])]) - 1.0) < 1e-15        # outside a unit cube, 1 from the face
    assert abs(it.call("f", [np.array([0.0, 0.0, 0.2])]) + 0.3) < 1e-15        # inside, 0.3 from the nearest face
    assert it.call("g", [np.array([0.25, 9.0])]) == 2 * 2 + 4 * 1 - 0.25       # i=0: j=1 -> 2*2; i=1: j=0 -> 4*1
    assert it.call("h", [np.array([2.0, 3.0])]) == -2.0                        # column-major mat2(0,1,-1,0) * (2,3) = (-3, 2)
    assert [it.call("k", [x]) for x in (2.0, 0.25, 0.75)] == [3.0, 0.5, 0.5]
