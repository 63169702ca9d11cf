"""forge/textsdf mirror (scaffold/textsdf.hpp + ttf.hpp) on the reference's own font (tests/golden/iso-3098.ttf is
the data file forge/textsdf embeds and its TestABC uses, forge/textsdf/embed.go:9, glyph_test.go:13-21). No GPU."""
import os
import struct

import numpy as np
import pytest

from scaffold.builder import Builder, ShapeError
from gsdf_amd._ctypes_common import OP
from oracle.oracle import OracleSDF

TTF = open(os.path.join(os.path.dirname(__file__), "golden", "iso-3098.ttf"), "rb").read()


def _tables(d):
    n = struct.unpack(">H", d[4:6])[0]
    return {d[12 + 16 * i:16 + 16 * i].decode(): struct.unpack(">II", d[20 + 16 * i:28 + 16 * i]) for i in range(n)}


def test_reference_test_string_builds_and_has_counters():
    """glyph_test.go TestABC: "Abp8". The line starts at x = 0, glyph heights are < 1 em in the 1/933 scale
    (scaleout = 1/min(font bbox size) = 1/933, font.go:207-212), and the counters of A, b, p, 8 are holes."""
    b = Builder()
    t = b.TextLine(TTF, "Abp8")
    bb = t.Bounds()
    assert -0.05 < bb[0] < 0.05 and 1.5 < bb[3] < 2.2 and -0.3 < bb[1] < 0 and 0.6 < bb[4] < 0.8
    sdf = OracleSDF(t.tree())
    n = 200
    xs, ys = np.meshgrid(np.linspace(bb[0], bb[3], n, dtype=np.float32), np.linspace(bb[1], bb[4], n // 2, dtype=np.float32))
    d = sdf.Evaluate(np.stack([xs, ys], -1).reshape(-1, 2)).reshape(n // 2, n)
    inside = d < 0
    assert 0.10 < inside.mean() < 0.35                      # ink coverage of four stroked letters
    # every letter has a counter: some outside region not connected to the border (flood fill from the frame)
    out = ~inside
    reach = np.zeros_like(out)
    reach[0, :] = out[0, :]; reach[-1, :] = out[-1, :]; reach[:, 0] = out[:, 0]; reach[:, -1] = out[:, -1]
    for _ in range(4 * n):
        grown = reach.copy()
        grown[1:, :] |= reach[:-1, :]; grown[:-1, :] |= reach[1:, :]; grown[:, 1:] |= reach[:, :-1]; grown[:, :-1] |= reach[:, 1:]
        grown &= out
        if (grown == reach).all():
            break
        reach = grown
    holes = out & ~reach
    cols = np.where(holes.any(axis=0))[0]
    assert len(cols) > 0
    groups = 1 + int((np.diff(cols) > 3).sum())
    assert groups >= 4                                       # A, b, p and 8 (two counters, one column group)


def test_metrics_follow_hmtx_and_scale():
    d = TTF
    tabs = _tables(d)
    head = tabs["head"][0]
    upem = struct.unpack(">H", d[head + 18:head + 20])[0]
    xmin, ymin, xmax, ymax = struct.unpack(">4h", d[head + 36:head + 44])
    assert (upem, xmin, ymin, xmax, ymax) == (1000, -123, -201, 810, 847)
    scaleout = np.float32(1) / np.float32(min(xmax - xmin, ymax - ymin))
    b = Builder()
    adv, kern = b.TextMetrics(TTF, "AV")
    assert kern == 0.0                                       # the font has neither a kern nor a GPOS table
    assert abs(adv / float(scaleout) - round(adv / float(scaleout))) < 1e-3 and 300 < adv / float(scaleout) < 900
    # spacing: the second glyph is translated by the first one's advance; a space adds its own advance; tab = 4 spaces
    def tx(text):
        t = b.TextLine(TTF, text).tree()
        root = t.nodes[t.root]
        kids = [t.nodes[t.links[root.link_off + k]] for k in range(root.nchild)]
        assert all(k.op == OP["TRANSLATE2D"] for k in kids)
        return [float(k.p[0]) for k in kids]
    a0, a1 = tx("AA")
    assert a0 == 0.0 and a1 == np.float32(adv)
    sp = tx("A A")[1] - a1
    assert sp > 0
    # a tab never reaches font.go:108's `advance *= 4`: unicode.IsGraphic('\t') is false, TextLine rejects it first
    with pytest.raises(ShapeError, match="not graphic"):
        b.TextLine(TTF, "A\tA")


def test_glyphs_are_cached_and_shared():
    b = Builder()
    t = b.TextLine(TTF, "ABAB").tree()
    root = t.nodes[t.root]
    kids = [t.nodes[t.links[root.link_off + k]] for k in range(root.nchild)]
    child = [t.links[k.link_off] for k in kids]
    assert child[0] == child[2] and child[1] == child[3] and child[0] != child[1]   # font.go:155-190 glyph cache


def test_errors_like_reference():
    b = Builder()
    with pytest.raises(ShapeError, match="no text provided"):
        b.TextLine(TTF, "   ")
    with pytest.raises(ShapeError, match="not graphic"):
        b.TextLine(TTF, "a\x01b")
    with pytest.raises(ShapeError, match="invalid RelativeGlyphTolerance"):
        b.TextLine(TTF, "A", reltol=1.0)
    with pytest.raises(ShapeError, match="sfnt"):
        b.TextLine(b"\x00\x01\x00\x00garbage", "A")
    assert b.TextLine(TTF, "Ø12 äöü").Bounds()[3] > 2.0       # multi-byte runes through the cmap
    with pytest.raises(ShapeError, match="glyph has no contours"):
        b.TextLine(TTF, "€")                                  # unmapped rune -> .notdef, which this font leaves empty (font.go:235)
    single = b.TextLine(TTF, "I")                            # one glyph: the translated glyph itself, no union
    assert b.op(single.id) == OP["TRANSLATE2D"]


def test_finer_tolerance_gives_more_vertices():
    b = Builder()
    coarse = b.TextLine(TTF, "O", reltol=0.3).tree()
    fine = b.TextLine(TTF, "O", reltol=0.01).tree()
    nv = lambda t: sum(t.nodes[i].aux_len for i in range(t.n_nodes) if t.nodes[i].op == OP["POLY2D"])
    assert nv(fine) > nv(coarse)
