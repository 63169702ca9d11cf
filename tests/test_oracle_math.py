"""The math32 restatement (oracle/orc_math.h) against float64 references: wrappers must be the
correctly rounded float32 of the float64 function (chewxy/math32 doc.go), float32-native ports
must stay within their algorithm's error bound."""
import numpy as np

from oracle.oracle import math_apply

RNG = np.random.default_rng(3)


def ulps(a, b):
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, np.int64(-2**31) - ia, ia)
    ib = np.where(ib < 0, np.int64(-2**31) - ib, ib)
    return np.abs(ia - ib)


def test_wrappers_are_rounded_float64():
    x = (RNG.random(20000, np.float32) * 40 - 20).astype(np.float32)
    y = (RNG.random(20000, np.float32) * 40 - 20).astype(np.float32)
    x64, y64 = x.astype(np.float64), y.astype(np.float64)
    assert ulps(math_apply("atan2", y, x), np.arctan2(y64, x64).astype(np.float32)).max() == 0
    assert ulps(math_apply("sin", x), np.sin(x64).astype(np.float32)).max() == 0
    assert ulps(math_apply("cos", x), np.cos(x64).astype(np.float32)).max() == 0
    assert ulps(math_apply("cbrt", x), np.cbrt(x64).astype(np.float32)).max() == 0
    u = (RNG.random(20000, np.float32) * 2 - 1).astype(np.float32)
    assert ulps(math_apply("acos", u), np.arccos(u.astype(np.float64)).astype(np.float32)).max() == 0


def test_atan2_special_cases():
    y = np.float32([0.0, -0.0, 0.0, -0.0, 1.0, -1.0, 0.0, 0.0])
    x = np.float32([1.0, 1.0, -1.0, -1.0, 0.0, 0.0, 0.0, -0.0])
    got = math_apply("atan2", y, x)
    want = np.arctan2(y.astype(np.float64), x.astype(np.float64)).astype(np.float32)
    assert (got.view(np.uint32) == want.view(np.uint32)).all()


def test_hypot_float32_port():
    p = (RNG.standard_normal(20000) * 10).astype(np.float32)
    q = (RNG.standard_normal(20000) * 10).astype(np.float32)
    ref = np.hypot(p.astype(np.float64), q.astype(np.float64)).astype(np.float32)
    assert ulps(math_apply("hypot", p, q), ref).max() <= 2
    z = np.float32([0.0, 3.0, 0.0, -4.0])
    w = np.float32([0.0, 4.0, 5.0, 3.0])
    np.testing.assert_array_equal(math_apply("hypot", z, w), np.float32([0, 5, 5, 5]))


def test_sincos_float32_port():
    x = (RNG.random(20000, np.float32) * 14 - 7).astype(np.float32)
    s, c = math_apply("sincos_s", x), math_apply("sincos_c", x)
    assert np.abs(s - np.sin(x.astype(np.float64))).max() < 3e-7
    assert np.abs(c - np.cos(x.astype(np.float64))).max() < 3e-7
    z = np.float32([0.0, -0.0])
    assert (math_apply("sincos_s", z).view(np.uint32) == z.view(np.uint32)).all()  # Sincos(+-0) = +-0, 1
    np.testing.assert_array_equal(math_apply("sincos_c", z), np.float32([1, 1]))


def test_min_max_go_semantics():
    a = np.float32([0.0, -0.0, 1.0, np.nan, -np.inf, 2.0])
    b = np.float32([-0.0, 0.0, np.nan, 1.0, np.nan, 3.0])
    mn = math_apply("min", a, b)
    mx = math_apply("max", a, b)
    assert np.signbit(mn[0]) and np.signbit(mn[1])          # Min(+-0, -+0) = -0
    assert not np.signbit(mx[0]) and not np.signbit(mx[1])  # Max(+-0, -+0) = +0
    assert np.isnan(mn[2]) and np.isnan(mn[3]) and np.isnan(mx[2]) and np.isnan(mx[3])
    assert mn[4] == -np.inf                                  # Min(-Inf, NaN) = -Inf
    assert mn[5] == 2 and mx[5] == 3


def test_round_floor():
    x = np.float32([0.5, 1.5, 2.5, -0.5, -1.5, 0.49999997, -2.5, 3.2, -3.7])
    np.testing.assert_array_equal(math_apply("round", x), np.float32([1, 2, 3, -1, -2, 0, -3, 3, -4]))  # half away from zero
    np.testing.assert_array_equal(math_apply("floor", x), np.floor(x))


def test_pow13():
    x = (RNG.random(5000, np.float32) * 10).astype(np.float32)
    ref = np.cbrt(x.astype(np.float64))
    assert np.abs(math_apply("pow13", x) - ref).max() < 1e-6 * 3
