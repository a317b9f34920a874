"""include/rayn_detmath.h (host side, through the oracle's probe): within 1 ulp of a double-precision
reference and identical under both FMA policies (the pinned functions never use mul_add)."""
import math

import numpy as np
import pytest


def _ulp_diff(a, b):
    ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


@pytest.mark.parametrize("op,fn,lo,hi", [
    (0, np.exp, -40.0, 10.0), (1, np.sin, -20.0, 20.0), (2, np.cos, -20.0, 20.0), (3, np.tan, -1.55, 1.55)])
def test_unary(oracle, op, fn, lo, hi):
    x = np.random.default_rng(op).uniform(lo, hi, 100000).astype(np.float32)
    got = oracle.detmath(op, x)
    want = fn(x.astype(np.float64)).astype(np.float32)
    assert _ulp_diff(got, want).max() <= 1
    assert np.array_equal(got, oracle.detmath(op, x, fma=True))


def test_atan2_pow(oracle):
    rng = np.random.default_rng(9)
    y, x = rng.uniform(-5, 5, 100000).astype(np.float32), rng.uniform(-5, 5, 100000).astype(np.float32)
    assert _ulp_diff(oracle.detmath(4, y, x), np.arctan2(y.astype(np.float64), x).astype(np.float32)).max() <= 1
    b, e = rng.uniform(0, 1, 100000).astype(np.float32), rng.uniform(0.05, 12, 100000).astype(np.float32)
    assert _ulp_diff(oracle.detmath(5, b, e), np.power(b.astype(np.float64), e).astype(np.float32)).max() <= 1


def test_special_values(oracle):
    f = lambda op, a, b=0.0: float(oracle.detmath(op, np.array([a], np.float32), np.array([b], np.float32))[0])
    assert f(0, 0.0) == 1.0 and f(0, -200.0) == 0.0 and math.isinf(f(0, 100.0)) and math.isnan(f(0, float("nan")))
    assert f(1, 0.0) == 0.0 and f(2, 0.0) == 1.0 and math.isnan(f(1, float("inf")))
    assert f(4, 0.0, -1.0) == np.float32(math.pi) and f(4, -0.0, -1.0) == -np.float32(math.pi) and f(4, 1.0, 0.0) == np.float32(math.pi / 2)
    assert f(5, 0.0, 2.0) == 0.0 and f(5, 1.0, 7.0) == 1.0 and f(5, 0.5, 0.0) == 1.0 and math.isnan(f(5, -1.0, 0.5))
