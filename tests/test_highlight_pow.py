"""The highlight's pow (raytrace.comp:357, toon.glsl:33) as the HIP kernel evaluates it, restated in numpy.

The kernel computes x^y = 2^(y log2 x) in double precision (log2 through s = (m-1)/(m+1) and a degree-7
polynomial in s^2, 2^f through a degree-10 polynomial) and rounds once to float32; the CPU restatement of the
shader calls libm's powf.  Both should be the correctly rounded result almost always: here the formula is
checked against an 80-bit pow rounded to float32 and against libm, over the highlight's domain
(x in (0, 1], shininess up to 256) and beyond it.
"""
import ctypes
import math

import numpy as np

LN2 = math.log(2.0)
LOG_COEFF = [2.0 / LN2 / (2 * k + 1) for k in range(8)]
EXP_COEFF = [LN2 ** k / math.factorial(k) for k in range(11)]


def highlight_pow(x, y):
    m, e = np.frexp(x.astype(np.float64))
    low = m < 0.70710678118654752
    m, e = np.where(low, m * 2, m), np.where(low, e - 1, e)
    s = (m - 1.0) / (m + 1.0)
    s2 = s * s
    p = np.full_like(s, LOG_COEFF[7])
    for k in range(6, -1, -1):
        p = p * s2 + LOG_COEFF[k]
    t = np.clip(y.astype(np.float64) * (s * p + e), -1100, 1100)
    n = np.rint(t)
    f = t - n
    q = np.full_like(f, EXP_COEFF[10])
    for k in range(9, -1, -1):
        q = q * f + EXP_COEFF[k]
    return np.ldexp(q, n.astype(np.int64)).astype(np.float32)


def _ulps(a, b):
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


def test_formula_is_the_correctly_rounded_power():
    rng = np.random.default_rng(1)
    n = 200_000
    x = np.concatenate([rng.random(n), 1 - rng.random(n) * 1e-3, rng.random(n) * 1e-6, rng.uniform(1, 50, n)]).astype(np.float32)
    y = np.concatenate([rng.choice([8, 16, 32, 64, 128, 10.5, 3.3], n), rng.uniform(0.1, 256, n), rng.uniform(0.1, 40, n),
                        rng.uniform(-20, 20, n)]).astype(np.float32)
    x = np.maximum(x, np.float32(1e-38))
    mine = highlight_pow(x, y)
    with np.errstate(over="ignore", under="ignore"):
        ref = np.power(x.astype(np.longdouble), y.astype(np.longdouble)).astype(np.float32)
    ok = np.isfinite(ref) & (ref >= np.float32(1.2e-38))  # float32 normal range
    assert int(_ulps(mine[ok], ref[ok]).max()) <= 1
    assert float((mine[ok] != ref[ok]).mean()) < 1e-4


def test_formula_agrees_with_libm_powf():
    libm = ctypes.CDLL("libm.so.6")
    libm.powf.restype = ctypes.c_float
    libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
    rng = np.random.default_rng(2)
    n = 30_000
    x = rng.random(n).astype(np.float32)
    x[x == 0] = 1.0
    y = rng.choice([8.0, 16.0, 32.0, 64.0, 100.0, 7.25], n).astype(np.float32)
    g = np.array([libm.powf(float(a), float(b)) for a, b in zip(x, y)], dtype=np.float32)
    mine = highlight_pow(x, y)
    ok = g >= np.float32(1.2e-38)
    assert int(_ulps(mine[ok], g[ok]).max()) <= 1
    assert float((mine[ok] != g[ok]).mean()) < 5e-3


def whole_power(x, n):
    """The kernel's route for whole exponents: square and multiply in double precision, one rounding to float32."""
    b = x.astype(np.float64)
    r = np.ones_like(b)
    n = n.astype(np.int64).copy()
    with np.errstate(over="ignore", under="ignore"):
        while np.any(n > 1):
            odd = (n > 1) & ((n & 1) == 1)
            r = np.where(odd, r * b, r)
            b = np.where(n > 1, b * b, b)
            n = np.where(n > 1, n >> 1, n)
        return (r * b).astype(np.float32)


def test_square_and_multiply_is_the_correctly_rounded_power_for_whole_exponents():
    rng = np.random.default_rng(3)
    n = 300_000
    x = np.concatenate([rng.random(n), 1 - rng.random(n) * 1e-3, rng.uniform(1, 3, n)]).astype(np.float32)
    x = np.maximum(x, np.float32(1e-38))
    y = np.concatenate([rng.choice([1, 2, 3, 16, 32, 128, 4096], n), rng.integers(1, 4097, n), rng.integers(1, 200, n)]).astype(np.float32)
    mine = whole_power(x, y)
    with np.errstate(over="ignore", under="ignore"):
        ref = np.power(x.astype(np.longdouble), y.astype(np.longdouble)).astype(np.float32)
    ok = np.isfinite(ref)
    # also where the result is a float32 denormal or underflows to 0: the double-precision product is rounded once
    assert np.array_equal(mine[ok], ref[ok])
