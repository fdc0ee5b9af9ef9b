"""The division by a launch-uniform divisor the kernels use for pixel and sub-sample indices
(rtx_kernels.hpp: make_fast_div / fast_div, after Granlund & Montgomery 1994): the same arithmetic in
Python integers must give n // d for every 32-bit n and every divisor the host can hand it."""
import numpy as np

M32 = (1 << 32) - 1


def make_fast_div(d):
    l = 0
    while l < 32 and (1 << l) < d:
        l += 1
    mul = ((((1 << l) - d) << 32) // d + 1) & M32
    return mul, min(l, 1), max(l - 1, 0)


def fast_div(n, f):
    mul, s1, s2 = f
    t = (mul * n) >> 32
    return ((t + (((n - t) & M32) >> s1)) & M32) >> s2


def test_fast_division_is_exact():
    rng = np.random.default_rng(3)
    divisors = list(range(1, 130)) + [8 * w for w in (1, 3, 640, 1920, 3840, 100000)] + [n * n for n in range(1, 65)] + \
        [255, 256, 257, 65535, 65536, 65537, (1 << 31) - 1, 1 << 31, (1 << 31) + 1, M32 - 1, M32] + \
        [int(x) for x in rng.integers(1, 1 << 32, 300)]
    for d in divisors:
        f = make_fast_div(d)
        assert f[0] <= M32
        edge = [0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, M32 - 1, M32, M32 // d * d, max(0, M32 // d * d - 1)]
        for n in edge + [int(x) for x in rng.integers(0, 1 << 32, 200)]:
            if 0 <= n <= M32:
                assert fast_div(n, f) == n // d, (n, d)
