"""The single-precision child-box test of the BVH walk (rtx_kernels.hpp, SlabF32) may enter a box
needlessly but must never refuse one the FP64 slab test enters.  A numpy restatement of its
arithmetic (float32 operands, the fused multiply-add formed in float64 and rounded once) is
compared with the FP64 test on boxes and rays chosen to stress the error bound: origins far from
small boxes, rays nearly parallel to an axis, flat boxes, origins inside and on the faces."""
import numpy as np

F = np.float32


def not_below(x):
    f = F(x)
    return np.nextafter(f, F(np.inf)) if float(f) < x else f


def slab_f64(lo, hi, o, inv):
    t0, t1 = (lo - o) * inv, (hi - o) * inv
    return np.maximum.reduce(np.minimum(t0, t1)), np.minimum.reduce(np.maximum(t0, t1))


def slab_f32(lo, hi, o, inv, scale):
    """lo, hi: float32[3]; o, inv: float64[3].  Returns the widened (tn, tf) as the kernel forms them."""
    s = not_below(scale * 1.001)
    tn, tf = F(-np.inf), F(np.inf)
    for k in range(3):
        if abs(inv[k]) <= 2.0 ** 60 and abs(o[k] * inv[k]) <= 2.0 ** 100:
            i = F(inv[k])
            c = F(-(o[k] * inv[k]))
            reach = F(np.float64(abs(i)) * np.float64(s) + np.float64(abs(c)))       # one fused multiply-add
            delta = F(F(2.0 ** -22) * F(reach * F(1.000001))) + F(1e-30)
            cn, cf = F(c - delta), F(c + delta)
        else:
            i = F(1.0 if inv[k] >= 0 else -1.0)
            cn, cf = F(-np.inf), F(np.inf)
        # the constant that goes with the lower plane, and the one that goes with the upper plane
        a, b = (cn, cf) if inv[k] >= 0 else (cf, cn)
        fma = lambda p, m, c: F(np.float64(p) * np.float64(m) + np.float64(c))
        t0, t1 = fma(lo[k], i, a), fma(hi[k], i, b)
        tn = max(tn, min(t0, t1))
        tf = min(tf, max(t0, t1))
    return tn, tf


def cases(rng, n):
    for _ in range(n):
        kind = rng.integers(0, 6)
        size = 10.0 ** rng.uniform(-4, 1)
        centre = rng.uniform(-5, 5, 3)
        half = rng.uniform(0.0, size, 3)
        if kind == 1:
            half[rng.integers(0, 3)] = 0.0                      # flat box
        lo64, hi64 = centre - half, centre + half
        lo = np.array([np.nextafter(F(v), F(-np.inf)) if float(F(v)) > v else F(v) for v in lo64], dtype=F)
        hi = np.array([np.nextafter(F(v), F(np.inf)) if float(F(v)) < v else F(v) for v in hi64], dtype=F)
        if kind == 2:
            o = rng.uniform(lo64, hi64 + 1e-300)                 # origin inside
        elif kind == 3:
            o = centre + rng.uniform(-1, 1, 3) * 10.0 ** rng.uniform(0, 3)   # far away
        else:
            o = centre + rng.uniform(-3, 3, 3)
        if kind == 4:
            o[rng.integers(0, 3)] = float(lo[rng.integers(0, 3)])            # on a face plane
        target = rng.uniform(lo64, hi64 + 1e-300) + rng.normal(0, 1e-3 * size, 3) * rng.integers(0, 2)
        d = target - o
        if kind == 5:
            d[rng.integers(0, 3)] *= 10.0 ** rng.uniform(-12, -3)             # nearly parallel to an axis
        nrm = np.sqrt((d * d).sum())
        if not nrm > 0:
            continue
        d = d / nrm
        if np.any(d == 0.0):
            continue
        # scene_scale bounds every plane of the tree: the tightest value it can have, or a scene with a far floor
        tight = float(max(np.abs(lo).max(), np.abs(hi).max()))
        yield lo, hi, o, 1.0 / d, tight if rng.integers(0, 2) else 1000.0


def test_f32_box_test_never_refuses_what_the_f64_test_enters():
    rng = np.random.default_rng(2024)
    entered = looser = 0
    for lo, hi, o, inv, scale in cases(rng, 40000):
        tn64, tf64 = slab_f64(lo.astype(np.float64), hi.astype(np.float64), o, inv)
        tn32, tf32 = slab_f32(lo, hi, o, inv, scale)
        # the interval of the single-precision test contains the FP64 one
        assert float(tn32) <= tn64 and float(tf32) >= tf64, (lo, hi, o, inv, tn32, tn64, tf32, tf64)
        ok64 = tn64 <= tf64 and tf64 >= 0.0
        ok32 = tn32 <= tf32 and tf32 >= F(0.0)
        assert ok32 or not ok64
        entered += ok64
        looser += ok32 and not ok64
    assert entered > 5000          # the cases do hit their boxes
    assert looser < entered // 20  # and the widening stays small


def test_f32_bound_is_not_below_the_f64_one():
    rng = np.random.default_rng(5)
    for x in np.concatenate([rng.uniform(-1e6, 1e6, 1000), 10.0 ** rng.uniform(-30, 30, 1000), [0.0, np.inf]]):
        assert float(not_below(x)) >= x


def half_outwards(v, up):
    """The half-precision value not inside (v, v): what rtx_bvh.hpp: half_bits stores for a plane."""
    h = np.float16(v)
    if (up and float(h) < float(v)) or (not up and float(h) > float(v)):
        h = np.nextafter(h, np.float16(np.inf if up else -np.inf))
    return h


def test_half_precision_planes_never_refuse_what_the_f64_test_enters():
    """BvhNode16: the same arithmetic on planes rounded outwards once more, to half precision (a
    half converts to single precision exactly and v_fma_mix_f32 rounds once, like the fused
    multiply-add above): the interval still contains the FP64 one of the single-precision box."""
    rng = np.random.default_rng(77)
    entered = 0
    for lo, hi, o, inv, scale in cases(rng, 20000):
        lo16 = np.array([F(half_outwards(v, False)) for v in lo], dtype=F)
        hi16 = np.array([F(half_outwards(v, True)) for v in hi], dtype=F)
        assert np.all(lo16 <= lo) and np.all(hi16 >= hi)
        tn64, tf64 = slab_f64(lo.astype(np.float64), hi.astype(np.float64), o, inv)
        # |plane| <= scene_scale is what the widening assumes: the half-precision planes of a tree stay
        # inside the scene box rounded outwards, 2^-10 beyond it at most
        scale16 = max(scale, float(max(np.abs(lo16).max(), np.abs(hi16).max())))
        tn16, tf16 = slab_f32(lo16, hi16, o, inv, scale16)
        ok64 = tn64 <= tf64 and tf64 >= 0.0
        ok16 = tn16 <= tf16 and tf16 >= F(0.0)
        assert ok16 or not ok64, (lo, hi, o, inv)
        entered += ok64
    assert entered > 2500
