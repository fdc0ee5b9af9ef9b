"""Counter-based RNG (Philox-4x32-10) that replaces rand() at the reference's call sites."""
import ctypes as C

import numpy as np

from raytracer_amd import device
from oracle import binding as orc


def _philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    orc.lib().orc_philox4x32_10(c, k, o)
    return list(o)


def test_philox_known_answers():
    # known-answer vectors of the published Philox4x32-10 (Random123 kat_vectors)
    assert _philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert _philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert _philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_uniform_range_and_library_agreement():
    rng = np.random.default_rng(0)
    vals = []
    for _ in range(2000):
        seed = int(rng.integers(0, 2**63))
        frame, pixel, sample, branch, draw = (int(v) for v in rng.integers(0, 2**32, size=5))
        a = orc.uniform(seed, frame, pixel, sample, branch, draw)
        b = device.uniform(seed, frame, pixel, sample, branch, draw)
        assert a == b
        vals.append(a)
    vals = np.array(vals)
    assert vals.min() >= 0.0 and vals.max() <= 1.0          # inclusive like rand()/RAND_MAX
    assert abs(vals.mean() - 0.5) < 0.03


def test_streams_are_distinct():
    base = [orc.uniform(0, 0, 10, 0, 0, d) for d in range(8)]
    assert len(set(base)) == 8
    assert base != [orc.uniform(0, 0, 11, 0, 0, d) for d in range(8)]
    assert base != [orc.uniform(0, 0, 10, 1, 0, d) for d in range(8)]
    assert base != [orc.uniform(0, 0, 10, 0, 1, d) for d in range(8)]
    assert base != [orc.uniform(0, 1, 10, 0, 0, d) for d in range(8)]
    assert base != [orc.uniform(1, 0, 10, 0, 0, d) for d in range(8)]
