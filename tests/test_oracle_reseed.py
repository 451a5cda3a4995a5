"""The `reseed` stream of the build (hod/abacus_hod.py:775-839 counterpart), CPU side: Philox4x32-10 of the oracle against
the generator's PUBLISHED known-answer vectors (Random123 kat_vectors, philox4x32 with 10 rounds), the fixed float64
log / sin / cos evaluations against libm, the distributions and dtypes the reference asks for, and invariance under
sharding.  The device side is held to this restatement bit for bit in tests/test_reseed_gpu.py."""
import math

import numpy as np

from oracle import oracle

# counter (4 words), key (2 words) -> output (4 words)
KAT = [
    ((0x00000000,) * 4, (0x00000000,) * 2, (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),      # digits of pi
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_philox_known_answer_vectors():
    for ctr, key, want in KAT:
        want = np.array(want, dtype=np.uint32)
        np.testing.assert_array_equal(oracle.philox4x32_10(ctr, key), want)       # the C restatement
        np.testing.assert_array_equal(oracle.philox4x32_10_py(ctr, key), want)    # straight from the published rounds


def test_fixed_log_sin_cos_agree_with_libm():
    rng = np.random.default_rng(1)
    for x in np.concatenate([rng.random(2000), 2.0 ** -rng.integers(0, 60, 200), [1.0, 2.0, 2.0**-24, 1.0 - 2.0**-24]]):
        assert abs(oracle.rs_log(float(x)) - math.log(x)) <= 5e-14 * max(1.0, abs(math.log(x)))   # series cut at s^15: 1e-14
    for t in np.concatenate([rng.random(2000), [0.0, 0.25, 0.5, 0.75, 1.0 - 2.0**-24]]):
        s, c = oracle.rs_sincos2pi(float(t))
        assert abs(s - math.sin(2 * math.pi * t)) < 1e-14 and abs(c - math.cos(2 * math.pi * t)) < 1e-14


def test_stream_distributions_dtypes_and_scaling():
    n = 400_000
    sig = 300.0 + np.arange(n) % 7
    hr, hv, pr = oracle.reseed(600, n, n + 3, hsigma3d=sig)
    for u in (hr, pr):                                   # float32 draws in [0, 1) (:780,819)
        assert u.min() >= 0.0 and u.max() < 1.0 and np.array_equal(u, u.astype(np.float32).astype(np.float64))
        assert abs(u.mean() - 0.5) < 4 / np.sqrt(12 * len(u)) and abs(u.var() - 1 / 12) < 1e-3
    g = hv * np.sqrt(3.0) / sig[:, None]                 # hveldev = r2 * hsigma3d / sqrt(3) (:826-833)
    assert abs(g.mean()) < 5 / np.sqrt(3 * n) and abs(g.var() - 1.0) < 1e-2
    assert abs(np.mean(np.abs(g) < 1.0) - 0.6826895) < 3e-3 and abs(np.mean(np.abs(g) > 3.0) - 0.0026998) < 3e-4
    assert abs(np.corrcoef(g[:, 0], g[:, 1])[0, 1]) < 5e-3 and abs(np.corrcoef(g[:, 0], g[:, 2])[0, 1]) < 5e-3
    _, he, _ = oracle.reseed(600, n, 0, hsigma3d=np.full(n, np.sqrt(3.0)), want_expvel=True)
    assert abs(he.mean()) < 0.01 and abs(np.mean(np.abs(he)) - 1.0) < 0.01 and abs(he.var() - 2.0) < 0.05   # Laplace(0, 1)
    a = oracle.reseed(601, 1000, 1000, hsigma3d=np.ones(1000))
    assert not np.array_equal(a[0], hr[:1000])           # another seed, another stream


def test_stream_is_invariant_under_sharding():
    sig = np.linspace(100, 400, 10_000)
    hr, hv, pr = oracle.reseed(77, 10_000, 10_001, hsigma3d=sig)
    for h0, p0 in ((0, 0), (3333, 2501), (9999, 6)):
        a, b, c = oracle.reseed(77, 10_000 - h0, 10_001 - p0, hsigma3d=sig[h0:], halo_index0=h0, part_index0=p0)
        np.testing.assert_array_equal(a, hr[h0:])
        np.testing.assert_array_equal(b, hv[h0:])
        np.testing.assert_array_equal(c, pr[p0:])
