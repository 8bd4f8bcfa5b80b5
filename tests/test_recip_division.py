"""The fused resampler's division (csrc/sn_proposal.h, sn_pdf_lane RECIP): weight / sum from the correctly rounded reciprocal of the
loop-invariant sum plus one exact residual step,
    y = RN(1 / d);  q0 = RN(n y);  e = n - q0 d (one fma, exact);  q = RN(q0 + e y)
is claimed to BE the IEEE quotient RN(n / d) unless the significand of d is all ones (such a ray sends its wave down the plain division).
ADVICE r02: q0 may be 1.5 ulp off when n / d sits just below a power of two, so Markstein's theorem does not apply verbatim.  The claim
is therefore checked here as arithmetic, with the three fp32 instructions emulated exactly in float64 (a 24 x 24-bit product is exact,
the residual cancels exactly): random operands over the resampler's range and DIRECTED ones whose quotient lies within 64 ulps below a
power of two.  (r03 ran the same script over 80 M random + 400 M directed pairs: 0 mismatches.)  The GPU counterpart, which runs the real
instructions, is tests/test_gpu_render.py::test_resampler_reciprocal_division_is_bit_identical."""
import numpy as np

f32, f64 = np.float32, np.float64


def _fma32(a, b, c):
    return (a.astype(f64) * b.astype(f64) + c.astype(f64)).astype(f32)


def _recip_quotient(num, den):
    y = (f32(1.0) / den).astype(f32)
    q0 = (num * y).astype(f32)
    return _fma32(_fma32(-q0, den, num), y, q0)


def _mismatches(num, den):
    ok_den = (den.view(np.uint32) & 0x7FFFFF) != 0x7FFFFF          # the kernel's guard
    return int(((_recip_quotient(num, den) != (num / den).astype(f32)) & ok_den).sum())


def test_random_operands_over_the_resamplers_range():
    rng = np.random.default_rng(0)
    for it in range(4):
        n = 1_000_000
        den = rng.uniform(0.4, 4.0, n).astype(f32) if it % 2 == 0 else np.exp(rng.uniform(np.log(1e-5), np.log(300.0), n)).astype(f32)
        num = (rng.uniform(0, 1, n) ** 3 * np.minimum(den, 1.0) + 0.01).astype(f32)   # (w + 0.01) + padding, w in [0, 1]
        assert _mismatches(num, den) == 0


def test_quotients_just_below_a_power_of_two():
    rng = np.random.default_rng(1)
    n = 1_000_000
    for _ in range(2):
        den = rng.uniform(0.4, 4.0, n).astype(f32)
        target = np.ldexp(1.0, rng.integers(-6, 1, n)) * (1.0 - rng.integers(1, 64, n) * 2.0**-24)
        num = (target * den.astype(f64)).astype(f32)
        for d in (0, 1, -1, 2, -2):
            assert _mismatches((num.view(np.int32) + d).view(f32), den) == 0


def test_all_ones_denominators_hold_too():
    """The kernel sends a wave with an all-ones denominator significand down the plain divisions (Markstein's condition for a reciprocal
    refined by Newton steps).  Here y comes from an IEEE division, and the identity holds for those denominators as well: the guard is
    conservative, not load-bearing."""
    den = (np.float32(2.0) - np.float32(2.0**-23)) * np.ldexp(1.0, np.arange(1 << 20) % 8 - 4).astype(f32)
    assert ((den.view(np.uint32) & 0x7FFFFF) == 0x7FFFFF).all()
    num = np.random.default_rng(2).uniform(0.01, 1.0, den.size).astype(f32)
    assert int((_recip_quotient(num, den) != (num / den).astype(f32)).sum()) == 0


# ---- r06: the reciprocal of the main kernel's exact position map (csrc/sn_device.h sn_sample_q_exact) ----------------------------------
def test_two_newton_steps_give_the_ieee_reciprocal_for_every_significand():
    """tools/recip_exhaustive.py, exhaustively over all 2^23 significands and every start within one ulp of RN(1 / m): two Newton steps
    return RN(1 / m) except (possibly) for the all-ones significand, which the kernel substitutes; ONE step would not be enough."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import recip_exhaustive as rx

    for d in (-1, 0, 1):
        bad = rx.mismatches(2, d)
        assert set(int(b) for b in bad) <= {0x7FFFFF}, (d, bad[:8])
    assert rx.mismatches(1, 1).size > 1            # why the kernel takes two
    for e in range(-3, 12):                        # the substitution itself
        m = np.array([np.float32(2.0 - 2.0 ** -23) * np.float32(2.0 ** e)], dtype=f32)
        assert (np.uint32(0x7F000000) - m.view(np.uint32)).view(f32)[0] == (f64(1.0) / m.astype(f64)).astype(f32)[0]


def test_quotients_of_the_contraction_from_the_exact_reciprocal():
    """p / m for the contraction's operands: m = max |coordinate| in [1, 2000], |p| <= m (quotients in [-1, 1], many of them just below 1 or
    just below a power of two), by q0 = RN(p y), q = RN(q0 + (p - q0 m) y) with y = RN(1 / m)."""
    rng = np.random.default_rng(3)
    n = 2_000_000
    m = np.exp(rng.uniform(0.0, np.log(2000.0), n)).astype(f32)
    for k in range(3):
        p = (m * (rng.uniform(0, 1, n) ** (1 + 2 * k))).astype(f32)
        assert _mismatches(p, m) == 0 and _mismatches(-p, m) == 0
    for _ in range(2):                              # within 64 ulp of the norm, and the same scaled by 2^-k
        j = rng.integers(0, 64, n).astype(np.int32)
        p = (m.view(np.int32) - j).view(f32) * np.ldexp(1.0, -rng.integers(0, 8, n)).astype(f32)
        assert _mismatches(p, m) == 0
    assert _mismatches(m.copy(), m) == 0            # the coordinate that IS the norm: exactly 1
