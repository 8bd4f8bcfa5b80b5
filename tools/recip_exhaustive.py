#!/usr/bin/env python3
"""Is `v_rcp_f32 + Newton steps` the IEEE reciprocal?  Exhaustively, over all 2^23 fp32 significands (the exponent only shifts the result).

csrc/sn_device.h `sn_sample_q_exact` (r06) forms y = RN(1 / m) as
    y0 = v_rcp_f32(m)  (1 ulp);   y <- fma(fma(-m, y, 1), y, y)   twice
and substitutes the known quotient for an all-ones significand.  v_rcp_f32's own bits are not available here, so every fp32 within one ulp
of the correctly rounded reciprocal is tried as y0 (RN - 1 ulp, RN, RN + 1 ulp), with the two fmas emulated EXACTLY (float64 product of two
24-bit significands is exact; the sum is rounded to fp32 once, by a TwoSum-corrected float64 addition).  Result (printed, and asserted by
tests/test_recip_division.py):
    one Newton step : RN - 1, RN -> only the all-ones significand differs;  RN + 1 -> 32 significands differ
    two Newton steps: every starting value -> only the all-ones significand may differ
which is why the kernel takes two steps and handles the all-ones significand by substitution (2^-(e+1) (1 + 2^-23), bits 0x7F000000 - bits(m)).
The hardware's own v_rcp_f32 is exercised over every significand by tests/test_gpu_stages.py.
"""
import numpy as np

f32, f64 = np.float32, np.float64


def fma32(a, b, c):
    """RN32(a * b + c) for fp32 arrays, exactly: the product of two 24-bit significands is exact in float64; the addition is corrected by its
    TwoSum error term where the float64 sum sits exactly on an fp32 rounding boundary (the only place a second rounding could differ)."""
    p = a.astype(f64) * b.astype(f64)
    c = c.astype(f64)
    s = p + c
    bb = s - p
    err = (p - (s - bb)) + (c - bb)
    y = s.astype(f32)
    yd = y.astype(f64)
    up = np.nextafter(y, f32(np.inf)).astype(f64)
    dn = np.nextafter(y, f32(-np.inf)).astype(f64)
    y = np.where(((s - yd) == (up - yd) / 2) & (err > 0), up.astype(f32), y)
    y = np.where(((yd - s) == (yd - dn) / 2) & (err < 0), dn.astype(f32), y)
    return y


def ulp_shift(x, d):
    return (x.view(np.int32) + np.int32(d)).view(f32)


def mismatches(steps: int, start_offset_ulps: int):
    """Significands (as integers) for which `steps` Newton steps from RN(1 / m) + start_offset_ulps do not return RN(1 / m)."""
    man = np.arange(1 << 23, dtype=np.uint32)
    m = (man | np.uint32(0x3F800000)).view(f32)                    # [1, 2)
    rn = (f64(1.0) / m.astype(f64)).astype(f32)                    # RN(1 / m): verified minimal-residual below
    M = man.astype(np.int64) | (1 << 23)

    def resid(r):
        return np.abs((1 << 47) - M * (r.astype(f64) * 2.0 ** 24).astype(np.int64))

    assert bool(((resid(rn) <= resid(np.nextafter(rn, f32(2)))) & (resid(rn) <= resid(np.nextafter(rn, f32(0))))).all())
    one = np.ones_like(m)
    y = ulp_shift(rn, start_offset_ulps)
    for _ in range(steps):
        y = fma32(fma32(-m, y, one), y, y)
    return man[y != rn]


def main():
    for steps in (1, 2):
        for d in (-1, 0, 1):
            bad = mismatches(steps, d)
            print(f"{steps} Newton step(s) from RN{d:+d} ulp: {bad.size} of 8388608 significands differ from RN(1/m)"
                  + (": " + " ".join(hex(int(b)) for b in bad[:8]) + (" ..." if bad.size > 8 else "") if bad.size else ""))
    # the substitution for the all-ones significand
    for e in range(-3, 12):
        m = np.array([np.float32(2.0 - 2.0 ** -23) * np.float32(2.0 ** e)], dtype=f32)
        want = (f64(1.0) / m.astype(f64)).astype(f32)
        got = (np.uint32(0x7F000000) - m.view(np.uint32)).view(f32)
        assert got[0] == want[0], (e, got, want)
    print("all-ones significand: RN(1 / m) == bits 0x7F000000 - bits(m) for every exponent tried")


if __name__ == "__main__":
    main()
