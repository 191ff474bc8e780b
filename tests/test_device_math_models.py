"""CPU models of two pieces of device arithmetic introduced in round 3 (apemost_amd/csrc/pt_device.h),
checked against 200-bit arithmetic -- what the GPU probes and parity tests show on the card
(profiles/r03_rcp_probe.txt, tests/test_gpu_parity.py), restated where no GPU is needed:

  * LogProdT: sum_i ln y_i as the logarithm of a running product split into mantissa and exponent
    (two points at a time, a renormalisation per eight points), one logarithm per lane, lanes added;
  * rcp_nr: 1 / b from a reciprocal good to 2^-24 by ONE third-order step of three FMAs.
"""
import math

import numpy as np
import pytest

mp = pytest.importorskip("mpmath")   # (comes with torch's sympy; the 200-bit reference)

mp.mp.prec = 200


def _fma(a, b, c):
    """a * b + c rounded once"""
    return float(mp.mpf(a) * mp.mpf(b) + mp.mpf(c))


def _log_product_lane(num, den):
    """one lane of Engine's data loop for a kLogProduct model with y = num / den: pairs of points, mul_ratio,
    renorm every eight points, finish<false> up to the logarithm itself"""
    m, md, e = 1.0, 1.0, 0
    for k in range(0, len(num) - 1, 2):
        pn, pd = num[k] * num[k + 1], den[k] * den[k + 1]
        fm, fe = math.frexp(pn)
        m, e = m * fm, e + fe
        fm, fe = math.frexp(pd)
        md, e = md * fm, e - fe
        if k % 8 == 6:
            m, fe = math.frexp(m)
            e += fe
            md, fe = math.frexp(md)
            e -= fe
    if len(num) % 2:
        fm, fe = math.frexp(num[-1])
        m, e = m * fm, e + fe
        fm, fe = math.frexp(den[-1])
        md, e = md * fm, e - fe
    return math.log(m / md) + e * math.log(2.0)


def test_sum_of_logarithms_as_the_logarithm_of_a_running_product():
    rs = np.random.RandomState(2)
    n_lanes, per_lane = 64, 1024                       # config 5: 65 536 points over a wavefront
    for spread in (1.0, 30.0, 120.0):                    # decades the numerators / denominators range over
        num = 10.0 ** rs.uniform(-spread, spread, (n_lanes, per_lane))
        den = 10.0 ** rs.uniform(-spread, spread, (n_lanes, per_lane))
        exact = sum(mp.log(mp.mpf(a)) - mp.log(mp.mpf(b)) for a, b in zip(num.ravel(), den.ravel()))
        lanes = [_log_product_lane(num[i], den[i]) for i in range(n_lanes)]
        got = math.fsum(lanes)
        # the reference's way: a logarithm per point, added serially
        serial = 0.0
        for a, b in zip(num.ravel(), den.ravel()):
            serial += math.log(a / b)
        scale = max(1.0, float(abs(exact)))
        err_prod, err_serial = float(abs(got - exact)), float(abs(serial - exact))
        assert err_prod <= 1e-13 * scale + n_lanes * per_lane * 1.2e-16, (spread, err_prod)
        assert err_prod <= 4 * err_serial + 1e-10, (spread, err_prod, err_serial)   # no worse than the sum it replaces


def test_third_order_reciprocal_step():
    rs = np.random.RandomState(3)
    worst = 0.0
    for _ in range(4000):
        b = math.ldexp(rs.uniform(1.0, 2.0), int(rs.randint(-300, 300))) * (1 if rs.uniform() < 0.5 else -1)
        r0 = float((1 / mp.mpf(b)) * (1 + mp.mpf(rs.uniform(-1, 1)) * mp.mpf(2) ** -24))   # v_rcp_f64: 2^-24 or better
        e = _fma(-b, r0, 1.0)
        t = _fma(e, e, e)
        r = _fma(r0, t, r0)
        ref = 1 / mp.mpf(b)
        ulp = math.ulp(float(ref))
        worst = max(worst, float(abs(mp.mpf(r) - ref)) / ulp)
    assert worst <= 0.75, worst
