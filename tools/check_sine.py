#!/usr/bin/env python3
"""Accuracy of the device sine (pt_device.h: sin_cw, two-term Cody-Waite reduction modulo pi + one
odd polynomial on [-pi/2, pi/2]) against 200-bit arithmetic, operation by operation as the kernel
rounds them (every FMA rounds once).  Reports the absolute error (what a likelihood sees: it adds
a sin() to numbers of order one) and the relative error in ulp, for the kernel's two-term
reduction and for a three-term one (which keeps the relative error near the zeros of the sine at
the price of one more FMA per sine), and compares the two evaluation orders of the polynomial:
Horner (shortest instruction count) and Estrin (shortest dependent chain).

    python tools/check_sine.py [samples]"""
import random
import sys

import mpmath as mp

mp.mp.prec = 200

INV_PI = 3.18309886183790691216e-01
NPI_HI, NPI_MID, NPI_LO = -3.14159265358979311600e+00, -1.22464679914735320717e-16, 2.99476980971833966589e-33
MAGIC = 6755399441055744.0
S = [-1.66666666666666657415e-01, 8.33333333333331587045e-03, -1.98412698412549659988e-04, 2.75573192191608328544e-06,
     -2.50521076166904495328e-08, 1.60589772926427431318e-10, -7.64396966398807388923e-13, 2.73143687693798929796e-15]


def fma(a, b, c):
    return float(mp.mpf(a) * mp.mpf(b) + mp.mpf(c))


def reduce_(x, terms=2):
    fm = fma(x, INV_PI, MAGIC)
    fn = fm - MAGIC
    r = fma(fn, NPI_HI, x)
    r = fma(fn, NPI_MID, r)
    if terms == 3:
        r = fma(fn, NPI_LO, r)
    return r, int(fn) & 1


def horner(x, terms=2):
    r, odd = reduce_(x, terms)
    z = r * r
    q = fma(z, S[7], S[6])
    for k in (5, 4, 3, 2, 1, 0):
        q = fma(z, q, S[k])
    v = fma(r * z, q, r)
    return -v if odd else v


def estrin(x, terms=2):
    r, odd = reduce_(x, terms)
    z = r * r
    z2 = z * z
    p01, p23 = fma(z, S[1], S[0]), fma(z, S[3], S[2])
    p45, p67 = fma(z, S[5], S[4]), fma(z, S[7], S[6])
    z4 = z2 * z2
    lo, hi = fma(z2, p23, p01), fma(z2, p67, p45)
    q = fma(z4, hi, lo)
    v = fma(r * z, q, r)
    return -v if odd else v


def ulp_err(got, x):
    ref = mp.sin(mp.mpf(x))
    if ref == 0:
        return 0.0
    e = mp.frexp(ref)[1]
    return float(abs(mp.mpf(got) - ref) / mp.ldexp(1, e - 53))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    rnd = random.Random(1)
    worst = {"abs2": 0.0, "ulp2": 0.0, "ulp2_away": 0.0, "ulp3": 0.0, "estrin_abs2": 0.0, "estrin_ulp3": 0.0}
    for i in range(n):
        kind = i % 5
        if kind == 0:
            x = rnd.uniform(-10, 10)
        elif kind == 1:
            x = rnd.uniform(-1e4, 1e4)
        elif kind == 2:
            x = rnd.choice((-1, 1)) * 10 ** rnd.uniform(0, 13.5)
        elif kind == 3:  # neighbourhoods of k pi / 2
            x = float(mp.mpf(rnd.randrange(-100000, 100000)) * mp.pi / 2) + rnd.uniform(-1e-6, 1e-6)
        else:  # the doubles nearest to k pi: the worst cancellation the reduction can meet
            x = float(mp.mpf(rnd.randrange(1, 10 ** rnd.randrange(1, 13))) * mp.pi)
        ref = mp.sin(mp.mpf(x))
        h2, h3 = horner(x, 2), horner(x, 3)
        worst["abs2"] = max(worst["abs2"], float(abs(mp.mpf(h2) - ref)))
        worst["ulp2"] = max(worst["ulp2"], ulp_err(h2, x))
        if abs(ref) > 1e-6:
            worst["ulp2_away"] = max(worst["ulp2_away"], ulp_err(h2, x))
        worst["ulp3"] = max(worst["ulp3"], ulp_err(h3, x))
        worst["estrin_abs2"] = max(worst["estrin_abs2"], float(abs(mp.mpf(estrin(x, 2)) - ref)))
        worst["estrin_ulp3"] = max(worst["estrin_ulp3"], ulp_err(estrin(x, 3), x))
    print("max error over %d samples, |x| < 3e13" % n)
    print("  kernel (two-term reduction, Horner): absolute %.3g; relative %.3g ulp where |sin| > 1e-6, %.3g ulp anywhere"
          % (worst["abs2"], worst["ulp2_away"], worst["ulp2"]))
    print("  three-term reduction, Horner: %.3g ulp anywhere" % worst["ulp3"])
    print("  Estrin: absolute %.3g (two-term), %.3g ulp (three-term)" % (worst["estrin_abs2"], worst["estrin_ulp3"]))


if __name__ == "__main__":
    main()
