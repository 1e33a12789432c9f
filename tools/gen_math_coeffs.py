#!/usr/bin/env python3
"""Regenerates the fitted coefficients used in csrc/rl_math.h (needs mpmath).  Taylor coefficients
(sin, cos, exp, log) are 1/n! resp. 1/(2n+1) and need no generator; only the asin core is a fit."""
import mpmath as mp

mp.mp.dps = 40


def f(z):
    z = mp.mpf(z)
    if z == 0:
        return mp.mpf(1) / 6
    y = mp.sqrt(z)
    return (mp.asin(y) / y - 1) / z


poly, err = mp.chebyfit(f, [0, mp.mpf("0.2501")], 10, error=True)
print("asin core, degree 9 in z on [0, 0.25], max |error| =", mp.nstr(err, 5))
for c in poly:  # highest degree first, Horner order
    print(repr(float(c)))
