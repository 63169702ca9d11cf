#!/usr/bin/env python3
"""Coefficients of dm::atan2_fast (gsdf_amd/csrc/dev_math.h): atan(t) = t * P(t^2) on t in [0, 1], P a polynomial of degree N in
u = t^2 fitted to atan(sqrt u) / sqrt u on [0, 1] (Chebyshev fit, mpmath, 80 digits), coefficients rounded to binary64. Prints the
largest RELATIVE error of the rounded polynomial against atan over a dense sample (exact arithmetic for the evaluation: the
kernel's Horner steps are fp64 FMAs, 2^-53 each) and the C initialiser."""
import sys
import mpmath as mp
mp.mp.dps = 80

def f(u):
    if u == 0:
        return mp.mpf(1)
    s = mp.sqrt(u)
    return mp.atan(s) / s

def fit(N):
    c = mp.chebyfit(f, [0, 1], N + 1)          # N + 1 coefficients, highest power first
    c = [mp.mpf(float(x)) for x in c]          # round to binary64
    worst = mp.mpf(0)
    M = 20000
    for i in range(M + 1):
        t = mp.mpf(i) / M
        u = t * t
        p = mp.polyval(c, u)
        ref = f(u)
        worst = max(worst, abs(p - ref) / ref)
    return c, worst

if __name__ == "__main__":
    for N in ([int(a) for a in sys.argv[1:]] or range(12, 20)):
        c, w = fit(N)
        print(f"degree {N}: max rel err 2^{float(mp.log(w, 2)):.2f}")
        if len(sys.argv) > 1:
            print("  {" + ", ".join(float(x).hex() for x in c) + "}   /* highest power of u first */")
