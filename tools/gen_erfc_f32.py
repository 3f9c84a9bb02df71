#!/usr/bin/env python3
"""Coefficients of the Float32 erfc with RELATIVE accuracy used by the ARG-2000 activation kernels (csrc/cmx_arg_kernels.hip erfc_rel_dev<float>):

    erfc(x) = t P(t) exp(-x^2),   t = 1/(1 + p x),   x >= 0,      erfc(-x) = 2 - erfc(x)

P of degree D is the minimax fit (Remez exchange, mpmath) of erfcx(x)/t in the RELATIVE sense over 0 <= x <= 10 (erfc(9.2) is the smallest normal
Float32), for the p of a scan that minimises the error (the error has narrow minima in p where the fit gains an alternation).  Same shape as Abramowitz & Stegun 7.1.26 (which is the degree-4 fit of the ABSOLUTE error of erf, 1.5e-7: a relative
error of 1e-3 at x = 2.7 and unbounded beyond) -- two more Horner steps buy a relative error of 6.4e-7 uniformly in x.

    python tools/gen_erfc_f32.py [D [p ...]]      prints p, the coefficients (ascending), the fit's error and the error of the Float32 evaluation
"""
import sys
import numpy as np
import mpmath as mp

mp.mp.dps = 50


def erfcx(x):
    x = mp.mpf(x)
    if x > 25:       # asymptotic series
        s, term, k = mp.mpf(1), mp.mpf(1), 1
        while k < 40:
            term *= -(2 * k - 1) / (2 * x * x)
            s += term
            k += 1
        return s / (x * mp.sqrt(mp.pi))
    return mp.erfc(x) * mp.exp(x * x)


def target(t, p):
    t = mp.mpf(t)
    if t == 0:
        return p / mp.sqrt(mp.pi)
    x = (1 / t - 1) / p
    return erfcx(x) / t


def remez(p, deg, xmax=10.0, iters=40, N=2000):
    """minimax fit of the RELATIVE error over x in [0, xmax] (erfc(9.2) is the smallest normal Float32), t in [1/(1 + p xmax), 1]"""
    tmin = mp.mpf(1) / (1 + mp.mpf(p) * xmax)
    n = deg + 2
    grid = [tmin + (1 - tmin) * mp.mpf(k) / N for k in range(N + 1)]
    fg = [target(t, p) for t in grid]
    idx = [int(round((1 - np.cos(np.pi * k / (n - 1))) / 2 * N)) for k in range(n)]
    for _ in range(iters):
        A = mp.matrix(n, n)
        b = mp.matrix(n, 1)
        for i, k in enumerate(idx):
            t, f = grid[k], fg[k]
            for j in range(deg + 1):
                A[i, j] = t ** j
            A[i, deg + 1] = (-1) ** i * f          # the relative error is levelled
            b[i] = f
        sol = mp.lu_solve(A, b)
        c = [sol[j] for j in range(deg + 1)]
        err = np.array([float(sum(c[j] * t ** j for j in range(deg + 1)) / f - 1) for t, f in zip(grid, fg)])
        ext = [k for k in range(N + 1) if (k == 0 or abs(err[k]) >= abs(err[k - 1])) and (k == N or abs(err[k]) >= abs(err[k + 1]))]
        alt = []
        for k in ext:                              # alternating signs, the largest of each run
            if alt and (err[alt[-1]] > 0) == (err[k] > 0):
                if abs(err[k]) > abs(err[alt[-1]]):
                    alt[-1] = k
            else:
                alt.append(k)
        while len(alt) > n:
            alt.pop(0) if abs(err[alt[0]]) < abs(err[alt[-1]]) else alt.pop()
        if len(alt) < n or alt == idx:
            break
        idx = alt
    return c, float(np.abs(err).max())


def f32_eval(c, p, x):
    """the kernel's arithmetic in numpy float32 (fma emulated in float64 -> float32: exact product, one rounding)"""
    f = np.float32
    x = x.astype(f)
    t = (f(1) / (x.astype(np.float64) * np.float64(f(p)) + 1.0).astype(f)).astype(f)
    acc = np.full(x.shape, f(c[-1]), dtype=f)
    for ck in c[-2::-1]:
        acc = (acc.astype(np.float64) * t.astype(np.float64) + np.float64(f(ck))).astype(f)
    xx = (x * x).astype(f)
    arg = (xx * f(-1.4426950408889634)).astype(f)
    e = np.exp2(arg.astype(np.float64)).astype(f)
    return ((acc * t).astype(f) * e).astype(f)


def main():
    deg = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    p_list = [float(v) for v in sys.argv[2:]] or [0.370, 0.372, 0.374, 0.375, 0.376, 0.378, 0.380]
    best = None
    for p in p_list:
        c, e = remez(p, deg)
        print(f"p = {p:<8} minimax relative error {e:.3e}")
        if best is None or e < best[2]:
            best = (p, c, e)
    p, c, e = best
    print(f"\nbest p = {p}: relative error of the fit {e:.3e}")
    for j, cj in enumerate(c):
        print(f"  c[{j}] = {mp.nstr(cj, 12)}")
    x = np.concatenate([np.linspace(0, 10, 200001), np.random.default_rng(1).uniform(0, 6, 200000)])
    got = f32_eval([float(v) for v in c], p, x).astype(np.float64)
    ref = np.array([float(mp.erfc(mp.mpf(float(np.float32(v))))) for v in x])
    ok = ref > 1e-37
    rel = np.abs(got[ok] / ref[ok] - 1)
    xs = x[ok]
    for lo, hi in ((0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 8), (8, 9.2)):
        m = (xs >= lo) & (xs < hi)
        print(f"  Float32 evaluation, x in [{lo}, {hi}): max relative error {rel[m].max():.2e}")


if __name__ == "__main__":
    main()
