#!/usr/bin/env python3
"""Derive the polynomial coefficients of rd_log2f / rd_exp2f (the pinned pow of DESIGN.md §3).

The reference computes pow() in WGSL (src/gpu/shaders.rs:217, :261), which a Vulkan driver
lowers to exp2(y*log2(x)) on hardware approximations that cannot be reproduced on a CPU.
We therefore pin ONE polynomial pair, evaluated with explicit fmaf in a fixed order, and
use it verbatim in the oracle (oracle/develop_ref.c), the numpy twin and the HIP kernel.

  log2(m) = t*P(t),  t = m-1,  m in [sqrt(1/2), sqrt(2))     (degree LOG_DEG polynomial P)
  2^f     = Q(f),    f in [-1/2, 1/2]                         (degree EXP_DEG polynomial Q, Q(0)=1)

Weighted Remez exchange in float64 (mpmath for the target functions), coefficients rounded
to float32 and printed as C hex-float literals.  Run:  python tools/fit_pow.py
"""
import numpy as np
import mpmath as mp

mp.mp.dps = 40


def remez(f, a, b, deg, weight=None, iters=40, fixed0=None):
    """Minimax fit of sum c_k x^k to f on [a,b] (absolute error * weight).
    fixed0: if not None, c_0 is fixed to that value (fit the remaining coefficients)."""
    n = deg + (1 if fixed0 is None else 0)  # unknown coefficients
    k = np.arange(n + 1)
    xs = 0.5 * (a + b) + 0.5 * (b - a) * np.cos(np.pi * k / n)[::-1]
    w = weight or (lambda x: 1.0)
    fv = np.vectorize(lambda x: float(f(mp.mpf(x))))
    wv = np.vectorize(lambda x: float(w(x)))
    grid = np.linspace(a, b, 20001)
    fg, wg = fv(grid), wv(grid)
    for _ in range(iters):
        A = np.zeros((n + 1, n + 1))
        powers = range(deg + 1) if fixed0 is None else range(1, deg + 1)
        for j, p in enumerate(powers):
            A[:, j] = xs ** p
        A[:, n] = ((-1.0) ** np.arange(n + 1)) / wv(xs)
        rhs = fv(xs) - (0.0 if fixed0 is None else fixed0)
        sol = np.linalg.solve(A, rhs)
        c = sol[:n]
        coef = np.concatenate(([fixed0], c)) if fixed0 is not None else c
        err = (np.polyval(coef[::-1], grid) - fg) * wg
        # exchange: pick extrema between sign changes
        idx = [0]
        for i in range(1, len(grid)):
            if np.sign(err[i]) != np.sign(err[idx[-1]]) and err[i] != 0:
                idx.append(i)
            elif abs(err[i]) > abs(err[idx[-1]]):
                idx[-1] = i
        if len(idx) < n + 1:
            break
        # keep the n+1 largest consecutive alternations
        while len(idx) > n + 1:
            if abs(err[idx[0]]) < abs(err[idx[-1]]):
                idx.pop(0)
            else:
                idx.pop()
        new = grid[idx]
        if np.allclose(new, xs, rtol=0, atol=1e-12):
            break
        xs = new
    return coef, np.max(np.abs(err))


def hexf(x):
    return float(np.float32(x)).hex()


if __name__ == "__main__":
    s = float(mp.sqrt(mp.mpf(1) / 2))
    # log2(1+t)/t ; absolute error of t*P(t) is what matters (result rel. err = ln2*y*dlog2), so weight by |t|
    g = lambda t: mp.log(1 + t, 2) / t if t != 0 else 1 / mp.log(2)
    for deg in (7, 8, 9):
        c, e = remez(g, s - 1.0, 2 * s - 1.0, deg, weight=lambda t: max(abs(t), 1e-3))
        print(f"log2 deg {deg}: max |t*P(t)-log2(1+t)| = {e:.3e}")
        print("   ", ", ".join(f"{hexf(v)}f" for v in c))
    for deg in (5, 6):
        c, e = remez(lambda f: mp.power(2, f), -0.5, 0.5, deg, fixed0=1.0)
        print(f"exp2 deg {deg}: max abs err = {e:.3e}")
        print("   ", ", ".join(f"{hexf(v)}f" for v in c))
