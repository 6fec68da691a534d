#!/usr/bin/env python3
"""Minimax fit of atan(r) = r*Q(r^2) on r in [0,1] (Remez exchange, fp64),
then the error of the fp32 Horner evaluation.  Produces the coefficients
hard-coded in amcpy_amd/csrc/amcx_math.h."""
import numpy as np, sys

def remez(nterms, iters=60):
    # unknowns: c0..c_{n-1}, E ; error e(r) = r*Q(r^2) - atan(r), equioscillating
    n = nterms
    k = np.arange(n + 1)
    r = 0.5 * (1 - np.cos(np.pi * (k + 0.5) / (n + 1)))  # nodes in (0,1)
    r = np.clip(r, 1e-6, 1.0); r[-1] = 1.0
    grid = np.linspace(1e-9, 1.0, 400001)
    for _ in range(iters):
        A = np.zeros((n + 1, n + 1))
        for j in range(n):
            A[:, j] = r ** (2 * j + 1)
        A[:, n] = (-1.0) ** k
        sol = np.linalg.solve(A, np.arctan(r))
        c = sol[:n]
        e = sum(c[j] * grid ** (2 * j + 1) for j in range(n)) - np.arctan(grid)
        # new extrema: local maxima of |e| between sign changes
        idx = [0]
        s = np.sign(e)
        # split at zero crossings, take argmax |e| in each segment
        cross = np.where(s[1:] * s[:-1] < 0)[0]
        bounds = np.concatenate([[0], cross + 1, [len(grid)]])
        ext = []
        for a, b in zip(bounds[:-1], bounds[1:]):
            seg = np.abs(e[a:b]); ext.append(a + int(np.argmax(seg)))
        if len(ext) != n + 1:
            break
        r = grid[ext]
    return c, np.abs(e).max()

for n in (7, 8, 9, 10):
    c, err = remez(n)
    # fp32 evaluation
    rr = np.linspace(0, 1, 2000001).astype(np.float32)
    s = rr * rr
    q = np.float32(c[-1]) * np.ones_like(s)
    for cj in c[-2::-1]:
        q = (q * s + np.float32(cj)).astype(np.float32)   # not true fma but close
    val = (q * rr).astype(np.float32)
    e32 = np.abs(val.astype(np.float64) - np.arctan(rr.astype(np.float64))).max()
    print(n, "terms: fp64 minimax err %.3e  fp32 eval err %.3e" % (err, e32))
    print("   ", ", ".join("%.9ef" % v for v in c))
