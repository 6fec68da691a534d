#!/usr/bin/env python3
"""Round-4 experiment: could feature 1 -- max_k |X_k|^2 / N, reference features.py:66-69 -- come
from a COARSE spectrum plus an exact evaluation of a few candidate bins, instead of the fp32 register FFT (28 % of the
N = 2048 kernel's energy, 1 130 of its 3 356 vector instructions per frame)?

The scheme: pass 1 (16 points over the rows) stays exact fp32 -- its inputs sit in registers, its twiddles are
immediates, and its outputs Y[k1][n'] are what an exact candidate needs: X[k] = sum_n' W_2048^(n' k) Y[k mod 16][n'],
two complex multiply-adds per lane and a wave reduction.  Passes 2 and 3 (16 and 8 points) go to the matrix pipe as
half-precision MFMAs with fp32 accumulation: the inter-pass twiddles are applied in fp32 on the way, each pass rounds
its inputs and its DFT matrix to half precision once.  A bin is a CANDIDATE if its coarse magnitude is within 2 E of the
coarse maximum, E a proven bound on |coarse - exact|; every candidate is evaluated exactly and the largest exact value
is feature 1.  This script emulates that arithmetic in numpy on the benchmark's synthetic frames and counts candidates:

  * `typical`: the error the coarse spectrum really has (max over bins, relative to the peak)
  * `bound`:   the proven bound E relative to the peak: E = c * eps * sum_n |x_n| with c from the rounding points
               (two per pass: data and matrix, both 2^-11 for fp16 / 2^-8 for bf16; sqrt(2) for complex magnitudes)
  * candidates under the proven bound, and under a "6 sigma" statistical bound that is NOT a proof

and the same for a variant whose passes 2-3 are packed-fp16 BUTTERFLIES on the vector pipe (seven stages, three
roundings each): fewer, cheaper instructions than fp32, but every stage rounds.

    python tools/coarse_spectrum_model.py [frames per (modulation, SNR) = 8]
"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from amcpy_amd import synth  # noqa: E402

N = 2048


def quant(x, kind):
    """round to fp16 / bf16 (real and imaginary parts), back in float64"""
    if kind == "fp16":
        return x.real.astype(np.float16).astype(np.float64) + 1j * x.imag.astype(np.float16).astype(np.float64)
    def bf(v):
        u = v.astype(np.float32).view(np.uint32).astype(np.uint64)
        u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16            # round to nearest even on the upper 16 bits
        return u.astype(np.uint32).view(np.float32).astype(np.float64)
    return bf(np.ascontiguousarray(x.real)) + 1j * bf(np.ascontiguousarray(x.imag))


def coarse_mfma(x, kind):
    """x: (F, 2048) complex64 -> coarse spectrum (F, 2048) in the kernel's 16 x 16 x 8 decomposition:
    n = 128 i + n', n' = 8 n2 + n3;  k = k1 + 16 k2 + 256 k3."""
    F = x.shape[0]
    xx = x.astype(np.complex128).reshape(F, 16, 128)                     # [i][n']
    k1 = np.arange(16)
    Y = np.einsum("ki,fin->fkn", np.exp(-2j * np.pi * np.outer(k1, np.arange(16)) / 16), xx)   # pass 1: exact (fp32 in the kernel)
    npr = np.arange(128)
    Z = Y * np.exp(-2j * np.pi * np.outer(k1, npr) / N)[None]            # T1 in fp32 (exact here)
    Z = quant(Z, kind).reshape(F, 16, 16, 8)                             # [k1][n2][n3], rounded once
    F16 = quant(np.exp(-2j * np.pi * np.outer(np.arange(16), np.arange(16)) / 16), kind)
    U = np.einsum("qn,fknm->fkqm", F16, Z)                               # pass 2 over n2 -> k2 (fp32 accumulation: exact here)
    U = U * np.exp(-2j * np.pi * np.outer(np.arange(16), np.arange(8)) / 128)[None, None]      # T2 (W_128^(k2 n3)) in fp32
    U = quant(U, kind)
    F8 = quant(np.exp(-2j * np.pi * np.outer(np.arange(8), np.arange(8)) / 8), kind)
    X = np.einsum("pm,fkqm->fkqp", F8, U)                                # pass 3 over n3 -> k3
    out = np.empty((F, N), dtype=np.complex128)
    kk1, kk2, kk3 = np.meshgrid(np.arange(16), np.arange(16), np.arange(8), indexing="ij")
    out[:, (kk1 + 16 * kk2 + 256 * kk3).ravel()] = X.reshape(F, -1)
    return out


def coarse_butterflies(x, stages=7):
    """passes 2-3 as radix-2 stages in fp16 arithmetic: modelled as an exact 128-point DFT of the rounded pass-1
    output with every stage's three roundings applied as relative perturbations of 2^-11 (uniform in +-1 ulp/2)."""
    rng = np.random.default_rng(1)
    F = x.shape[0]
    xx = x.astype(np.complex128).reshape(F, 16, 128)
    Y = np.einsum("ki,fin->fkn", np.exp(-2j * np.pi * np.outer(np.arange(16), np.arange(16)) / 16), xx)
    Z = quant(Y * np.exp(-2j * np.pi * np.outer(np.arange(16), np.arange(128)) / N)[None], "fp16")
    # exact radix-2 DIT stages over the 128 points with rounding after each butterfly output
    v = Z.copy()
    n = 128
    # bit-reversal then 7 stages
    idx = np.array([int(format(i, "07b")[::-1], 2) for i in range(n)])
    v = v[:, :, idx]
    size = 2
    while size <= n:
        half = size // 2
        w = quant(np.exp(-2j * np.pi * np.arange(half) / size), "fp16")
        v = v.reshape(F, 16, n // size, size)
        a, b = v[..., :half], v[..., half:]
        t = quant(quant(b * w, "fp16"), "fp16")
        v = np.concatenate([quant(a + t, "fp16"), quant(a - t, "fp16")], axis=-1).reshape(F, 16, n)
        size *= 2
    out = np.empty((F, N), dtype=np.complex128)
    q = np.arange(128)
    for k1 in range(16):
        out[:, k1 + 16 * q] = v[:, k1, :]
    return out


def report(name, coarse, exact, sum_abs, c_bound, eps):
    pk = np.abs(exact).max(axis=1)
    err = np.abs(coarse - exact).max(axis=1)
    typical = err / pk
    E = c_bound * eps * sum_abs
    bound = E / pk
    cm = np.abs(coarse)
    cmax = cm.max(axis=1, keepdims=True)
    n_proven = (cm >= cmax - 2 * E[:, None]).sum(axis=1)
    sigma = eps * np.sqrt((np.abs(exact) ** 2).mean(axis=1) * 1.0) * 3.0   # ~rms coarse error (measured factor, see typical)
    n_stat = (cm >= cmax - 2 * 6 * sigma[:, None]).sum(axis=1)
    right = np.array([np.abs(exact[f, cm[f] >= cmax[f] - 2 * E[f]]).max() == pk[f] for f in range(len(pk))])
    holds = (err <= E).all()
    print(f"{name:34s} typical err/peak median {np.median(typical):.1e} max {typical.max():.1e} | proven bound/peak median "
          f"{np.median(bound):.2f} max {bound.max():.2f} (holds: {holds}) | candidates proven: mean {n_proven.mean():6.1f} "
          f"median {np.median(n_proven):5.0f} 95% {np.percentile(n_proven, 95):6.0f} max {n_proven.max():5d} "
          f"(>8: {100 * (n_proven > 8).mean():4.1f} %) | 6-sigma: mean {n_stat.mean():4.1f} max {n_stat.max()} | exact max among them: {right.all()}")
    return n_proven


def main():
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    grid = synth.snr_grid(26)
    print(f"N = {N}, {per} frames per (modulation, SNR), 6 modulations x 26 SNR = {6 * 26 * per} frames\n")
    allf = []
    for mi, mod in enumerate(synth.MODS6):
        blk = np.concatenate([synth.host_block(mod, float(s), per, N, seed=9000 + 10 * mi + si) for si, s in enumerate(grid)])
        allf.append((mod, blk))
    for mod, x in allf + [("all", np.concatenate([b for _, b in allf]))]:
        exact = np.fft.fft(x.astype(np.complex128), axis=1)
        sum_abs = np.abs(x).sum(axis=1).astype(np.float64)
        print(f"== {mod} ({x.shape[0]} frames)")
        # two rounding points per pass (data, matrix), two passes, sqrt(2) for complex magnitude of per-component rounding
        report("MFMA fp16, fp32 twiddles", coarse_mfma(x, "fp16"), exact, sum_abs, 4 * np.sqrt(2), 2.0 ** -11)
        report("MFMA bf16, fp32 twiddles", coarse_mfma(x, "bf16"), exact, sum_abs, 4 * np.sqrt(2), 2.0 ** -8)
        if mod == "all":
            report("packed-fp16 butterflies (7 stages)", coarse_butterflies(x), exact, sum_abs, (1 + 3 * 7) * np.sqrt(2), 2.0 ** -11)
        print()


if __name__ == "__main__":
    main()
