#!/usr/bin/env python3
"""Calibration of the cancellation flag (round 6, GPU box): how far the kernel's cumulants (fp32 sums) are from the
oracle's, against candidate predicates a finaliser could evaluate from its own moments.

    python tests/manual/cancel_study.py [frame_size=2048] [frames_per_cell=128] [variant=wave]

Per cumulant id (10, 12-18) it prints: frames over the criterion 1e-5 max(|gold|, S); the distribution of
err / Sabs (Sabs: every moment replaced by the mean of its summands' magnitudes) and of err / E (E: first-order error
scale, each moment's absolute error taken as eps x that mean); and for a list of kappa the rate of frames with
S < kappa Sabs resp. max(|C|, S) < kappa E, together with the worst ratio err / (1e-5 max(|gold|, S)) among the frames a
rule does NOT flag.  Not part of the test-suite."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch  # noqa: E402
from amcpy_amd import synth  # noqa: E402
from amcpy_amd.features import features18  # noqa: E402
from oracle import iq_features_oracle as orc  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(args[0]) if len(args) > 0 else 2048
per = int(args[1]) if len(args) > 1 else 128
variant = args[2] if len(args) > 2 else "wave"
# --device-arena: the frames bench.py times (synth.device_frames, rank 0: the benchmark's own arena, cell by cell) instead of
# the host generator's; `per` = 4096 is then the whole BASELINE shape
DEVICE_ARENA = "--device-arena" in sys.argv
IDS = (10, 12, 13, 14, 15, 16, 17, 18)


def error_scales(m):
    """First-order absolute error of each cumulant when moment m_pq carries an absolute error of (1 unit) x the mean of its
    summands' magnitudes (m21 for order 2, m42 for order 4, m63 for order 6)."""
    a20, a40, a41 = np.abs(m["m20"]), np.abs(m["m40"]), np.abs(m["m41"])
    m21, m42, m63 = np.abs(m["m21"]), np.abs(m["m42"]), np.abs(m["m63"])
    d2, d4, d6 = m21, m42, m63
    return {
        10: d2,
        12: d4 + 6 * a20 * d2,
        13: d4 + 3 * (a20 * d2 + m21 * d2),
        14: d4 + 2 * a20 * d2 + 4 * m21 * d2,
        15: d6 + 15 * (a20 * d4 + a40 * d2) + 9 * a20 ** 2 * d2,
        16: d6 + 5 * (m21 * d4 + a40 * d2) + 10 * (a20 * d4 + a41 * d2) + 30 * (2 * a20 * m21 * d2 + a20 ** 2 * d2),
        17: d6 + 6 * (a20 * d4 + m42 * d2) + 8 * (m21 * d4 + a41 * d2) + (a20 * d4 + a40 * d2) + 18 * a20 ** 2 * d2
            + 24 * (2 * m21 * a20 * d2 + m21 ** 2 * d2),
        18: d6 + 9 * (m21 * d4 + m42 * d2) + 36 * m21 ** 2 * d2 + 6 * (a20 * d4 + a41 * d2) + 18 * (2 * a20 * m21 * d2 + a20 ** 2 * d2),
    }


def device_rule(m, kappa):
    """The predicate as the finaliser evaluates it (amcx_math.h: cancellation_suspect): leading moment exact (squares), the
    other complex moments by max(|re|, |im|) <= |z| <= |re| + |im|.  Returns (F,) bool."""
    a20 = np.abs(m["m20"]); n20 = a20 ** 2
    m21, m42, m63 = m["m21"].real, m["m42"].real, m["m63"].real
    lo = lambda z: np.maximum(np.abs(z.real), np.abs(z.imag))  # noqa: E731
    hi = lambda z: np.abs(z.real) + np.abs(z.imag)             # noqa: E731
    a40l, a40h, a41l, a41h = lo(m["m40"]), hi(m["m40"]), lo(m["m41"]), hi(m["m41"])
    E12 = m42 + 6 * a20 * m21
    E13 = m42 + 3 * m21 ** 2 + 3 * a20 * m21
    E15 = m63 + 15 * (a20 * m42 + a40h * m21) + 9 * n20 * m21
    E16 = m63 + 5 * m21 * m42 + 5 * a40h * m21 + 10 * a20 * m42 + 10 * a41h * m21 + 60 * a20 * m21 ** 2 + 30 * n20 * m21
    E17 = m63 + 14 * m21 * m42 + 24 * m21 ** 3 + 7 * a20 * m42 + (a40h + 8 * a41h) * m21 + 18 * n20 * m21 + 48 * m21 ** 2 * a20
    t12 = kappa * E12 - 3 * n20
    t13 = kappa * E13 - 3 * a20 * m21
    t15 = kappa * E15 - 15 * a20 * a40l - 3 * a20 * n20
    t16 = kappa * E16 - 5 * m21 * a40l - 10 * a20 * a41l - 30 * n20 * m21
    t17 = kappa * E17 - 6 * a20 * m42 - 8 * m21 * a41l - a20 * a40l - 6 * a20 * n20 - 24 * m21 ** 2 * a20
    f = (t12 > 0) & (np.abs(m["m40"]) < t12)
    f |= (t13 > 0) & (np.abs(m["m41"]) < t13)
    f |= (t15 > 0) & (np.abs(m["m60"]) < t15)
    f |= (t16 > 0) & (np.abs(m["m61"]) < t16)
    f |= (t17 > 0) & (np.abs(m["m62"]) < t17)
    return f


KAPPAS = (4e-3, 6e-3, 8e-3, 1e-2, 1.5e-2, 2e-2)
dev_flags = {k: [] for k in KAPPAS}
rows = []
for mi, mod in enumerate(synth.MODS6):
    for si, snr in enumerate(synth.snr_grid(26)):
        if DEVICE_ARENA:
            if si == 0:
                cell_block = synth.device_frames(mod, 26, per, N, device=torch.device("cuda", 0), rank=0, mod_idx=mi)
            xd = cell_block[si]
            x = xd.cpu().numpy()
            got = features18(xd, variant=variant).cpu().numpy().astype(np.float64)
        else:
            x = synth.host_block(mod, float(snr), per, N, seed=70000 + 100 * mi + si)
            got = features18(torch.from_numpy(x).cuda(), variant=variant).cpu().numpy().astype(np.float64)
        m = orc.batch_moments(x)
        terms = orc.cumulant_terms(m)
        S = orc.conditioning_scales(x)
        Sabs = orc.conditioning_scales(x, absolute=True)
        E = error_scales(m)
        for k in KAPPAS:
            dev_flags[k].append(device_rule(m, k))
        for fid in IDS:
            gold = np.abs(sum(terms[fid])).astype(np.float32).astype(np.float64)
            err = np.abs(got[:, fid - 1] - gold)
            rows.append(np.stack([np.full(per, fid), np.full(per, mi), np.full(per, snr), err, gold, S[:, fid - 1],
                                  Sabs[:, fid - 1], E[fid]], axis=1))
R = np.concatenate(rows)
fid, mi, snr, err, gold, S, Sabs, E = R.T
lim = 1e-5 * np.maximum(gold, S)
ratio = err / lim
n_frames = len(synth.MODS6) * 26 * per
print(f"N={N} variant={variant}: {n_frames} frames" + (" of the benchmark's own arena (synth.device_frames, rank 0)" if DEVICE_ARENA else ""))
for f in IDS:
    k = fid == f
    q = lambda v: " ".join(f"{x:.2e}" for x in np.quantile(v, [0.5, 0.9, 0.99, 0.999, 1.0]))  # noqa: E731
    print(f"id {f:2d}: over {int((ratio[k] > 1).sum()):4d}  worst ratio {ratio[k].max():6.2f}   err/Sabs q50/90/99/99.9/max {q(err[k] / Sabs[k])}"
          f"   err/E {q(err[k] / E[k])}")
frame_key = (R[:, 1] * 26 + (R[:, 2] + 20) / 2) * per
# a per-frame view needs the frame index: rows of one cell are in frame order, one block per id
idx = np.concatenate([np.arange(per) for _ in range(len(rows))])
frame_id = (frame_key + idx).astype(np.int64)
print("rule: S < kappa * Sabs")
for kappa in (5e-4, 1e-3, 2e-3, 3e-3, 4e-3):
    flag = S < kappa * Sabs
    flagged_frames = np.unique(frame_id[flag]).size
    print(f"  kappa {kappa:.0e}: flagged {flagged_frames / n_frames:7.3%} of frames; worst unflagged ratio {ratio[~flag].max():.2f}")
print("rule: max(|C|, S) < kappa * E   (E in units of the summands' mean magnitude)")
for kappa in (5e-4, 1e-3, 2e-3, 3e-3, 4e-3):
    flag = np.maximum(gold, S) < kappa * E
    flagged_frames = np.unique(frame_id[flag]).size
    print(f"  kappa {kappa:.0e}: flagged {flagged_frames / n_frames:7.3%} of frames; worst unflagged ratio {ratio[~flag].max():.2f}")
by_mod = {}
flag = np.maximum(gold, S) < 2e-3 * E
for m_i, mod in enumerate(synth.MODS6):
    k = mi == m_i
    print(f"  {mod:6s} flagged by E-rule at 2e-3: {np.unique(frame_id[flag & k]).size / (26 * per):7.3%}   by Sabs-rule at 2e-3: "
          f"{np.unique(frame_id[(S < 2e-3 * Sabs) & k]).size / (26 * per):7.3%}")
print("rule: the finaliser's predicate (leading moment exact, the others by max-norm / 1-norm)")
cell = len(synth.MODS6) * 26
for kappa in KAPPAS:
    fl = np.concatenate(dev_flags[kappa])                       # per frame, cells in generation order
    per_row = np.concatenate([np.tile(fl[c * per:(c + 1) * per], len(IDS)) for c in range(cell)])
    worst = ratio[~per_row].max()
    by_mod = " ".join(f"{synth.MODS6[i]}:{fl[i * 26 * per:(i + 1) * 26 * per].mean():.3%}" for i in range(len(synth.MODS6)))
    print(f"  kappa {kappa:.1e}: flagged {fl.mean():7.3%}; worst unflagged ratio {worst:.2f}; {by_mod}")
# the tail of err / E among cancelled cumulants (S < 0.05 E: the float32 store's own rounding is out of the picture)
k = S < 0.05 * E
print("err/E where S < 0.05 E: n", int(k.sum()), "q50/90/99/99.9/max", " ".join(f"{x:.2e}" for x in np.quantile(err[k] / E[k], [0.5, 0.9, 0.99, 0.999, 1.0])))
for m_i, mod in enumerate(synth.MODS6):
    kk = k & (mi == m_i)
    if kk.any():
        print(f"  {mod:6s} n {int(kk.sum()):6d} max err/E {np.max(err[kk] / E[kk]):.2e} at snr {snr[kk][np.argmax(err[kk] / E[kk])]:.0f}")
