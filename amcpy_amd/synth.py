"""Synthetic IQ for benchmarks, fixtures and tests.

Every frame is: unit-average-power constellation symbols drawn i.i.d. uniform,
rectangular pulse of ``sps`` samples per symbol, one uniform carrier phase per
frame, plus complex AWGN of standard deviation ``10**(-snr_db/20)``
(N(0, 1/2) per component before scaling).  ``WGN`` is noise only, sigma = 1.
Layout is the reference's container layout: ``(n_snr, n_frames, frame_size)``
complex64 per modulation (reference README.md:60-73,
feature_extraction.py:46-48).

Two generators with the same distribution:

* :func:`host_frames`   numpy, seed = 1000 + 10*mod_idx + snr_idx per
  (modulation, SNR) block -- fixtures, the small CPU-runnable config, tests.
* :func:`device_frames` torch on the GPU, seeded by
  (base=20260701, rank, mod_idx, snr_idx) -- fills the multi-GB benchmark
  shapes directly in HBM (nothing crosses PCIe inside a timed region).
"""

from __future__ import annotations

import math
from typing import Sequence

import numpy as np

MODS6 = ("BPSK", "QPSK", "8PSK", "16QAM", "64QAM", "WGN")
DEVICE_SEED_BASE = 20260701
SAMPLES_PER_SYMBOL = 8


def constellation(mod: str) -> np.ndarray:
    """Unit-average-power constellation points (complex128)."""
    m = mod.upper()
    if m in ("WGN", "NOISE"):
        return np.zeros(1, dtype=np.complex128)
    if m.endswith("PSK"):
        order = {"BPSK": 2, "QPSK": 4}.get(m) or int(m[:-3])
        k = np.arange(order)
        off = math.pi / 4 if order == 4 else 0.0
        return np.exp(1j * (2 * math.pi * k / order + off))
    if m.endswith("QAM"):
        order = int(m[:-3])
        side = int(round(math.sqrt(order)))
        if side * side != order:
            raise ValueError(f"non-square QAM order: {mod}")
        lv = 2.0 * np.arange(side) - (side - 1)
        pts = (lv[:, None] + 1j * lv[None, :]).ravel()
        return pts / math.sqrt((np.abs(pts) ** 2).mean())
    raise ValueError(f"unknown modulation {mod!r}")


def snr_grid(n_snr: int) -> np.ndarray:
    """SNR values in dB: {0, 10} for the 2-point plumbing config, otherwise
    n_snr points from -20 dB in 2 dB steps (26 -> -20..+30)."""
    if n_snr == 2:
        return np.array([0.0, 10.0])
    return -20.0 + 2.0 * np.arange(n_snr)


def host_block(mod: str, snr_db: float, n_frames: int, frame_size: int,
               seed: int, sps: int = SAMPLES_PER_SYMBOL) -> np.ndarray:
    """(n_frames, frame_size) complex64 for one (modulation, SNR)."""
    rng = np.random.default_rng(seed)
    pts = constellation(mod)
    n_sym = -(-frame_size // sps)
    noise = (rng.standard_normal((n_frames, frame_size))
             + 1j * rng.standard_normal((n_frames, frame_size))) * math.sqrt(0.5)
    if len(pts) == 1:                                # WGN: noise only, sigma = 1
        return noise.astype(np.complex64)
    sym = pts[rng.integers(0, len(pts), size=(n_frames, n_sym))]
    base = np.repeat(sym, sps, axis=1)[:, :frame_size]
    phase = np.exp(1j * rng.uniform(0.0, 2 * math.pi, size=(n_frames, 1)))
    sigma = 10.0 ** (-snr_db / 20.0)
    return (base * phase + sigma * noise).astype(np.complex64)


def host_frames(mods: Sequence[str], n_snr: int, n_frames: int, frame_size: int):
    """dict mod -> (n_snr, n_frames, frame_size) complex64, host seeds."""
    grid = snr_grid(n_snr)
    out = {}
    for mi, mod in enumerate(mods):
        blk = np.empty((n_snr, n_frames, frame_size), dtype=np.complex64)
        for si, snr in enumerate(grid):
            blk[si] = host_block(mod, float(snr), n_frames, frame_size,
                                 seed=1000 + 10 * mi + si)
        out[mod] = blk
    return out


def modulation_cycle(n_mods: int):
    """Names for n_mods classes: the six generators cycled (24-class config)."""
    return [MODS6[i % 6] if i < 6 else f"{MODS6[i % 6]}#{i // 6}" for i in range(n_mods)]


def device_frames(mod: str, n_snr: int, n_frames: int, frame_size: int, *,
                  device, rank: int = 0, mod_idx: int = 0,
                  sps: int = SAMPLES_PER_SYMBOL, out=None):
    """(n_snr, n_frames, frame_size) complex64 torch tensor generated in HBM.

    One generator stream per (rank, mod_idx, snr_idx).  ``out`` may be a
    preallocated complex64 tensor of that shape (e.g. a slice of the shard's
    arena) and is filled in place.
    """
    import torch

    base_name = mod.split("#")[0]
    pts = torch.from_numpy(constellation(base_name).astype(np.complex64)).to(device)
    if out is None:
        out = torch.empty((n_snr, n_frames, frame_size), dtype=torch.complex64, device=device)
    grid = snr_grid(n_snr)
    n_sym = -(-frame_size // sps)
    # bound temporaries to ~256 MiB whatever the block size
    chunk = max(1, min(n_frames, (32 << 20) // max(frame_size, 1)))
    for si in range(n_snr):
        gen = torch.Generator(device=device)
        gen.manual_seed((DEVICE_SEED_BASE * 1000003 + rank * 10007 + mod_idx * 101 + si)
                        & 0x7FFFFFFFFFFFFFFF)
        sigma = 10.0 ** (-float(grid[si]) / 20.0) if pts.numel() > 1 else 1.0
        for f0 in range(0, n_frames, chunk):
            f1 = min(n_frames, f0 + chunk)
            dst = torch.view_as_real(out[si, f0:f1])
            dst.normal_(0.0, sigma * math.sqrt(0.5), generator=gen)
            if pts.numel() > 1:
                idx = torch.randint(0, pts.numel(), (f1 - f0, n_sym), device=device, generator=gen)
                ph = torch.rand((f1 - f0, 1), device=device, generator=gen) * (2 * math.pi)
                sym = pts[idx] * torch.polar(torch.ones_like(ph), ph)
                out[si, f0:f1] += sym.repeat_interleave(sps, dim=1)[:, :frame_size]
    return out
