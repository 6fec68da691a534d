"""Device-side versions of the two consumers directly behind the extraction path
(SURVEY.md section 8f), so the (frames x 18) feature matrix can stay in HBM:

* :func:`snr_statistics`  -- per-SNR mean / std over frames, the triple loop of the
  reference's ``graphics._compute_stats`` (graphics.py:50-62; np.std is the population
  standard deviation);
* :func:`select_standardize` -- column pick + ``StandardScaler().fit_transform``
  (preprocessing.py:52-62).  Column indices are taken as the reference passes them,
  ``list(cfg.features.used)`` used as 0-based columns (so ids (2,4,6,8,12,14) select
  features 3,5,7,9,13,15 -- the reference's behaviour, SURVEY.md section 8f, kept so
  that downstream models see the same inputs).

Inputs and outputs are torch tensors on the GPU; the arithmetic is in the HIP kernels
of amcpy_amd/csrc/amcx_post_kernels.h behind the C ABI.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np

from . import _lib


def _as_rows(feats):
    import torch
    if not isinstance(feats, torch.Tensor) or feats.dtype != torch.float32 or not feats.is_cuda:
        raise TypeError("feats must be a float32 CUDA tensor")
    if feats.dim() < 2 or feats.stride(-1) != 1:
        raise ValueError("feats must be (..., rows, cols) with unit stride in the last dimension")
    return feats


def snr_statistics(feats):
    """(..., n_frames, C) float32 -> (mean, std), each (..., C) float64: statistics over
    the frame axis for every leading index (e.g. (n_mods, n_snr, n_frames, 18))."""
    import torch
    feats = _as_rows(feats)
    lead, R, Cc = feats.shape[:-2], feats.shape[-2], feats.shape[-1]
    flat = feats.reshape(-1, R, Cc)
    if flat.data_ptr() != feats.data_ptr() or (flat.shape[0] > 1 and flat.stride(0) != R * flat.stride(1)):
        flat = feats.contiguous().reshape(-1, R, Cc)
    G = flat.shape[0]
    mean = torch.empty((G, Cc), dtype=torch.float64, device=feats.device)
    std = torch.empty_like(mean)
    with torch.cuda.device(feats.device):
        _lib.check(_lib.load().amcx_group_stats_f32(
            flat.data_ptr(), G, R, flat.stride(1), Cc, mean.data_ptr(), std.data_ptr(),
            torch.cuda.current_stream(feats.device).cuda_stream))
    return mean.reshape(lead + (Cc,)), std.reshape(lead + (Cc,))


def _column_stats(rows, chunk: int = 1024):
    """Column mean / population std over ALL rows of a 2-D matrix: the group kernel over
    row chunks (one workgroup each, so a 600 k-row matrix fills the chip), then the exact
    pooled combination of the per-chunk (n, mean, M2) in fp64 on the tiny result."""
    import torch
    R = rows.shape[0]
    parts = []
    full = (R // chunk) * chunk
    if full:
        m, s = snr_statistics(rows[:full].reshape(R // chunk, chunk, rows.shape[1]))
        parts.append((torch.full((m.shape[0], 1), float(chunk), dtype=torch.float64, device=rows.device), m, s))
    if R - full:
        m, s = snr_statistics(rows[full:][None])
        parts.append((torch.full((1, 1), float(R - full), dtype=torch.float64, device=rows.device), m, s))
    n = torch.cat([p[0] for p in parts]); m = torch.cat([p[1] for p in parts]); sd = torch.cat([p[2] for p in parts])
    mean = (n * m).sum(dim=0) / R
    m2 = (n * sd * sd + n * (m - mean) ** 2).sum(dim=0)
    return mean, torch.sqrt(m2 / R)


def select_standardize(rows, cols: Sequence[int]) -> Tuple["object", "object", "object"]:
    """rows: (R, C) float32 on the GPU -> (scaled (R, len(cols)) float32, mean, scale):
    ``StandardScaler().fit_transform(rows[:, cols])`` (zero-variance columns get scale 1,
    as sklearn does)."""
    import torch
    rows = _as_rows(rows)
    if rows.dim() != 2:
        raise ValueError("rows must be 2-D (flatten (mod, snr, frame) first)")
    R, Cc = rows.shape
    cols = [int(c) for c in cols]
    if any(c < 0 or c >= Cc for c in cols):
        raise IndexError("column index out of range")
    mean_all, std_all = _column_stats(rows)
    idx = torch.tensor(cols, dtype=torch.int64, device=rows.device)
    mean = mean_all[idx].contiguous()
    scale = std_all[idx].contiguous()
    scale = torch.where(scale < 10 * np.finfo(np.float64).eps, torch.ones_like(scale), scale)
    cols_dev = torch.tensor(cols, dtype=torch.int32, device=rows.device)
    out = torch.empty((R, len(cols)), dtype=torch.float32, device=rows.device)
    with torch.cuda.device(rows.device):
        _lib.check(_lib.load().amcx_select_scale_f32(
            rows.data_ptr(), R, rows.stride(0), cols_dev.data_ptr(), len(cols), mean.data_ptr(),
            scale.data_ptr(), out.data_ptr(), out.stride(0),
            torch.cuda.current_stream(rows.device).cuda_stream))
    return out, mean, scale
