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

import ctypes
from typing import Sequence, Tuple


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
    lib = _lib.load()
    need = lib.amcx_group_stats_workspace_bytes(G, R, Cc)
    if need < 0:
        raise ValueError(f"statistics over {R} rows x {Cc} columns: at least one row, at most 32 columns")
    ws = torch.empty((max(int(need), 8),), dtype=torch.uint8, device=feats.device)
    with torch.cuda.device(feats.device):
        _lib.check(lib.amcx_group_stats_ws_f32(
            flat.data_ptr(), G, R, flat.stride(1), Cc, mean.data_ptr(), std.data_ptr(), ws.data_ptr(), ws.numel(),
            torch.cuda.current_stream(feats.device).cuda_stream))
    return mean.reshape(lead + (Cc,)), std.reshape(lead + (Cc,))


def select_standardize(rows, cols: Sequence[int]) -> Tuple["object", "object", "object"]:
    """rows: (R, C) float32 on the GPU -> (scaled (R, len(cols)) float32, mean, scale):
    ``StandardScaler().fit_transform(rows[:, cols])`` (columns sklearn cannot tell from constant get
    scale 1, by its own variance bound).  One C-ABI call, three launches, no host round trip."""
    import torch
    rows = _as_rows(rows)
    if rows.dim() != 2:
        raise ValueError("rows must be 2-D (flatten (mod, snr, frame) first)")
    R, Cc = rows.shape
    cols = [int(c) for c in cols]
    if any(c < 0 or c >= Cc for c in cols):
        raise IndexError("column index out of range")
    if not cols or len(cols) > 32 or Cc > 32:
        raise ValueError("between 1 and 32 columns")
    lib = _lib.load()
    out = torch.empty((R, len(cols)), dtype=torch.float32, device=rows.device)
    mean = torch.empty((len(cols),), dtype=torch.float64, device=rows.device)
    scale = torch.empty_like(mean)
    if R == 0:
        raise ValueError("no rows to fit the scaler on")
    need = lib.amcx_standardize_workspace_bytes(R, Cc)
    ws = torch.empty((int(need),), dtype=torch.uint8, device=rows.device)
    cols_c = (ctypes.c_int32 * len(cols))(*cols)
    with torch.cuda.device(rows.device):
        # three launches on the current stream, nothing else: all-column statistics (read 1), chunk merge, transform (read 2)
        _lib.check(lib.amcx_standardize_fit_transform_f32(
            rows.data_ptr(), R, rows.stride(0), Cc, cols_c, len(cols), out.data_ptr(), out.stride(0),
            mean.data_ptr(), scale.data_ptr(), ws.data_ptr(), ws.numel(),
            torch.cuda.current_stream(rows.device).cuda_stream))
    return out, mean, scale
