#!/usr/bin/env python3
"""Timing of the consumer kernels on the benchmark-shaped feature matrix
(6 x 26 x 4096 x 18 float32 = 46 MB): per-SNR statistics and select+standardise."""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from amcpy_amd.postprocess import snr_statistics, select_standardize

x = torch.randn((6, 26, 4096, 18), device="cuda") * 3 + 1
rows = x.reshape(-1, 18)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
nbytes = x.numel() * 4
ms1 = t(lambda: snr_statistics(x))
ms2 = t(lambda: select_standardize(rows, [2, 4, 6, 8, 12, 14]))
# algorithmic bytes: the statistics read the matrix once; fit + transform read it twice and write 6 of 18 columns
print(json.dumps({"frames": rows.shape[0], "matrix_MB": nbytes / 1e6,
                  "snr_statistics_ms": ms1, "snr_statistics_GBps": nbytes / ms1 / 1e6,
                  "select_standardize_ms": ms2, "select_standardize_GBps": (2 * nbytes + rows.shape[0] * 24) / ms2 / 1e6}))
