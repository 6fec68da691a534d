#!/usr/bin/env python3
"""Worst case for the tie path: noiseless BPSK on the real axis -- every symbol change is
an exact +-pi step, so every frame is flagged and recomputed by the fix-up kernel."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from amcpy_amd.features import features18
from oracle import iq_features_oracle as orc
F, N = 156 * 1024, 2048
g = torch.Generator(device="cuda").manual_seed(1)
sym = (torch.randint(0, 2, (F, N // 8), device="cuda", generator=g) * 2 - 1).float()
x = torch.complex(sym.repeat_interleave(8, dim=1), torch.zeros((F, N), device="cuda"))
for _ in range(2): y = features18(x)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5): y = features18(x)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 5
print(f"{F} all-flagged frames: {ms:.3f} ms per pass = {F / ms / 1e3:.1f} M frames/s")
xs = x[:16].cpu().numpy()
gold = orc.features18_batch(xs).astype(np.float32)
got = y[:16].cpu().numpy()
print("f5 got/gold:", got[:3, 4], gold[:3, 4], " f9:", got[:3, 8], gold[:3, 8])
print("max rel err f5,f9:", np.abs(got[:, [4, 8]] - gold[:, [4, 8]]).max() / np.abs(gold[:, [4, 8]]).max())
