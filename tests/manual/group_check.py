#!/usr/bin/env python3
"""Quick look at the group kernels (N = 16384, 32768) against the oracle on the GPU box: synthetic modulations over a
wide SNR range, ragged frame counts, an out-of-range frame in the middle, a NaN frame, timing on a resident shard."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from amcpy_amd import _lib, synth                      # noqa: E402
from amcpy_amd.features import features18              # noqa: E402
from oracle import iq_features_oracle as orc           # noqa: E402

for N in (16384, 32768):
    print(f"== N = {N}: {_lib.kernel_name(N)}")
    x = np.concatenate([synth.host_block(m, snr, 3, N, seed=900 + 7 * i + j)
                        for i, m in enumerate(synth.MODS6) for j, snr in enumerate((-20.0, 0.0, 14.0, 30.0))]).astype(np.complex64)
    x = x[: x.shape[0] - 1]                           # 71 frames: ragged against batches of 4
    x[10] *= 3e7                                      # outside the fp32 sums' range: re-run path
    x[11] *= 1e-8
    x[40, 5] = np.nan
    got = features18(torch.from_numpy(x).cuda()).cpu().numpy()
    gold = orc.features18_batch(x)
    ok = np.ones(len(x), bool); ok[40] = False
    assert np.isnan(got[40]).all(), got[40]
    plain, scaled = orc.parity_errors(got[ok], gold[ok].astype(np.float32), orc.conditioning_scales(x[ok]))
    print("worst scaled per feature:", " ".join(f"{v:.1e}" for v in scaled.max(axis=0)))
    print("worst plain  per feature:", " ".join(f"{v:.1e}" for v in plain.max(axis=0)))
    print("frames beyond 1e-5:", int((scaled > 1e-5).any(axis=1).sum()), "worst", scaled.max(), "at frame", np.flatnonzero(ok)[scaled.max(axis=1).argmax()])
    # one frame alone == the same frame in the batch
    alone = features18(torch.from_numpy(x[5:6]).cuda()).cpu().numpy()
    print("batch position independent:", np.array_equal(alone.view(np.int32), got[5:6].view(np.int32)))
    # timing
    F = (4 << 30) // (8 * N)
    arena = torch.empty((F, N), dtype=torch.complex64, device="cuda")
    synth.device_frames("QPSK", 1, F, N, device="cuda", rank=0, mod_idx=1, out=arena.view(1, F, N))
    out = torch.empty((F, 18), dtype=torch.float32, device="cuda")
    for _ in range(3):
        features18(arena, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        features18(arena, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{F} frames: {dt * 1e3:.2f} ms per launch, {F / dt / 1e6:.2f} M frames/s, {(8 * N + 72) * F / dt / 1e9:.0f} GB/s = {(8 * N + 72) * F / dt / 8e12:.3f} of 8 TB/s")
    del arena, out
