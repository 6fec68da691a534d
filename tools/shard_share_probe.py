#!/usr/bin/env python3
"""One rank's share of a (26, 512, 2048) complex128 variable of an uncompressed .mat (world 8, rank 3) through the
engine, under the two ways of cutting a container over ranks: a contiguous range of the snr-major flattening
(1664 frames = 3.25 snr rows: a few elements per run of a column-major plane) and a frame range of every snr row
(26 x 64 frames: one contiguous run per plane, read straight from the file).  Same frame count, same GPU work.
    python tools/shard_share_probe.py"""
import sys
import tempfile
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import scipy.io
from amcpy_amd.feature_extraction import FrameColumns, FrameRows, HipEngine
from amcpy_amd.matfile import load_variable
from amcpy_amd.sharding import shard_by_frames, shard_range

S, K, N, W, R = 26, 512, 2048, 8, 3
rng = np.random.default_rng(0)
x = np.asfortranarray((rng.standard_normal((S, K, N)) + 1j * rng.standard_normal((S, K, N))))
with tempfile.TemporaryDirectory(dir="/dev/shm" if Path("/dev/shm").is_dir() else None) as td:
    path = Path(td) / "c.mat"
    scipy.io.savemat(str(path), {"x": x})
    fx = load_variable(path, "x", direct=True)
    mapped = load_variable(path, "x")
    eng = HipEngine(N, threads=8)
    lo, hi = shard_range(S * K, R, W)
    k_lo, k_hi = shard_range(K, R, W)
    assert shard_by_frames(S, K, W) and hi - lo == S * (k_hi - k_lo)
    eng(FrameRows(x, S, K))                                    # context, pinned slots, device scratch
    for label, rows in [("flattening range, array in memory", FrameRows(x, S, K, lo, hi)),
                        ("flattening range, memory-mapped file", FrameRows(mapped, S, K, lo, hi)),
                        ("frame range of every snr row, array in memory", FrameColumns(x, S, K, k_lo, k_hi)),
                        ("frame range of every snr row, memory-mapped file", FrameColumns(mapped, S, K, k_lo, k_hi)),
                        ("frame range of every snr row, read from the file", FrameColumns(fx, S, K, k_lo, k_hi))]:
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); out = eng(rows); ts.append((time.perf_counter() - t0) * 1e3)
        print(f"{label:52s} {rows.shape[0]} frames: " + " ".join(f"{t:6.2f}" for t in ts) + f" ms | best {rows.shape[0] * N * 16 / min(ts) / 1e6:5.1f} GB/s "
              f"of samples, from_file {eng.stats['from_file']}")
    a = eng(FrameColumns(fx, S, K, k_lo, k_hi)).reshape(S, k_hi - k_lo, 18)
    b = eng(FrameRows(x, S, K)).reshape(S, K, 18)[:, k_lo:k_hi]
    print("frame-axis share equals the same frames of the whole container:", bool(np.array_equal(a, b, equal_nan=True)))
