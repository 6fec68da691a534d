#!/usr/bin/env python3
"""Where a call of the real-data upload path spends its time, by pinned-slot size and staging threads
(amcx_upload_stats).  python tools/h2d_tune.py [small|big]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from bench import _fortran_container
from amcpy_amd.feature_extraction import FrameRows, HipEngine

which = sys.argv[1] if len(sys.argv) > 1 else "small"
S, K, N = (2, 500, 2048) if which == "small" else (26, 4096, 2048)
rows = FrameRows(_fortran_container(S, K, N), S, K)
reps = 12 if which == "small" else 2
print(f"{which}: ({S}, {K}, {N}) complex128 Fortran-ordered, {S * K * N * 16 / 1e6:.1f} MB per call, {reps} calls")
for slot_mb in (8, 32):
    for threads in (4, 8, 16):
        eng = HipEngine(N, chunk_bytes=slot_mb << 20, threads=threads)
        eng(rows)
        t0 = time.perf_counter()
        for _ in range(reps):
            eng(rows)
        wall = (time.perf_counter() - t0) / reps
        st = eng.stats
        print(f"slot {slot_mb:3d} MB thr {threads:2d}: {S * K * N * 16 / wall / 1e9:6.1f} GB/s  call {wall * 1e6:8.0f} us  native {st['seconds_native'] * 1e6:8.0f}"
              f"  prepare {st['seconds_prepare'] * 1e6:5.0f} staging {st['seconds_staging'] * 1e6:7.0f} waiting {st['seconds_waiting'] * 1e6:7.0f}"
              f" tail {st['seconds_tail'] * 1e6:6.0f}  chunks {st['chunks']}")
        eng.close()
