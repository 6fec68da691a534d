#!/usr/bin/env python3
"""How fast does a RadioML-shaped HDF5 container go through extract_radioml_hdf5 (GPU box)?  Writes F frames of (1024, 2)
float32 pairs with amcpy_amd.hdf5_min -- contiguous, and chunked + deflate -- into the temp dir, runs the extraction twice
(second run: page cache warm) and prints frames/s and container GB/s.
    python tools/hdf5_ingest_probe.py [frames=106496]          # 26 x 4096: one configs[4] modulation, 872 MB"""
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from amcpy_amd import hdf5_min, synth  # noqa: E402
from amcpy_amd.feature_extraction import extract_radioml_hdf5  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 26 * 4096
base = np.concatenate([synth.host_block(m, 8.0, 512, 1024, seed=3 + i) for i, m in enumerate(synth.MODS6)])   # 3072 frames
frames = base[np.arange(F) % base.shape[0]]
pairs = np.ascontiguousarray(np.stack([frames.real, frames.imag], axis=-1).astype(np.float32))
print(f"library {hdf5_min.library_path()}; {F} frames x 1024 x 2 float32 = {pairs.nbytes / 1e6:.0f} MB")
with tempfile.TemporaryDirectory() as d:
    for label, kw in (("contiguous", {}), ("chunked (64 frames) + deflate 1", {"chunks": (64, 1024, 2), "deflate": 1})):
        path = Path(d) / "x.h5"
        with hdf5_min.File(path, "w") as fh:
            fh.create_dataset("X", pairs, **kw)
        size = path.stat().st_size
        for run in range(2):
            t0 = time.perf_counter()
            out = extract_radioml_hdf5(path)
            dt = time.perf_counter() - t0
            print(f"  {label:32s} file {size / 1e6:7.0f} MB  run {run}: {dt * 1e3:8.1f} ms = {F / dt / 1e6:6.3f} M frames/s, "
                  f"{pairs.nbytes / dt / 1e9:5.2f} GB/s of samples")
        assert out.shape == (F, 18) and np.isfinite(out[:, 0]).all()
