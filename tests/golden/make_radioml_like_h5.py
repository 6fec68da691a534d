#!/usr/bin/env python3
"""Writes tests/golden/radioml_like.h5 with the REAL h5py (the library the reference's legacy scripts read RadioML with,
old/dataset.py:43-56): the container layout of RadioML 2018.01A's GOLD_XYZ_OSC.0001_1024.hdf5 -- `X` float32 (F, 1024, 2)
(I, Q) pairs, `Y` int64 (F, 24) one-hot classes, `Z` int64 (F, 1) SNR in dB -- at F = 24 frames of this repo's synthetic
modulations (the real 21 GB file is not available here: no network).  `X` is chunked, shuffled and gzip-compressed, `Z`
chunked, `Y` contiguous, so that the reader meets all three storage layouts.

h5py is not importable by the image's system interpreter; its conda interpreter has it:

    /opt/conda/bin/python3.9 tests/golden/make_radioml_like_h5.py

radioml_like.json records the SHA-256 of each array's bytes as they were BEFORE h5py wrote them: a reader that returns
the same bytes has read the file correctly, whichever library decodes it."""
import hashlib
import json
import sys
from pathlib import Path

import h5py
import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parents[1]))
from amcpy_amd import synth  # noqa: E402  (numpy only)

frames = np.concatenate([synth.host_block(m, snr, 2, 1024, seed=4100 + 10 * i + j)
                         for i, m in enumerate(synth.MODS6) for j, snr in enumerate((2.0, 18.0))])     # (24, 1024) complex64
X = np.ascontiguousarray(np.stack([frames.real, frames.imag], axis=-1).astype(np.float32))            # (24, 1024, 2)
Y = np.zeros((24, 24), dtype=np.int64)
Y[np.arange(24), np.repeat(np.arange(6), 4)] = 1
Z = np.tile(np.array([2, 2, 18, 18], dtype=np.int64), 6)[:, None]
with h5py.File(HERE / "radioml_like.h5", "w") as fh:
    fh.create_dataset("X", data=X, chunks=(8, 1024, 2), compression="gzip", compression_opts=4, shuffle=True)
    fh.create_dataset("Y", data=Y)
    fh.create_dataset("Z", data=Z, chunks=(16, 1))
meta = {"written_with": f"h5py {h5py.__version__} / HDF5 {h5py.version.hdf5_version} / numpy {np.__version__}",
        "sha256": {k: hashlib.sha256(v.tobytes()).hexdigest() for k, v in (("X", X), ("Y", Y), ("Z", Z))},
        "shape": {"X": list(X.shape), "Y": list(Y.shape), "Z": list(Z.shape)}}
(HERE / "radioml_like.json").write_text(json.dumps(meta, indent=1) + "\n")
print(meta)
