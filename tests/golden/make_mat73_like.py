#!/usr/bin/env python3
"""Writes tests/golden/mat73_like.mat with the REAL h5py, in the layout MATLAB's `save -v7.3` uses (an HDF5 file behind
a 512-byte MATLAB header): one root-level dataset per variable, dimensions REVERSED (a MATLAB (n_snr, n_frames, L)
array is an HDF5 (L, n_frames, n_snr) dataset -- its bytes are the column-major variable), complex numbers as the
compound {real, imag}, a `MATLAB_class` string attribute.  The reference hands such a file to scipy.io.loadmat, which
refuses it (feature_extraction.py:46-47) -- but a BASELINE configs[1] modulation is 3.49 GB of complex128, and MATLAB
stores variables above 2 GB only this way.

The six modulation variables of the reference's container (config.py:55-63 names) at (2 SNR, 5 frames, 300 samples),
stored the three ways MATLAB does -- chunked + deflate (its default), chunked without compression (`-nocompression`),
contiguous -- plus a `single` complex variable, a real one, and a `char` variable (unsupported: must fall through to
the reader's own error).  mat73_like.json holds the SHA-256 of each numeric array's bytes in MATLAB's (column-major)
order, as they were BEFORE h5py wrote them.

h5py is not importable by the image's system interpreter; its conda interpreter has it:

    /opt/conda/bin/python3.9 tests/golden/make_mat73_like.py
"""
import hashlib
import json
import sys
from pathlib import Path

import h5py
import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parents[1]))
from amcpy_amd import synth  # noqa: E402  (numpy only)

NAMES = {"BPSK": "signal_bpsk", "QPSK": "signal_qpsk", "8PSK": "signal_8psk", "16QAM": "signal_qam16",
         "64QAM": "signal_qam64", "WGN": "signal_noise"}
N_SNR, N_FRAMES, L = 2, 5, 300


def variable(mi: int, mod: str) -> np.ndarray:
    """(n_snr, n_frames, L) complex128, genuine doubles (not float32 values widened)."""
    rows = [synth.host_block(mod, snr, N_FRAMES, L, seed=7300 + 10 * mi + si).astype(np.complex128) * (1.0 + 1e-9 * (mi + 1))
            for si, snr in enumerate((0.0, 10.0))]
    return np.stack(rows)


def store(fh, name, arr, cls, **kw):
    """arr in MATLAB's shape; the dataset gets the reversed shape, i.e. arr's column-major bytes."""
    if np.iscomplexobj(arr):
        part = np.float64 if arr.dtype == np.complex128 else np.float32
        pair = np.dtype([("real", part), ("imag", part)])
        data = np.empty(arr.shape[::-1], dtype=pair)
        data["real"] = arr.real.T
        data["imag"] = arr.imag.T
    else:
        data = np.ascontiguousarray(arr.T)
    ds = fh.create_dataset(name, data=data, **kw)
    ds.attrs["MATLAB_class"] = np.bytes_(cls)
    return ds


path = HERE / "mat73_like.mat"
sha, layouts = {}, {}
with h5py.File(path, "w", userblock_size=512, libver="earliest") as fh:
    ways = [dict(chunks=(100, 5, 2), compression="gzip", compression_opts=3), dict(chunks=(150, 5, 1)), dict()]
    for mi, (mod, name) in enumerate(NAMES.items()):
        arr = variable(mi, mod)
        kw = ways[mi % 3]
        store(fh, name, arr, "double", **kw)
        sha[name] = hashlib.sha256(np.asfortranarray(arr).tobytes(order="F")).hexdigest()
        layouts[name] = "contiguous" if not kw else ("chunked+deflate" if "compression" in kw else "chunked")
    single = variable(0, "BPSK").astype(np.complex64)
    store(fh, "signal_single", single, "single")
    sha["signal_single"] = hashlib.sha256(np.asfortranarray(single).tobytes(order="F")).hexdigest()
    real_only = variable(1, "QPSK").real.copy()
    store(fh, "signal_real", real_only, "double")
    sha["signal_real"] = hashlib.sha256(np.asfortranarray(real_only).tobytes(order="F")).hexdigest()
    # not MATLAB's: a chunked + deflate dataset of which only two chunks were ever written, with a non-zero fill value --
    # the rows nobody wrote must read as the fill value through every path of the reader
    part = fh.create_dataset("partly_written", shape=(40, 6), dtype=np.float32, chunks=(8, 6), compression="gzip", fillvalue=2.5)
    rng = np.random.default_rng(73)
    expect = np.full((40, 6), 2.5, dtype=np.float32)
    for a in (8, 32):
        expect[a:a + 8] = rng.standard_normal((8, 6)).astype(np.float32)
        part[a:a + 8] = expect[a:a + 8]
    sha["partly_written"] = hashlib.sha256(expect.tobytes()).hexdigest()
    text = np.frombuffer("not a signal".encode("utf-16-le"), dtype=np.uint16).reshape(-1, 1)
    store(fh, "note", text.T, "char")
header = ("MATLAB 7.3 MAT-file, Platform: GLNXA64, Created on: Sun Oct  4 12:00:00 2026 HDF5 schema 1.00 .").encode("ascii")
block = bytearray(b" " * 116 + b"\0" * 8 + bytes([0x00, 0x02]) + b"IM" + b"\0" * 384)
block[:len(header)] = header
with open(path, "r+b") as fh:
    fh.write(bytes(block))
meta = {"written_with": f"h5py {h5py.__version__} / HDF5 {h5py.version.hdf5_version} / numpy {np.__version__}",
        "shape": [N_SNR, N_FRAMES, L], "sha256_column_major": sha, "layout": layouts}
(HERE / "mat73_like.json").write_text(json.dumps(meta, indent=1) + "\n")
print(meta, path.stat().st_size)
