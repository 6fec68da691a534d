#!/usr/bin/env python3
"""A BASELINE configs[1] modulation as the REAL container MATLAB would write it -- 3.49 GB of complex128 is above the 2 GB that
level-5 files hold, so it can only be a -v7.3 (HDF5) file -- through matfile.load_variable + the engine (GPU box).  Writes
(26, frames, 2048) complex128 with amcpy_amd.hdf5_min in MATLAB's layout into /dev/shm (or the temp dir): contiguous
(`save -v7.3 -nocompression` of a large array / any non-chunking writer), and chunked + deflate (MATLAB's default), runs the
upload-and-extract twice each and prints frames/s and GB/s of container bytes; then the same arrays as a level-5 variable where
that is possible (frames <= 2048: below 2 GB) for comparison.
    python tools/mat73_ingest_probe.py [frames=4096]"""
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from amcpy_amd import hdf5_min, matfile, synth  # noqa: E402
from amcpy_amd.feature_extraction import FrameRows, HipEngine  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S, N = 26, 2048
rng = np.random.default_rng(5)
block = np.concatenate([synth.host_block(m, 6.0, 256, N, seed=11 + i) for i, m in enumerate(synth.MODS6)]).astype(np.complex128)
block *= 1.0 + 1e-9 * rng.standard_normal(block.shape)          # genuine doubles
x = np.asfortranarray(block[np.arange(S * K) % block.shape[0]].reshape(S, K, N))
print(f"library {hdf5_min.library_path()}; ({S}, {K}, {N}) complex128 = {x.nbytes / 1e9:.2f} GB")
root = "/dev/shm" if Path("/dev/shm").is_dir() else None
eng = HipEngine(N, 0)
ref = None
with tempfile.TemporaryDirectory(dir=root) as d:
    for label, kw in (("v7.3 contiguous", {}), ("v7.3 chunked (256, 64, 2) + deflate 1", {"chunks": (256, 64, 2), "deflate": 1})):
        path = Path(d) / "m.mat"
        t0 = time.perf_counter()
        with hdf5_min.File(path, "w", userblock=512) as fh:
            fh.write_mat73_variable("signal_bpsk", x, **kw)
        hdf5_min.write_matlab_header(path)
        print(f"  wrote {label}: {path.stat().st_size / 1e9:.2f} GB in {time.perf_counter() - t0:.1f} s")
        for run in range(2):
            t0 = time.perf_counter()
            v = matfile.load_variable(path, "signal_bpsk", None, True)
            t1 = time.perf_counter()
            out = eng(FrameRows(v, S, K))
            dt = time.perf_counter() - t0
            getattr(v, "release", lambda: None)()
            print(f"  {label:40s} run {run}: locate / decode {1e3 * (t1 - t0):8.1f} ms, total {dt * 1e3:8.1f} ms = "
                  f"{S * K / dt / 1e6:6.3f} M frames/s, {x.nbytes / dt / 1e9:6.2f} GB/s of container bytes  ({type(v).__name__})")
        if ref is None:
            ref = out
        assert np.array_equal(out.view(np.int32), ref.view(np.int32)), "the two layouts give different features"
        path.unlink()
    if x.nbytes < (2 << 30):
        import scipy.io
        path = Path(d) / "m5.mat"
        scipy.io.savemat(str(path), {"signal_bpsk": x})
        for run in range(2):
            t0 = time.perf_counter()
            v = matfile.load_variable(path, "signal_bpsk", None, True)
            out = eng(FrameRows(v, S, K))
            dt = time.perf_counter() - t0
            getattr(v, "release", lambda: None)()
            print(f"  {'level 5 (split real / imaginary arrays)':40s} run {run}: total {dt * 1e3:8.1f} ms = {S * K / dt / 1e6:6.3f} M frames/s, "
                  f"{x.nbytes / dt / 1e9:6.2f} GB/s of container bytes")
        assert np.array_equal(out.view(np.int32), ref.view(np.int32))
eng.close()
print("features identical across the layouts: yes")
