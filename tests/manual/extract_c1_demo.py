#!/usr/bin/env python3
"""BASELINE configs[0] end to end on a GPU box, timed: synthesise the reference's container
(6 mods x 2 SNR x 500 frames x 2048, complex64 and complex128 .mat), run `run_extraction` in
process (cold, then warm) and `python -m amcpy_amd extract` as a subprocess.  The collected test
`test_extract_cli_on_the_configs0_shape` checks the values; this prints the wall times that
DESIGN.md section 5 quotes beside the reference's 15.99 s on 8 cores."""
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import scipy.io

REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO))
from amcpy_amd import synth  # noqa: E402
from amcpy_amd.config import Config, Paths, SignalConfig  # noqa: E402
from amcpy_amd.feature_extraction import run_extraction  # noqa: E402

blocks = synth.host_frames(synth.MODS6, 2, 500, 2048)
for dtype in (np.complex64, np.complex128):
    with tempfile.TemporaryDirectory() as td:
        cfg = Config(paths=Paths(root=Path(td)), signals=SignalConfig(snr_values={0: "0", 1: "10"}, num_frames=500))
        cfg.paths.ensure_dirs()
        scipy.io.savemat(str(cfg.paths.mat_data / cfg.paths.mat_filename),
                         {cfg.signals.mat_info[m]: blocks[m].astype(dtype) for m in synth.MODS6})
        size_mb = (cfg.paths.mat_data / cfg.paths.mat_filename).stat().st_size / 1e6
        for label in ("cold", "warm"):
            t0 = time.time()
            run_extraction(cfg, verbose=False)
            print(f"{np.dtype(dtype).name:10s} container ({size_mb:.0f} MB): run_extraction {label}: {time.time() - t0:.2f} s "
                  f"for 6000 frames (loadmat + gather + upload + kernels + savemat)")
        t0 = time.time()
        r = subprocess.run([sys.executable, "-m", "amcpy_amd", "extract", "--root", td, "--num-frames", "500",
                            "--snr-values", "0", "10"], cwd=str(REPO), capture_output=True, text=True)
        print(f"{np.dtype(dtype).name:10s} `python -m amcpy_amd extract` as a subprocess: rc {r.returncode}, "
              f"{time.time() - t0:.2f} s wall including interpreter start-up and HIP initialisation (torch is not imported)")

# ---- round 3: a container at scale -- 6 modulations x (26, 512, 2048) complex128 = 2.6 GB in the file -------------------
# How long does the container take to get from the file to the features, against decoding it with scipy alone?
big = {m: np.asfortranarray(np.tile(blocks[m].astype(np.complex128)[:, :256], (13, 2, 1))) for m in synth.MODS6}   # (26, 512, 2048)
shm = "/dev/shm" if Path("/dev/shm").is_dir() else None
for compress in (False, True):
    with tempfile.TemporaryDirectory(dir=shm) as td:
        cfg = Config(paths=Paths(root=Path(td)),
                     signals=SignalConfig(snr_values={i: str(v) for i, v in enumerate(range(-20, 32, 2))}, num_frames=512))
        cfg.paths.ensure_dirs()
        path = cfg.paths.mat_data / cfg.paths.mat_filename
        t0 = time.time()
        scipy.io.savemat(str(path), {cfg.signals.mat_info[m]: big[m] for m in synth.MODS6}, do_compression=compress)
        t_save = time.time() - t0
        size_gb = path.stat().st_size / 1e9
        t0 = time.time()
        scipy.io.loadmat(str(path), variable_names=[cfg.signals.mat_info["BPSK"]])
        t_one = time.time() - t0
        for label in ("first", "again"):
            t0 = time.time()
            run_extraction(cfg, verbose=False)
            dt = time.time() - t0
            print(f"{'compressed' if compress else 'uncompressed'} container {size_gb:.2f} GB (savemat took {t_save:.1f} s; scipy.io.loadmat of ONE "
                  f"of its six variables {t_one:.2f} s): run_extraction {label}: {dt:.2f} s for {6 * 26 * 512} frames "
                  f"= {6 * 26 * 512 / dt / 1e3:.0f} k frames/s, {6 * 26 * 512 * 2048 * 16 / dt / 1e9:.1f} GB/s of samples, file in -> six files out")
