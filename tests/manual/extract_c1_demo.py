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
              f"{time.time() - t0:.2f} s wall including interpreter start-up and `import torch`")
