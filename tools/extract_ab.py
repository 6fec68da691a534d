#!/usr/bin/env python3
"""run_extraction on the 2.6 GB uncompressed container (6 x (26, 512, 2048) complex128 in /dev/shm), five times in a
fresh process per setting of an environment switch: an A/B of host-side choices on one box.
    python tools/extract_ab.py AMCX_DIRECT_FILE 0 1
    python tools/extract_ab.py AB_THREADS 4 8 12 16        (SignalConfig.num_threads = staging threads)
    python tools/extract_ab.py --compressed AMCX_READ_AHEAD 1 3 6"""
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    from amcpy_amd.config import Config, Paths, SignalConfig
    from amcpy_amd.feature_extraction import run_extraction
    cfg = Config(paths=Paths(root=Path(sys.argv[2])),
                 signals=SignalConfig(snr_values={i: str(v) for i, v in enumerate(range(-20, 32, 2))}, num_frames=512,
                                      num_threads=int(os.environ.get("AB_THREADS", "8"))))
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); run_extraction(cfg, verbose=False); ts.append((time.perf_counter() - t0) * 1e3)
    print("   run_extraction ms:", " ".join(f"{t:.1f}" for t in ts), f"| best {6 * 26 * 512 * 2048 * 16 / min(ts) / 1e6:.1f} GB/s of samples")
    sys.exit(0)

import numpy as np
import scipy.io
from amcpy_amd import synth
from amcpy_amd.config import Config, Paths, SignalConfig

compress = "--compressed" in sys.argv          # MATLAB's default save: every variable one deflate stream
args = [a for a in sys.argv[1:] if a != "--compressed"]
var, values = (args[0], args[1:]) if len(args) > 1 else ("AMCX_DIRECT_FILE", ["0", "1"])
for f in ("enabled", "defrag", "shmem_enabled"):
    p = Path("/sys/kernel/mm/transparent_hugepage") / f
    print(f"transparent_hugepage/{f}: {p.read_text().strip() if p.exists() else 'absent'}")
blocks = synth.host_frames(synth.MODS6, 2, 500, 2048)
big = {m: np.asfortranarray(np.tile(blocks[m].astype(np.complex128)[:, :256], (13, 2, 1))) for m in synth.MODS6}
with tempfile.TemporaryDirectory(dir="/dev/shm" if Path("/dev/shm").is_dir() else None) as td:
    cfg = Config(paths=Paths(root=Path(td)),
                 signals=SignalConfig(snr_values={i: str(v) for i, v in enumerate(range(-20, 32, 2))}, num_frames=512))
    cfg.paths.ensure_dirs()
    scipy.io.savemat(str(cfg.paths.mat_data / cfg.paths.mat_filename), {cfg.signals.mat_info[m]: big[m] for m in synth.MODS6},
                     do_compression=compress)
    print(f"container: {(cfg.paths.mat_data / cfg.paths.mat_filename).stat().st_size / 1e9:.2f} GB on disk, compressed: {compress}")
    del big
    for rep in range(2):
        for v in values:
            print(f"{var}={v}", flush=True)
            subprocess.run([sys.executable, __file__, "--child", td], env={**os.environ, var: v}, check=False)
