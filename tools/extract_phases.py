#!/usr/bin/env python3
"""Where run_extraction's time goes on a container at scale (6 x (26, 512, 2048) complex128, uncompressed, in
/dev/shm): per modulation the reader (map + pre-fault), the engine call with its phase timers, savemat.
    python tools/extract_phases.py"""
import sys
import tempfile
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import scipy.io
from amcpy_amd import synth
from amcpy_amd.config import Config, Paths, SignalConfig
from amcpy_amd.feature_extraction import FrameRows, HipEngine, _load_variable, run_extraction

blocks = synth.host_frames(synth.MODS6, 2, 500, 2048)
big = {m: np.asfortranarray(np.tile(blocks[m].astype(np.complex128)[:, :256], (13, 2, 1))) for m in synth.MODS6}
with tempfile.TemporaryDirectory(dir="/dev/shm" if Path("/dev/shm").is_dir() else None) as td:
    cfg = Config(paths=Paths(root=Path(td)),
                 signals=SignalConfig(snr_values={i: str(v) for i, v in enumerate(range(-20, 32, 2))}, num_frames=512))
    cfg.paths.ensure_dirs()
    path = cfg.paths.mat_data / cfg.paths.mat_filename
    scipy.io.savemat(str(path), {cfg.signals.mat_info[m]: big[m] for m in synth.MODS6})
    run_extraction(cfg, verbose=False)
    t0 = time.perf_counter(); run_extraction(cfg, verbose=False); print(f"run_extraction: {(time.perf_counter() - t0) * 1e3:.1f} ms")
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable(); run_extraction(cfg, verbose=False); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(8)
    for _ in range(3):
        t0 = time.perf_counter(); run_extraction(cfg, verbose=False); print(f"run_extraction: {(time.perf_counter() - t0) * 1e3:.1f} ms")
    eng = HipEngine(2048, threads=8)
    from amcpy_amd.matfile import BufferPool
    pool = BufferPool()
    for rep in range(5):
        how = None if rep == 0 else pool                  # first round: memory-mapped; then: preadv into reused buffers; then: located only
        direct = rep >= 3
        print("--- variables", "located in the file, read by the engine's staging threads" if direct else
              "memory-mapped" if how is None else "read with preadv into a buffer pool")
        for m in synth.MODS6:
            key = cfg.signals.mat_info[m]
            t0 = time.perf_counter(); v = _load_variable(path, key, how, direct); t1 = time.perf_counter()
            feats = eng(FrameRows(v, 26, 512)); t2 = time.perf_counter()
            scipy.io.savemat(str(cfg.paths.calculated_features / f"{m}_features.mat"), {"Modulation": m, key: feats.reshape(26, 512, 18)})
            t3 = time.perf_counter()
            getattr(v, "release", lambda: None)()
            st = eng.stats
            print(f"{m:6s} read / map {(t1 - t0) * 1e3:6.1f} ms | engine {(t2 - t1) * 1e3:6.1f} ms (native {st['seconds_native'] * 1e3:.1f}: staging "
                  f"{st['seconds_staging'] * 1e3:.1f}, waiting {st['seconds_waiting'] * 1e3:.1f}, tail {st['seconds_tail'] * 1e3:.1f}, prepare "
                  f"{st['seconds_prepare'] * 1e3:.1f}; {st['source_bytes'] / (t2 - t1) / 1e9:.0f} GB/s) | savemat {(t3 - t2) * 1e3:5.1f} ms")
