#!/usr/bin/env python3
"""Capture golden vectors from the imported reference  --  TEST INFRASTRUCTURE.

Runs ONLY in the build container, where the reference is mounted read-only at
/root/reference; it writes nothing but inputs and the reference's outputs
(``tests/golden/*.npz`` / ``*.json``).  The GPU box has no reference: tests
there read these fixtures.

    python oracle/capture_golden.py                      # regenerate every fixture
    python oracle/capture_golden.py frames 128 512 8192  # only these frames_n{N}.npz
    python oracle/capture_golden.py edges 1024 4096 8192 # only these edges_n{N}.npz

Fixtures:
  kat_n10.json            the reference's own known-answer vector and table
                          (features.py:240-255, 286-305) restated as data
  frames_n{N}.npz         N in {128, 256, 512, 1000, 1024, 2048, 4096}: 6 mods x 3 SNR x 2 frames
                          (N = 5000, 8192, 16384, 32768: 6 mods x 1 SNR x 2; from 16384 on the fixtures of every kind hold
                          the SHA-256 of their inputs instead of the inputs, which frames_inputs / edges_inputs /
                          range_inputs below rebuild with numpy alone) complex64 inputs (1000 and 5000: the block kernel's
                          two Bluestein forms); golden64 (reference on complex128
                          input, float32-stored as feature_extraction.py:35,56
                          does) + its unrounded float64; golden32 (reference on
                          the complex64 input as is); the 11 moments
  edges_n{N}.npz          N in {1000, 1024, 2048, 4096, 8192}: degenerate frames and what the reference returns
  range_n{N}.npz          N in {2048, 4096, 8192}: ordinary frames at scales 1e-12 ... 1e12 (and mixed-scale
                          ones): the reference's float32-stored outputs incl. their inf / 0 pattern
  range_extreme_n{N}.npz  the same frames at 1e-30, 1e-20, 1e20, 1e30 (the ends of float32)
  extract_roundtrip.npz   a tiny `run_extraction(cfg)` run by the reference:
                          input container + the six output files' contents
  extract_roundtrip_f64.npz  the same on a container of genuine doubles (not float32 casts)
  configs0_reference_run.npz BASELINE configs[0] (6 x 2 x 500 x 2048) through the reference's run_extraction: its outputs
                          and the SHA-256 of the seeded inputs (python oracle/capture_golden.py configs0)
  configs2_reference_run.npz the same at BASELINE configs[2]'s frame size: 6 x 2 x 50 x 4096
  configs4_reference_run.npz ... and at configs[4]'s (RadioML-2018 scale): 6 x 2 x 100 x 1024
  config_defaults.json    field names and defaults of the reference's config layer
"""

from __future__ import annotations

import json
import os
import sys
import tempfile
import warnings
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[1]
REF_SRC = Path("/root/reference/src")
OUT = REPO / "tests" / "golden"

sys.dont_write_bytecode = True
sys.path.insert(0, str(REPO))


def _import_reference():
    if not REF_SRC.exists():
        raise SystemExit("reference not mounted at /root/reference: nothing to capture")
    sys.path.insert(0, str(REF_SRC))
    import amcpy.config as rcfg                       # noqa: E402
    import amcpy.feature_extraction as rfe            # noqa: E402
    import amcpy.features as rfeat                    # noqa: E402
    return rfeat, rfe, rcfg


def _ref18(rfeat, frame, dtype):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with np.errstate(all="ignore"):
            return np.array(rfeat.calculate_features(list(range(1, 19)), frame.astype(dtype)),
                            dtype=np.float64)


def _ref_moments(rfeat, frame):
    m = rfeat.MomentValues(frame.astype(np.complex128))
    return np.array([complex(getattr(m, k)) for k in
                     ("m20", "m21", "m22", "m40", "m41", "m42", "m43", "m60", "m61", "m62", "m63")])


def capture_kat(rfeat):
    sig = rfeat._test_signal()
    # the table the reference asserts at rtol=1e-5 (features.py:286-305)
    expected = [405.0, 0.940293603578649, 1.5903100728408748, 0.3312693299999689,
                0.5153882032022075, 6.363961030678928, 0.7977443845417482,
                1.7757575757575754, 1.0627162629757787, 57.0, 57.0, 3613.8, 3613.8,
                3613.8, 3905583.0, 1094628.0, 311904.0, 1094628.0]
    got = _ref18(rfeat, sig, np.complex128)
    assert np.allclose(got, expected, rtol=1e-5), "reference disagrees with its own table?"
    iv = rfeat.InstantaneousValues(sig)
    mv = rfeat.MomentValues(sig)
    doc = {
        "source": "reference features.py:240-255 (_test_signal), :286-305 (expected), "
                  ":258-271 / :274-280 (spot values)",
        "signal_re": [float(v.real) for v in sig],
        "signal_im": [float(v.imag) for v in sig],
        "expected": expected,
        "rtol": 1e-5,
        "reference_output_float64": [float(v) for v in got],
        "spot": {"abs1": float(iv.abs[1]), "cna0": float(iv.cn_amplitude[0]),
                 "cna_last": float(iv.cn_amplitude[-1]), "len_frequency": int(len(iv.frequency)),
                 "m21": float(mv.m21), "m42": float(mv.m42), "m63": float(np.real(mv.m63))},
    }
    (OUT / "kat_n10.json").write_text(json.dumps(doc, indent=1))


SEEDED_FROM = 8193        # fixtures of frames this long hold no inputs: their SHA-256 and the recipe below rebuild them


def frames_inputs(N):
    """(complex64 (F, N) inputs, tags) of frames_n{N}.npz -- numpy only, no reference: the tests rebuild the inputs of
    the large fixtures with this (tests/conftest.py:load_npz) and check them against the stored SHA-256."""
    from amcpy_amd import synth
    snrs = (-10.0, 4.0, 20.0) if N <= 4096 else (4.0,)        # N = 5000, 8192: 12 frames keep the fixture under 1 MB
    frames, tags = [], []
    for mi, mod in enumerate(synth.MODS6):
        for si, snr in enumerate(snrs):
            blk = synth.host_block(mod, snr, 2, N, seed=1000 + 10 * mi + si)
            frames.append(blk)
            tags += [f"{mod}@{snr:g}dB#{k}" for k in range(2)]
    return np.concatenate(frames, axis=0).astype(np.complex64), tags


def _sha(a) -> str:
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def capture_frames(rfeat, N):
    x, tags = frames_inputs(N)
    g64 = np.stack([_ref18(rfeat, f, np.complex128) for f in x])
    g32 = np.stack([_ref18(rfeat, f, np.complex64) for f in x])
    mom = np.stack([_ref_moments(rfeat, f) for f in x])
    inputs = {"iq": x} if N < SEEDED_FROM else {"iq_sha256": np.array(_sha(x)), "iq_recipe": np.array(f"frames_inputs({N})")}
    np.savez(OUT / f"frames_n{N}.npz", golden64_f64=g64,
             golden64=g64.astype(np.float32), golden32=g32.astype(np.float32),
             moments=mom, tags=np.array(tags), **inputs)


def edge_frames(N):
    n = np.arange(N)
    z = {
        "zeros": np.zeros(N, np.complex64),
        "const_pos": np.ones(N, np.complex64),
        "const_neg": -np.ones(N, np.complex64),
        "const_neg_negzero": (-np.ones(N) - 0j * np.ones(N)).astype(np.complex64),
        "alternating": ((-1.0) ** n).astype(np.complex64),
        "tone_k5": np.exp(2j * np.pi * 5 * n / N).astype(np.complex64),
        "ramp_phase_pi": np.exp(1j * np.pi * n).astype(np.complex64),   # d = +-pi ties
        "real_only": (np.cos(0.37 * n) + 0.5).astype(np.complex64),
        "imag_only": (1j * (np.sin(0.11 * n) + 0.25)).astype(np.complex64),
        "impulse": np.where(n == 3, 1.0, 0.0).astype(np.complex64),
        "huge": (np.exp(0.9j * n) * 3e5).astype(np.complex64),
        "tiny": (np.exp(0.9j * n) * 1e-4 + 1e-5).astype(np.complex64),
    }
    nan = np.exp(0.3j * n).astype(np.complex64)
    nan[7] = np.nan
    z["one_nan"] = nan
    # explicit -0.0 imaginary part on the negative real axis: angle = -pi
    negz = np.empty(N, np.complex64)
    negz.real = -1.0
    negz.imag = -0.0
    z["const_neg_negzero"] = negz
    return z


def edges_inputs(N):
    z = edge_frames(N)
    names = sorted(z)
    return np.stack([z[k] for k in names]), names


def capture_edges(rfeat, N):
    x, names = edges_inputs(N)
    g64 = np.stack([_ref18(rfeat, f, np.complex128) for f in x])
    g32 = np.stack([_ref18(rfeat, f, np.complex64) for f in x])
    inputs = {"iq": x} if N < SEEDED_FROM else {"iq_sha256": np.array(_sha(x)), "iq_recipe": np.array(f"edges_inputs({N})")}
    np.savez(OUT / f"edges_n{N}.npz", names=np.array(names), golden64_f64=g64,
             golden64=g64.astype(np.float32), golden32=g32.astype(np.float32), **inputs)


def capture_roundtrip(rfe, rcfg):
    """The reference's own batch driver on a tiny container whose rows are
    LONGER than frame_size (exercises the [0:frame_size] slice,
    feature_extraction.py:68) and complex128 (MATLAB doubles)."""
    import scipy.io
    from amcpy_amd import synth
    frame_size, row_len, n_frames = 256, 300, 3
    mods = synth.MODS6
    blocks = synth.host_frames(mods, 2, n_frames, row_len)
    with tempfile.TemporaryDirectory() as td:
        paths = rcfg.Paths(root=Path(td))
        sig = rcfg.SignalConfig(snr_values={0: "0", 1: "10"}, num_frames=n_frames,
                                frame_size=frame_size, num_threads=2)
        cfg = rcfg.Config(paths=paths, signals=sig)
        paths.ensure_dirs()
        container = {sig.mat_info[m]: blocks[m].astype(np.complex128) for m in mods}
        scipy.io.savemat(str(paths.mat_data / paths.mat_filename), container)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            rfe.run_extraction(cfg)
        rec = {}
        for m in mods:
            d = scipy.io.loadmat(str(paths.calculated_features / f"{m}_features.mat"))
            keys = sorted(k for k in d if not k.startswith("__"))
            assert keys == sorted(["Modulation", sig.mat_info[m]]), keys
            arr = d[sig.mat_info[m]]
            assert arr.dtype == np.float32 and arr.shape == (2, n_frames, 18)
            rec[f"out_{m}"] = arr
            rec[f"label_{m}"] = np.array(str(np.ravel(d["Modulation"])[0]))
            rec[f"in_{m}"] = blocks[m]
    np.savez(OUT / "extract_roundtrip.npz", frame_size=frame_size, row_len=row_len,
             n_frames=n_frames, mods=np.array(mods), **rec)


def capture_configs0(rfe, rcfg, n_frames=500, fs=2048, name="configs0_reference_run.npz"):
    """BASELINE configs[0] -- the reference's own CPU-runnable case -- by the reference's own batch driver: 6 modulations x
    2 SNR x 500 frames x 2048 samples (synth.host_frames, the seeds of SURVEY.md section 8d) as a container of MATLAB
    doubles, `run_extraction(cfg)` with its default eight threads per modulation, the six {mod}_features.mat it writes.
    The 98 MB input is NOT stored: the test regenerates it from the seeds and checks the SHA-256 recorded here; the
    fixture holds the reference's 6 x (2, 500, 18) float32 outputs (432 KB)."""
    import hashlib
    import time
    import scipy.io
    from amcpy_amd import synth
    n_snr = 2
    mods = synth.MODS6
    blocks = synth.host_frames(mods, n_snr, n_frames, fs)
    with tempfile.TemporaryDirectory() as td:
        paths = rcfg.Paths(root=Path(td))
        sig = rcfg.SignalConfig(snr_values={0: "0", 1: "10"}, num_frames=n_frames, frame_size=fs)
        cfg = rcfg.Config(paths=paths, signals=sig)
        paths.ensure_dirs()
        scipy.io.savemat(str(paths.mat_data / paths.mat_filename),
                         {sig.mat_info[m]: blocks[m].astype(np.complex128) for m in mods})
        t0 = time.time()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            rfe.run_extraction(cfg)
        seconds = time.time() - t0
        rec = {}
        for m in mods:
            d = scipy.io.loadmat(str(paths.calculated_features / f"{m}_features.mat"))
            arr = d[sig.mat_info[m]]
            assert arr.dtype == np.float32 and arr.shape == (n_snr, n_frames, 18) and np.isfinite(arr).all()
            rec[f"out_{m}"] = arr
            rec[f"sha256_in_{m}"] = np.array(hashlib.sha256(np.ascontiguousarray(blocks[m]).tobytes()).hexdigest())
    np.savez(OUT / name, n_snr=n_snr, n_frames=n_frames, frame_size=fs, mods=np.array(mods),
             reference_seconds=seconds, reference_threads=sig.num_threads, **rec)
    print(f"reference run_extraction, 6 x {n_snr} x {n_frames} x {fs}: {seconds:.1f} s here ({6 * n_snr * n_frames / seconds:.0f} frames/s)")


def range_frames(N):
    """Frames of ordinary shape at extraordinary scales (and one whose halves differ by ten
    orders of magnitude): the reference evaluates in complex128 and stores float32
    (features.py:46-58, feature_extraction.py:35,56), so its sixth-order cumulants overflow to
    inf above |x| ~ 2.6e6 and flush to 0 below ~ 3e-8 while the low-order features stay finite."""
    from amcpy_amd import synth
    base = {
        "qpsk10": synth.host_block("QPSK", 10.0, 1, N, seed=4242)[0].astype(np.complex128),
        "qam16_20": synth.host_block("16QAM", 20.0, 1, N, seed=4243)[0].astype(np.complex128),
        "wgn": synth.host_block("WGN", 0.0, 1, N, seed=4244)[0].astype(np.complex128),
    }
    z = {}
    for tag, scale in (("1e-12", 1e-12), ("1e-8", 1e-8), ("1e-6", 1e-6), ("1e-3", 1e-3),
                       ("1e3", 1e3), ("1e6", 1e6), ("1e7", 1e7), ("1e12", 1e12)):
        for k, v in base.items():
            z[f"{k}_x{tag}"] = (v * scale).astype(np.complex64)
    burst = base["qpsk10"].copy()
    burst[: N // 2] *= 1e-3
    burst[N // 2:] *= 1e7
    z["burst_1e-3_then_1e7"] = burst.astype(np.complex64)
    spike = base["wgn"].copy()
    spike[N // 3] = 5e7 + 2e7j                     # one sample whose sixth power leaves float32
    z["unit_noise_one_spike_5e7"] = spike.astype(np.complex64)
    return z


def range_extreme_frames(N):
    """The same three frames at the ends of float32: 1e-30, 1e-20, 1e20, 1e30 (|x|^2 itself leaves float32)."""
    from amcpy_amd import synth
    base = {
        "qpsk10": synth.host_block("QPSK", 10.0, 1, N, seed=4242)[0].astype(np.complex128),
        "qam16_20": synth.host_block("16QAM", 20.0, 1, N, seed=4243)[0].astype(np.complex128),
        "wgn": synth.host_block("WGN", 0.0, 1, N, seed=4244)[0].astype(np.complex128),
    }
    return {f"{k}_x{tag}": (v * scale).astype(np.complex64)
            for tag, scale in (("1e-30", 1e-30), ("1e-20", 1e-20), ("1e20", 1e20), ("1e30", 1e30))
            for k, v in base.items()}


def range_extreme_inputs(N):
    ze = range_extreme_frames(N)
    names_e = sorted(ze)
    return np.stack([ze[k] for k in names_e]), names_e


def range_inputs(N):
    z = range_frames(N)
    names = sorted(z)
    return np.stack([z[k] for k in names]), names


def capture_range(rfeat, N):
    xe, names_e = range_extreme_inputs(N)
    ge = np.stack([_ref18(rfeat, f, np.complex128) for f in xe])
    inputs = {"iq": xe} if N < SEEDED_FROM else {"iq_sha256": np.array(_sha(xe)), "iq_recipe": np.array(f"range_extreme_inputs({N})")}
    with np.errstate(all="ignore"):
        np.savez(OUT / f"range_extreme_n{N}.npz", names=np.array(names_e), golden64_f64=ge,
                 golden64=ge.astype(np.float32), **inputs)
    x, names = range_inputs(N)
    g64 = np.stack([_ref18(rfeat, f, np.complex128) for f in x])
    with np.errstate(all="ignore"):
        g32 = g64.astype(np.float32)               # what feature_extraction.py:35,56 stores: inf / 0 at the ends
    inputs = {"iq": x} if N < SEEDED_FROM else {"iq_sha256": np.array(_sha(x)), "iq_recipe": np.array(f"range_inputs({N})")}
    np.savez(OUT / f"range_n{N}.npz", names=np.array(names), golden64_f64=g64, golden64=g32, **inputs)


def capture_roundtrip_f64(rfe, rcfg):
    """run_extraction by the reference on a container of GENUINE doubles (samples that are
    not float32-representable: MATLAB's randn + double arithmetic): the reference evaluates
    them in complex128 (feature_extraction.py:46-48,68), the engine rounds to complex64 first.
    Rows longer than frame_size, 6 mods x 2 SNR x 3 frames x 1024."""
    import scipy.io
    from amcpy_amd import synth
    frame_size, row_len, n_frames = 1024, 1100, 3
    mods = synth.MODS6
    rng = np.random.default_rng(20261004)
    blocks = {}
    for mi, m in enumerate(mods):
        rows = []
        for si, snr in enumerate((0.0, 10.0)):
            sig = synth.host_block(m, 300.0, n_frames, row_len, seed=7000 + 10 * mi + si).astype(np.complex128)
            sigma = 10.0 ** (-snr / 20.0)
            noise = (rng.standard_normal((n_frames, row_len)) + 1j * rng.standard_normal((n_frames, row_len))) \
                * (sigma / np.sqrt(2.0))
            gain = 1.0 + 0.37 * rng.standard_normal()          # an irrational-looking double scale
            rows.append((sig + noise) * gain * np.exp(1j * rng.uniform(0, 2 * np.pi)))
        blocks[m] = np.stack(rows)
        assert not np.array_equal(blocks[m], blocks[m].astype(np.complex64).astype(np.complex128))
    with tempfile.TemporaryDirectory() as td:
        paths = rcfg.Paths(root=Path(td))
        sig = rcfg.SignalConfig(snr_values={0: "0", 1: "10"}, num_frames=n_frames,
                                frame_size=frame_size, num_threads=2)
        cfg = rcfg.Config(paths=paths, signals=sig)
        paths.ensure_dirs()
        scipy.io.savemat(str(paths.mat_data / paths.mat_filename),
                         {sig.mat_info[m]: blocks[m] for m in mods})
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            rfe.run_extraction(cfg)
        rec = {}
        for m in mods:
            d = scipy.io.loadmat(str(paths.calculated_features / f"{m}_features.mat"))
            arr = d[sig.mat_info[m]]
            assert arr.dtype == np.float32 and arr.shape == (2, n_frames, 18)
            rec[f"out_{m}"] = arr
            rec[f"in_{m}"] = blocks[m]
    np.savez(OUT / "extract_roundtrip_f64.npz", frame_size=frame_size, row_len=row_len,
             n_frames=n_frames, mods=np.array(mods), **rec)


def capture_config_defaults(rcfg):
    """Field names and default values of the reference's configuration layer
    (config.py:15-186), as data: pins the mirror in amcpy_amd/config.py."""
    import dataclasses
    cfg = rcfg.Config(paths=rcfg.Paths(root=Path("/project")))
    doc = {}
    for grp in ("paths", "signals", "features", "training"):
        obj = getattr(cfg, grp)
        doc[grp] = {f.name: (str(v) if isinstance(v, Path) else v)
                    for f in dataclasses.fields(obj) for v in [getattr(obj, f.name)]}
    doc["features"]["names"] = {str(k): v for k, v in rcfg.FeatureConfig.names.items()}
    doc["features"]["used_names"] = cfg.features.used_names
    doc["features"]["num_used"] = cfg.features.num_used
    doc["training"]["feature_files"] = cfg.training.feature_files
    (OUT / "config_defaults.json").write_text(json.dumps(doc, indent=1, default=list))


def main():
    os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
    OUT.mkdir(parents=True, exist_ok=True)
    rfeat, rfe, rcfg = _import_reference()
    if len(sys.argv) > 2 and sys.argv[1] == "frames":
        for N in map(int, sys.argv[2:]):
            capture_frames(rfeat, N)
            p = OUT / f"frames_n{N}.npz"
            print(f"{p.name:28s} {p.stat().st_size:9d} B")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "configs0":
        capture_configs0(rfe, rcfg)
        capture_configs0(rfe, rcfg, n_frames=50, fs=4096, name="configs2_reference_run.npz")
        capture_configs0(rfe, rcfg, n_frames=100, fs=1024, name="configs4_reference_run.npz")
        for name in ("configs0_reference_run.npz", "configs2_reference_run.npz", "configs4_reference_run.npz"):
            print(f"{name:28s} {(OUT / name).stat().st_size:9d} B")
        return
    if len(sys.argv) > 2 and sys.argv[1] == "edges":
        for N in map(int, sys.argv[2:]):
            capture_edges(rfeat, N)
            p = OUT / f"edges_n{N}.npz"
            print(f"{p.name:28s} {p.stat().st_size:9d} B")
        return
    if len(sys.argv) > 2 and sys.argv[1] == "range":            # range 4096 8192: only these sizes' range fixtures
        for N in map(int, sys.argv[2:]):
            capture_range(rfeat, N)
            for name in (f"range_n{N}.npz", f"range_extreme_n{N}.npz"):
                print(f"{name:28s} {(OUT / name).stat().st_size:9d} B")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "range":
        capture_range(rfeat, 2048)
        capture_roundtrip_f64(rfe, rcfg)
        for name in ("range_n2048.npz", "range_extreme_n2048.npz", "extract_roundtrip_f64.npz"):
            print(f"{name:28s} {(OUT / name).stat().st_size:9d} B")
        return
    capture_kat(rfeat)
    for N in (128, 256, 512, 1000, 1024, 2048, 4096, 5000, 8192, 10000, 12289, 16384, 32767, 32768):
        capture_frames(rfeat, N)
    for N in (1000, 1024, 2048, 4096, 8192, 10000, 16384, 32767, 32768):
        capture_edges(rfeat, N)
    for N in (2048, 4096, 8192, 10000, 16384, 32768):
        capture_range(rfeat, N)
    capture_roundtrip(rfe, rcfg)
    capture_roundtrip_f64(rfe, rcfg)
    capture_configs0(rfe, rcfg)
    capture_configs0(rfe, rcfg, n_frames=50, fs=4096, name="configs2_reference_run.npz")
    capture_configs0(rfe, rcfg, n_frames=100, fs=1024, name="configs4_reference_run.npz")
    capture_config_defaults(rcfg)
    for p in sorted(OUT.iterdir()):
        print(f"{p.name:28s} {p.stat().st_size:9d} B")


if __name__ == "__main__":
    main()
