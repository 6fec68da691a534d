"""CPU-only tests of the host side: the C-ABI library loads and exports every
symbol include/amcx.h declares, the config mirror keeps the reference's
attribute paths and defaults, frame sharding, and the world_size-2 gather over
gloo.  No GPU compute is called here."""
import ctypes
import json
import os
import re
import socket
import subprocess
import sys
import textwrap
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parents[1]


def test_library_exports_every_declared_symbol():
    from amcpy_amd import _lib
    header = (REPO / "include" / "amcx.h").read_text()
    declared = set(re.findall(r"\b(amcx_[a-z0-9_]+)\s*\(", header))
    assert declared, "no prototypes found in include/amcx.h"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"libamcx.so does not export {name}"
    assert lib.amcx_abi_version() == _lib.ABI_VERSION
    m = re.search(r"#define\s+AMCX_ABI_VERSION\s+(\d+)", header)
    assert int(m.group(1)) == _lib.ABI_VERSION
    for code in (0, -1, -2, -3, -4, -5, -99):
        assert lib.amcx_strerror(code)            # never NULL


def test_argument_validation_needs_no_gpu():
    """Validation happens before any HIP call, so these are safe without a GPU."""
    from amcpy_amd import _lib
    lib = _lib.load()
    f = lib.amcx_features18_c64
    assert f(None, -1, 2048, 2048, None, 18, None) == _lib.EINVAL
    assert f(None, 4, 2048, 100, None, 18, None) == _lib.EINVAL       # stride < frame_size
    assert f(None, 4, 2048, 2048, None, 17, None) == _lib.EINVAL      # out stride < 18
    assert f(None, 4, 1, 2048, None, 18, None) == _lib.EINVAL         # frame_size < 2
    assert f(None, 4, 1 << 20, 1 << 20, None, 18, None) == _lib.EINVAL
    assert f(None, 0, 2048, 2048, None, 18, None) == _lib.OK          # empty batch is a no-op
    assert f(None, 4, 2048, 2048, None, 18, None) == _lib.EINVAL      # null buffers
    assert lib.amcx_features18_c64_ex(None, 0, 1000, 1000, None, 18, None, _lib.VARIANT_WAVE) == _lib.ENOTSUP
    assert lib.amcx_features18_c64_ex(None, 0, 2048, 2048, None, 18, None, 7) == _lib.EINVAL
    assert _lib.kernel_name(2048).startswith("amcx_features18_wave_kernel")
    assert _lib.kernel_name(1000) == "amcx_features18_block_kernel<2>"
    # ABI 6: every size up to 32768 has a kernel; the any-size path above 8192 samples says what workspace its FFT form wants
    assert _lib.kernel_name(10000).startswith("amcx_features18_stream_kernel")
    wsb = lib.amcx_features18_workspace_bytes
    assert wsb(2048, 100, _lib.VARIANT_AUTO) == 0 and wsb(8191, 100, _lib.VARIANT_BLOCK) == 0
    assert wsb(16384, 4, _lib.VARIANT_AUTO) == 0                       # the group kernel needs none
    assert wsb(16384, 4, _lib.VARIANT_BLOCK) == 4 * 16384 * 8          # the plain transform: one buffer per frame in flight
    assert wsb(10000, 4, _lib.VARIANT_AUTO) == (4 + 1) * 32768 * 8     # Bluestein: M = 32768, + the chirp's spectrum
    assert wsb(20000, 3, _lib.VARIANT_AUTO) == (3 + 1) * 65536 * 8
    assert wsb(10000, 0, _lib.VARIANT_AUTO) == 0
    assert wsb(40000, 4, _lib.VARIANT_AUTO) == -1 and wsb(10000, -1, _lib.VARIANT_AUTO) == -1
    assert wsb(10000, 4, _lib.VARIANT_WAVE) == -1                      # no such kernel: the call itself would refuse
    assert lib.amcx_features18_c64_ws(None, 4, 10000, 10000, None, 18, None, _lib.VARIANT_AUTO, None, 0) == _lib.EINVAL
    assert lib.amcx_features18_c64_ws(None, 0, 10000, 10000, None, 18, None, _lib.VARIANT_AUTO, None, 0) == _lib.OK
    with pytest.raises(ValueError):
        _lib.check(_lib.EINVAL)
    with pytest.raises(_lib.AmcxError):
        _lib.check(_lib.ENOTSUP)


def test_no_cpu_fallback_anywhere():
    """The product path must fail loudly without the HIP library / a GPU: nothing
    under amcpy_amd/ may import the oracle, and the host-buffer entry point
    reports ENODEV instead of computing on the CPU."""
    for p in (REPO / "amcpy_amd").rglob("*.py"):
        src = p.read_text()
        assert "oracle" not in src.replace("# oracle", ""), f"{p} mentions the oracle"
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: ENODEV path not reachable")
    from amcpy_amd import _lib
    from amcpy_amd.features import calculate_features, features18_host
    x = np.ones(64, np.complex64)
    with pytest.raises(_lib.AmcxError) as ei:
        features18_host(x[None, :])
    assert ei.value.code == _lib.ENODEV
    with pytest.raises(_lib.AmcxError):
        calculate_features([1, 2], x)
    with pytest.raises(KeyError):                 # id check comes first, as in the reference
        calculate_features([0], x)


def test_config_mirror_defaults_and_paths(tmp_path):
    from amcpy_amd.config import Config, FeatureConfig, Paths, SignalConfig
    cfg = Config()
    s = cfg.signals
    assert s.frame_size == 2048 and s.num_frames == 1000 and s.num_threads == 8
    assert s.modulations_with_noise == ("BPSK", "QPSK", "8PSK", "16QAM", "64QAM", "WGN")
    assert len(s.snr_values) == 16 and s.snr_values[0] == "-10" and s.snr_values[15] == "20"
    assert s.mat_info["16QAM"] == "signal_qam16" and s.mat_info["WGN"] == "signal_noise"
    assert cfg.features.all_features == tuple(range(1, 19))
    assert cfg.features.used == (2, 4, 6, 8, 12, 14) and cfg.features.num_used == 6
    assert FeatureConfig.names[14] == r"$C_{42}$" and FeatureConfig.names[1] == r"$\gamma_{max}$"
    p = Paths(root=tmp_path)
    assert p.mat_data == tmp_path / "mat-data" and p.mat_filename == "all_modulations.mat"
    assert p.calculated_features == tmp_path / "calculated-features"
    p.ensure_dirs()
    assert p.feature_figures.is_dir() and p.trained_ann.is_dir()
    custom = SignalConfig(snr_values={0: "0", 1: "10"}, num_frames=500)
    assert len(custom.snr_values) == 2 and custom.num_frames == 500
    with pytest.raises(Exception):                # frozen
        cfg.signals.frame_size = 1024


def test_shard_range_partitions_exactly():
    from amcpy_amd.sharding import shard_range
    for F in (0, 1, 7, 48, 638976, 10223616):
        for W in (1, 2, 3, 4, 8):
            cuts = [shard_range(F, r, W) for r in range(W)]
            assert cuts[0][0] == 0 and cuts[-1][1] == F
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(s for s in sizes if s or F < W) <= -(-F // W)
    with pytest.raises(ValueError):
        shard_range(10, 4, 4)


def test_run_extraction_host_logic_with_stub_engine(tmp_path):
    """Container in, six files out, keys/dtype/shape/slicing as the reference
    writes them -- with a stub standing in for the GPU so this runs anywhere.
    (The stub is the oracle: test infrastructure, injected only here.)"""
    import scipy.io
    from amcpy_amd.config import Config, Paths, SignalConfig
    from amcpy_amd.feature_extraction import run_extraction
    from oracle import iq_features_oracle as orc
    from tests.conftest import load_npz

    g = load_npz("extract_roundtrip.npz")
    fs, n_frames = int(g["frame_size"]), int(g["n_frames"])
    mods = [str(m) for m in g["mods"]]
    cfg = Config(paths=Paths(root=tmp_path),
                 signals=SignalConfig(snr_values={0: "0", 1: "10"}, num_frames=n_frames, frame_size=fs))
    cfg.paths.ensure_dirs()
    scipy.io.savemat(str(cfg.paths.mat_data / cfg.paths.mat_filename),
                     {cfg.signals.mat_info[m]: g[f"in_{m}"].astype(np.complex128) for m in mods})
    seen = []

    def stub(block):
        seen.append(block.shape)
        return orc.features18_batch(block[:, :fs]).astype(np.float32)

    run_extraction(cfg, compute=stub, verbose=False)
    assert len(seen) == 6 and all(s[0] == 2 * n_frames for s in seen)
    for m in mods:
        d = scipy.io.loadmat(str(cfg.paths.calculated_features / f"{m}_features.mat"))
        keys = sorted(k for k in d if not k.startswith("__"))
        assert keys == sorted(["Modulation", cfg.signals.mat_info[m]])
        arr = d[cfg.signals.mat_info[m]]
        assert arr.dtype == np.float32 and arr.shape == (2, n_frames, 18)
        assert str(np.ravel(d["Modulation"])[0]) == m
        # what the reference itself wrote for the same container
        assert np.allclose(arr, g[f"out_{m}"], rtol=2e-6, atol=0, equal_nan=True)
        # downstream indexing patterns (graphics.py:44-46 style) keep working
        assert arr[:, :, [1, 3, 5]].shape == (2, n_frames, 3)


_WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, os.environ["AMCX_REPO"])
    from amcpy_amd.sharding import shard_range, sharded_features
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    F, L, N = int(os.environ["AMCX_F"]), 40, 32
    rng = np.random.default_rng(5)
    frames = (rng.standard_normal((F, L)) + 1j * rng.standard_normal((F, L))).astype(np.complex64)
    calls = []
    def compute(block):                       # stand-in engine: row checksum in 18 columns
        calls.append(block.shape[0])
        base = np.abs(block[:, :N]).sum(axis=1, dtype=np.float64)
        return (base[:, None] * np.arange(1, 19)[None, :]).astype(np.float32)
    out = sharded_features(frames, N, compute, rank, world)
    lo, hi = shard_range(F, rank, world)
    assert sum(calls) == hi - lo, (calls, lo, hi)
    # the all-gather form (a device-side consumer on every rank): the same matrix everywhere
    import torch
    from amcpy_amd.sharding import all_gather_rows
    mine = torch.from_numpy(compute(frames[lo:hi])) if hi > lo else torch.empty((0, 18))
    everywhere = all_gather_rows(mine, F, rank, world)
    want_all = (np.abs(frames[:, :N]).sum(axis=1, dtype=np.float64)[:, None] * np.arange(1, 19)).astype(np.float32)
    assert tuple(everywhere.shape) == (F, 18) and np.array_equal(everywhere.numpy(), want_all), rank
    if rank == 0:
        want = (np.abs(frames[:, :N]).sum(axis=1, dtype=np.float64)[:, None] * np.arange(1, 19)).astype(np.float32)
        assert out.shape == (F, 18) and np.array_equal(out, want)
        print("GATHER_OK", F, world)
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()
''')


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("F", [13, 1, 64])
def test_two_rank_sharding_over_gloo(tmp_path, F):
    """N>1 path on CPU: two processes, gloo, contiguous frame shards, rank-0 gather."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), AMCX_REPO=str(REPO), AMCX_F=str(F),
                   PYTHONDONTWRITEBYTECODE="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert f"GATHER_OK {F} 2" in outs[0]


def test_raw_stream_framing(tmp_path):
    """extract_raw_stream: headerless complex64 file -> consecutive frames after the skipped
    samples, trailing partial frame dropped (reference old/read_binary_stream.py:28,54-56);
    the engine is stubbed, this is the host-side framing."""
    from amcpy_amd.feature_extraction import extract_raw_stream
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(1000) + 1j * rng.standard_normal(1000)).astype(np.complex64)
    path = tmp_path / "binary_awgn_BPSK(10)"
    x.tofile(path)
    seen = []

    def compute(block):
        seen.append(np.array(block))
        return np.zeros((block.shape[0], 18), dtype=np.float32)

    out = extract_raw_stream(path, 64, skip_samples=100, compute=compute)
    assert out.shape == (14, 18) and out.dtype == np.float32          # (1000 - 100) // 64
    assert np.array_equal(seen[0], x[100:100 + 14 * 64].reshape(14, 64))
    assert extract_raw_stream(path, 64, skip_samples=100, max_frames=3, compute=compute).shape == (3, 18)
    assert extract_raw_stream(path, 2048, compute=compute).shape == (0, 18)
    with pytest.raises(ValueError):
        extract_raw_stream(path, 1, compute=compute)


def test_synthetic_generator_statistics():
    from amcpy_amd import synth
    for mod in synth.MODS6[:5]:
        pts = synth.constellation(mod)
        assert np.isclose((np.abs(pts) ** 2).mean(), 1.0)
    x = synth.host_block("QPSK", 10.0, 64, 2048, seed=1)
    assert x.dtype == np.complex64 and x.shape == (64, 2048)
    p = (np.abs(x) ** 2).mean()
    assert abs(p - 1.1) < 0.02                      # signal power 1 + noise power 0.1
    w = synth.host_block("WGN", 0.0, 64, 2048, seed=2)
    assert abs((np.abs(w) ** 2).mean() - 1.0) < 0.02
    assert list(synth.snr_grid(26)[[0, -1]]) == [-20.0, 30.0] and list(synth.snr_grid(2)) == [0.0, 10.0]


def test_config_mirror_equals_reference_defaults():
    """Every field name and default of the reference's config layer, captured from
    the imported reference (oracle/capture_golden.py -> config_defaults.json)."""
    import dataclasses
    import json
    from amcpy_amd.config import Config, FeatureConfig, Paths
    ref = json.loads((REPO / "tests" / "golden" / "config_defaults.json").read_text())
    cfg = Config(paths=Paths(root=Path("/project")))

    def norm(v):
        if isinstance(v, Path):
            return str(v)
        if isinstance(v, tuple):
            return [norm(x) for x in v]
        if isinstance(v, dict):
            return {str(k): norm(x) for k, x in v.items()}
        return v

    for grp in ("paths", "signals", "features", "training"):
        obj = getattr(cfg, grp)
        mine = {f.name: norm(getattr(obj, f.name)) for f in dataclasses.fields(obj)}
        want = {k: norm(v) for k, v in ref[grp].items()
                if k not in ("names", "used_names", "num_used", "feature_files")}
        assert mine == want, (grp, mine, want)
    assert {str(k): v for k, v in FeatureConfig.names.items()} == ref["features"]["names"]
    assert cfg.features.used_names == ref["features"]["used_names"]
    assert cfg.features.num_used == ref["features"]["num_used"]
    assert cfg.training.feature_files == ref["training"]["feature_files"]


def test_frame_rows_gathers_fortran_ordered_containers():
    """loadmat returns Fortran-ordered arrays; FrameRows must deliver the snr-major flattening of
    parsed[:n_snr, :n_frames, :N] chunk by chunk without a full reshape copy."""
    from amcpy_amd.feature_extraction import FrameRows
    rng = np.random.default_rng(3)
    full = (rng.standard_normal((3, 300, 40)) + 1j * rng.standard_normal((3, 300, 40)))
    parsed = np.asfortranarray(full)                      # (n_snr+1, n_frames+..., L) as loadmat gives it
    n_snr, n_frames, N = 2, 260, 32
    want = full[:n_snr, :n_frames].reshape(n_snr * n_frames, 40)
    rows = FrameRows(parsed, n_snr, n_frames)
    assert rows.shape == (n_snr * n_frames, 40) and np.array_equal(rows.to_array(), want)
    for (g0, g1) in [(0, 1), (255, 265), (100, 520), (0, 520)]:      # chunks that straddle an snr row
        dst = np.empty((g1 - g0, N), dtype=np.complex128)
        rows.gather(dst, g0, g1, N)
        assert np.array_equal(dst, want[g0:g1, :N])
        # the rectangles the native engine is called on tile the range in order
        tiles = list(rows.slice(g0, g1).blocks())
        flat = [s * n_frames + k for s0, s1, k0, k1 in tiles for s in range(s0, s1) for k in range(k0, k1)]
        assert flat == list(range(g0, g1)) and len(tiles) <= 3
    part = rows.slice(250, 400)                           # a rank's contiguous range
    dst = np.empty((150, N), dtype=np.complex64)          # narrowing cast on the way is allowed
    part.gather(dst, 0, 150, N)
    assert np.array_equal(dst, want[250:400, :N].astype(np.complex64))


_EXTRACT_WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, os.environ["AMCX_REPO"])
    from pathlib import Path
    from amcpy_amd.config import Config, Paths, SignalConfig
    from amcpy_amd import feature_extraction as fe
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    loads = []
    real_load = fe._load_variable
    def counting_load(path, key, *more):
        loads.append(key)
        return real_load(path, key, *more)
    fe._load_variable = counting_load
    seen = []
    def compute(block):                       # stand-in engine: per-row checksum
        seen.append(block.shape)
        base = np.abs(np.asarray(block)[:, :16]).sum(axis=1, dtype=np.float64)
        return (base[:, None] * np.arange(1, 19)[None, :]).astype(np.float32)
    cfg = Config(paths=Paths(root=Path(os.environ["AMCX_ROOT"])),
                 signals=SignalConfig(snr_values={0: "0", 1: "10", 2: "20"}, num_frames=7, frame_size=16))
    fe.run_extraction(cfg, compute=compute, verbose=False)
    # a compressed container: rank 0 alone decodes it (and publishes); an uncompressed one is mapped by every
    # rank for itself; either way every rank computes only its share -- 3 snr rows x 7 frames over two ranks are cut
    # along the frame axis (sharding.shard_by_frames): frames 0-3 and 4-6 of every snr row
    direct = os.environ["AMCX_MODE"] == "direct"
    assert len(loads) == (6 if rank == 0 or direct else 0), (rank, loads)
    assert all(s[1] == (20 if direct else 16) for s in seen) and sum(s[0] for s in seen) == 6 * (12, 9)[rank], seen
    print("EXTRACT_OK", rank, len(loads), sum(s[0] for s in seen))
    dist.destroy_process_group()
''')


@pytest.mark.parametrize("mode", ["publish", "direct"])
def test_two_rank_run_extraction_rank0_decodes_and_publishes(tmp_path, mode):
    """run_extraction with two ranks over gloo.  "publish": a compressed container -- rank 0 decodes each
    variable once and publishes it through shared memory in the memory order it has.  "direct": an
    uncompressed one -- every rank maps the variable from the file itself, nothing is published.  Both ranks
    compute their shard, rank 0 writes files equal to the single-process result; no shared file is left."""
    import glob
    import scipy.io
    from amcpy_amd.config import Config, Paths, SignalConfig
    from amcpy_amd import feature_extraction as fe
    rng = np.random.default_rng(11)
    cfg = Config(paths=Paths(root=tmp_path / "two"),
                 signals=SignalConfig(snr_values={0: "0", 1: "10", 2: "20"}, num_frames=7, frame_size=16))
    cfg1 = Config(paths=Paths(root=tmp_path / "one"), signals=cfg.signals)
    container = {cfg.signals.mat_info[m]: (rng.standard_normal((3, 9, 20)) + 1j * rng.standard_normal((3, 9, 20)))
                 for m in cfg.signals.modulations_with_noise}
    for c in (cfg, cfg1):
        c.paths.ensure_dirs()
        scipy.io.savemat(str(c.paths.mat_data / c.paths.mat_filename), container, do_compression=(mode == "publish"))

    def compute(block):
        base = np.abs(np.asarray(block)[:, :16]).sum(axis=1, dtype=np.float64)
        return (base[:, None] * np.arange(1, 19)[None, :]).astype(np.float32)

    fe.run_extraction(cfg1, compute=compute, verbose=False)
    before = set(glob.glob(str(fe._shared_dir(0) / "amcx_frames_*")))
    script = tmp_path / "extract_worker.py"
    script.write_text(_EXTRACT_WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   AMCX_REPO=str(REPO), AMCX_ROOT=str(tmp_path / "two"), AMCX_MODE=mode, PYTHONDONTWRITEBYTECODE="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "EXTRACT_OK 0 6 72" in outs[0] and f"EXTRACT_OK 1 {6 if mode == 'direct' else 0} 54" in outs[1], outs
    assert set(glob.glob(str(fe._shared_dir(0) / "amcx_frames_*"))) == before, "shared frame files left behind"
    for m in cfg.signals.modulations_with_noise:
        a = scipy.io.loadmat(str(cfg1.paths.calculated_features / f"{m}_features.mat"))
        b = scipy.io.loadmat(str(cfg.paths.calculated_features / f"{m}_features.mat"))
        key = cfg.signals.mat_info[m]
        assert b[key].shape == (3, 7, 18) and np.array_equal(a[key], b[key])


def _build_abi_check(tmp_path):
    exe = tmp_path / "abi_check"
    lib_dir = REPO / "amcpy_amd" / "lib"
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", f"-I{REPO / 'include'}", str(REPO / "tests" / "c_abi" / "abi_check.c"),
           f"-L{lib_dir}", "-lamcx", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_plain_c_client_of_the_abi(tmp_path):
    """include/amcx.h is a C header: a C99 client compiles against it with -Wall -Werror, links
    libamcx.so and gets the documented error codes -- and, without a GPU, ENODEV from the
    host-buffer entries instead of a CPU computation."""
    exe = _build_abi_check(tmp_path)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "abi_check ok" in r.stdout, r.stdout + r.stderr


def test_iq_pair_dataset_framing(tmp_path):
    """extract_iq_pairs: an (F, L, 2) float32 dataset (RadioML layout, reference old/dataset.py:50-56) is
    re-viewed as complex64 frames chunk by chunk; works on anything sliceable (here: a memmap and a
    minimal h5py.Dataset look-alike); the engine is stubbed."""
    from amcpy_amd.feature_extraction import extract_iq_pairs, extract_radioml_hdf5, _pairs_as_complex
    rng = np.random.default_rng(2)
    pairs = rng.standard_normal((50, 64, 2)).astype(np.float32)
    want = pairs[..., 0] + 1j * pairs[..., 1]
    path = tmp_path / "pairs.f32"
    pairs.tofile(path)
    mm = np.memmap(path, dtype=np.float32, mode="r", shape=(50, 64, 2))

    class FakeDataset:                       # what h5py hands out: shape, dtype, __getitem__ returning ndarrays
        shape, dtype = pairs.shape, pairs.dtype
        reads = []

        def __getitem__(self, idx):
            self.reads.append(idx)
            return pairs[idx]

    seen = []

    def compute(block):
        seen.append(np.array(block))
        return np.zeros((block.shape[0], 18), dtype=np.float32)

    for ds in (pairs, mm, FakeDataset()):
        seen.clear()
        out = extract_iq_pairs(ds, 32, first_frame=5, max_frames=20, compute=compute)
        assert out.shape == (20, 18) and out.dtype == np.float32
        assert seen[0].dtype == np.complex64 and np.array_equal(seen[0], want[5:25, :32])
    view = _pairs_as_complex(pairs[13:20])
    assert view.base is not None and np.array_equal(view, want[13:20])          # a view, not a copy
    assert np.array_equal(_pairs_as_complex(np.asfortranarray(pairs[13:20])), want[13:20])
    assert extract_iq_pairs(pairs, compute=compute, first_frame=60).shape == (0, 18)
    with pytest.raises(ValueError):
        extract_iq_pairs(pairs[..., :1], compute=compute)
    with pytest.raises(TypeError):
        extract_iq_pairs(pairs.astype(np.float64), compute=compute)
    try:
        import h5py  # noqa: F401
    except ImportError:
        from amcpy_amd import hdf5_min
        # neither h5py nor an HDF5 C library: ImportError that names both; with the C library: the file is what is missing
        with pytest.raises(FileNotFoundError if hdf5_min.available() else ImportError):
            extract_radioml_hdf5(tmp_path / "missing.hdf5")


def _hdf5_or_skip():
    from amcpy_amd import hdf5_min
    if not hdf5_min.available():
        pytest.skip("no HDF5 C library (>= 1.10) on this machine: set AMCX_LIBHDF5")
    return hdf5_min


def test_hdf5_min_reads_the_container_h5py_wrote():
    """tests/golden/radioml_like.h5 was written by the REAL h5py (tests/golden/make_radioml_like_h5.py, run with the image's
    conda interpreter) in the layout of RadioML 2018.01A -- X float32 (F, 1024, 2) chunked + shuffled + gzip, Y int64 one-hot
    contiguous, Z int64 chunked; the reference reads such a file in old/dataset.py:43-56.  amcpy_amd.hdf5_min (ctypes over
    libhdf5) returns exactly the bytes h5py was given (SHA-256 recorded by the writing script), whole and in row ranges, and
    extract_radioml_hdf5 hands the engine the complex64 frames those pairs are."""
    import hashlib
    h5 = _hdf5_or_skip()
    from amcpy_amd.feature_extraction import extract_radioml_hdf5
    meta = json.loads((REPO / "tests" / "golden" / "radioml_like.json").read_text())
    path = REPO / "tests" / "golden" / "radioml_like.h5"
    with h5.File(path) as fh:
        assert "X" in fh and "nope" not in fh
        with pytest.raises(KeyError):
            fh["nope"]
        X = fh["X"]
        assert X.shape == (24, 1024, 2) and X.dtype == np.float32 and len(X) == 24 and X.ndim == 3
        # chunked + shuffled + gzip: read as raw chunks, inflated and unshuffled on several threads outside the library
        assert X.chunks == (8, 1024, 2) and X.filters == (h5.H5Z_FILTER_SHUFFLE, h5.H5Z_FILTER_DEFLATE) and X._parallel_chunks
        assert X.file_offset is None and fh["Y"].chunks is None and not fh["Z"]._parallel_chunks
        assert fh["Y"].dtype == np.int64 and fh["Z"].shape == (24, 1)
        whole = {k: fh[k][:] for k in "XYZ"}
        for k, arr in whole.items():
            assert hashlib.sha256(arr.tobytes()).hexdigest() == meta["sha256"][k], k
            assert list(arr.shape) == meta["shape"][k]
        assert np.array_equal(X[5:19], whole["X"][5:19]) and np.array_equal(X[7], whole["X"][7])
        assert np.array_equal(X[-3:], whole["X"][-3:]) and X[24:].shape == (0, 1024, 2) and X[...].shape == (24, 1024, 2)
        with pytest.raises(ValueError):
            X[::2]
        with pytest.raises(IndexError):
            X[24]
        assert np.array_equal(X[-1], whole["X"][23]) and np.array_equal(X[-24], whole["X"][0])
        assert (whole["Y"].sum(axis=1) == 1).all() and set(whole["Z"][:, 0]) == {2, 18}
    with pytest.raises(ValueError):
        X[0:1]                                   # the file is closed
    want = whole["X"][..., 0] + 1j * whole["X"][..., 1]
    seen = []

    def compute(block):
        seen.append(np.array(block))
        return np.zeros((block.shape[0], 18), dtype=np.float32)

    out = extract_radioml_hdf5(path, first_frame=3, max_frames=17, frame_size=512, compute=compute)
    assert out.shape == (17, 18) and seen[0].dtype == np.complex64 and np.array_equal(seen[0], want[3:20, :512])
    with pytest.raises(KeyError):
        extract_radioml_hdf5(path, key="W", compute=compute)


def _mat73_variable(mi: int, mod: str):
    """The arrays tests/golden/make_mat73_like.py wrote, rebuilt from their seeds (numpy only)."""
    from amcpy_amd import synth
    rows = [synth.host_block(mod, snr, 5, 300, seed=7300 + 10 * mi + si).astype(np.complex128) * (1.0 + 1e-9 * (mi + 1))
            for si, snr in enumerate((0.0, 10.0))]
    return np.stack(rows)


def test_mat73_container_reads_bit_for_bit(tmp_path):
    """MATLAB -v7.3 containers (HDF5 behind a 512-byte header; the only form MATLAB has for a variable above 2 GB -- one
    BASELINE configs[1] modulation is 3.49 GB -- and one the reference's loadmat refuses, feature_extraction.py:46-47).
    tests/golden/mat73_like.mat was written by the real h5py in MATLAB's layout: reversed dimensions, {real, imag}
    compounds, MATLAB_class attributes; six double-complex variables stored chunked + deflate / chunked / contiguous, a
    single-complex one, a real one, a char one.  Every numeric variable must come back with the bytes it was written
    from (SHA-256 in mat73_like.json), through the decoded path and -- contiguous ones -- as offsets into the file; a
    char variable and a missing one give the reader's own errors; a chunk nobody wrote reads as the fill value."""
    import hashlib
    import json
    from amcpy_amd import hdf5_min, matfile
    from amcpy_amd.feature_extraction import FileComplex, FrameRows
    if not hdf5_min.available():
        pytest.skip("no HDF5 C library on this machine")
    path = REPO / "tests" / "golden" / "mat73_like.mat"
    meta = json.loads((REPO / "tests" / "golden" / "mat73_like.json").read_text())
    assert matfile.is_v73(path) and not matfile.is_v73(REPO / "tests" / "golden" / "kat_n10.json")
    sha = meta["sha256_column_major"]

    def digest(a):
        return hashlib.sha256(np.asfortranarray(a).tobytes(order="F")).hexdigest()

    names = ["signal_bpsk", "signal_qpsk", "signal_8psk", "signal_qam16", "signal_qam64", "signal_noise"]
    for mi, (name, mod) in enumerate(zip(names, ["BPSK", "QPSK", "8PSK", "16QAM", "64QAM", "WGN"])):
        want = _mat73_variable(mi, mod)
        assert digest(want) == sha[name]                             # the fixture is what its script says it is
        got = matfile.load_variable(path, name)
        assert isinstance(got, np.ndarray) and got.dtype == np.complex128 and got.shape == (2, 5, 300) and got.flags.f_contiguous
        assert digest(got) == sha[name], name
        direct = matfile.load_variable(path, name, None, True)
        if meta["layout"][name] == "contiguous":
            assert isinstance(direct, FileComplex) and direct.interleaved and direct.order == "F"
            assert direct.strides_elems == (1, 2, 10) and direct.shape == (2, 5, 300)
            assert digest(np.asarray(direct[:])) == sha[name]
            rows = FrameRows(direct, 2, 5)                           # frames as the engine's stand-ins see them
            assert np.array_equal(rows.to_array(), want.reshape(10, 300))
            direct.release()
        else:
            assert isinstance(direct, np.ndarray) and digest(direct) == sha[name]
    single = matfile.load_variable(path, "signal_single", None, True)
    assert isinstance(single, FileComplex) and single.dtype == np.complex64 and digest(np.asarray(single[:])) == sha["signal_single"]
    assert digest(matfile.load_variable(path, "signal_single")) == sha["signal_single"]
    real = matfile.load_variable(path, "signal_real")
    assert real.dtype == np.float64 and digest(real) == sha["signal_real"]
    real_d = matfile.load_variable(path, "signal_real", None, True)
    assert isinstance(real_d, FileComplex) and not real_d.interleaved and real_d.imag_offset is None
    assert np.array_equal(np.asarray(real_d[:]).real, real) and not np.asarray(real_d[:]).imag.any()
    with pytest.raises(KeyError):
        matfile.load_variable(path, "signal_absent")
    with pytest.raises(NotImplementedError, match="v7.3"):          # scipy's own words for what this reader leaves alone
        matfile.load_variable(path, "note")
    with hdf5_min.File(path) as fh:
        part = fh["partly_written"]
        assert part._parallel_chunks and part.chunks == (8, 6)
        for sl in (slice(None), slice(3, 30), slice(16, 24)):
            got = part[sl]
            assert got.shape[0] == len(range(*sl.indices(40)))
        assert hashlib.sha256(part[:].tobytes()).hexdigest() == sha["partly_written"]
        assert (part[0:8] == 2.5).all() and (part[16:32] == 2.5).all() and not (part[8:16] == 2.5).all()
        assert fh["signal_bpsk"].attr_string("MATLAB_class") == "double" and fh["signal_bpsk"].attr_string("absent") is None
        assert fh["signal_bpsk"][37:222].shape == (185, 5, 2) and np.array_equal(fh["signal_bpsk"][37:222], fh["signal_bpsk"][:][37:222])


def test_hdf5_min_writes_the_mat73_layout_it_reads(tmp_path):
    """hdf5_min's own writer of MATLAB's -v7.3 layout (user block with the MATLAB header, reversed dimensions, {real, imag}
    compounds, MATLAB_class) -- what tools/mat73_ingest_probe.py writes a configs[1]-sized container with -- read back
    through matfile.load_variable both ways, contiguous and chunked + deflate, double and single, complex and real; and
    refused by scipy.io.loadmat the way a real -v7.3 file is."""
    import scipy.io
    from amcpy_amd import hdf5_min, matfile
    from amcpy_amd.feature_extraction import FileComplex
    if not hdf5_min.available():
        pytest.skip("no HDF5 C library on this machine")
    rng = np.random.default_rng(1)
    x = rng.standard_normal((3, 7, 40)) + 1j * rng.standard_normal((3, 7, 40))
    path = tmp_path / "own.mat"
    with hdf5_min.File(path, "w", userblock=512) as fh:
        fh.write_mat73_variable("a", x)
        fh.write_mat73_variable("b", x.astype(np.complex64), chunks=(10, 7, 3), deflate=3)
        fh.write_mat73_variable("r", x.real.copy())
    hdf5_min.write_matlab_header(path)
    assert matfile.is_v73(path) and path.read_bytes()[:19] == b"MATLAB 7.3 MAT-file" and path.read_bytes()[124:128] == b"\x00\x02IM"
    for name, ref in (("a", x), ("b", x.astype(np.complex64)), ("r", x.real)):
        got = matfile.load_variable(path, name)
        assert got.dtype == ref.dtype and got.flags.f_contiguous and np.array_equal(got, ref), name
        direct = matfile.load_variable(path, name, None, True)
        if name == "b":
            assert isinstance(direct, np.ndarray) and np.array_equal(direct, ref)
        else:
            assert isinstance(direct, FileComplex)
            back = np.asarray(direct[:])
            assert np.array_equal(back.real if name == "r" else back, ref)
            direct.release()
    with hdf5_min.File(path) as fh:
        assert fh["a"].attr_string("MATLAB_class") == "double" and fh["b"].attr_string("MATLAB_class") == "single"
        assert fh["a"].shape == (40, 7, 3) and fh["a"].complex_pair and fh["a"].file_offset >= 512
    with pytest.raises(NotImplementedError):
        scipy.io.loadmat(str(path))


def test_run_extraction_takes_a_mat73_container(tmp_path):
    """run_extraction on the -v7.3 container (stand-in engine, no GPU here) writes the same six feature files as on a
    level-5 container of the same arrays -- contiguous variables as offsets into the file, compressed ones decoded."""
    import shutil
    import scipy.io
    from amcpy_amd import hdf5_min
    from amcpy_amd.config import Config, Paths, SignalConfig
    from amcpy_amd.feature_extraction import run_extraction
    if not hdf5_min.available():
        pytest.skip("no HDF5 C library on this machine")
    outs = {}
    for kind in ("v73", "v5"):
        cfg = Config(paths=Paths(root=tmp_path / kind),
                     signals=SignalConfig(snr_values={0: "0", 1: "10"}, num_frames=5, frame_size=256))
        cfg.paths.ensure_dirs()
        target = cfg.paths.mat_data / cfg.paths.mat_filename
        if kind == "v73":
            shutil.copy(REPO / "tests" / "golden" / "mat73_like.mat", target)
        else:
            scipy.io.savemat(str(target), {cfg.signals.mat_info[m]: _mat73_variable(i, m)
                                           for i, m in enumerate(cfg.signals.modulations_with_noise)})
        run_extraction(cfg, compute=lambda x: _marker_features(x[:, :256]), verbose=False)
        outs[kind] = {m: scipy.io.loadmat(str(cfg.paths.calculated_features / f"{m}_features.mat"))[cfg.signals.mat_info[m]]
                      for m in cfg.signals.modulations_with_noise}
    for m, a in outs["v73"].items():
        assert a.shape == (2, 5, 18) and a.dtype == np.float32 and np.array_equal(a, outs["v5"][m]), m


def test_hdf5_min_round_trip_and_concurrent_readers(tmp_path):
    """Files hdf5_min writes itself -- contiguous, chunked, chunked + deflate -- read back bit for bit, also from several
    threads at once (libhdf5 builds are usually not thread-safe: every call is made under one lock; extract_iq_pairs slices
    a dataset from reader threads); something that is not HDF5 is refused."""
    import threading
    h5 = _hdf5_or_skip()
    rng = np.random.default_rng(11)
    data = rng.standard_normal((130, 64, 2)).astype(np.float32)
    labels = rng.integers(-20, 31, size=(130, 1)).astype(np.int64)
    for kw in ({}, {"chunks": (16, 64, 2)}, {"chunks": (7, 64, 2), "deflate": 4}):
        path = tmp_path / "t.h5"
        with h5.File(path, "w") as fh:
            fh.create_dataset("X", data, **kw)
            fh.create_dataset("Z", labels)
        with h5.File(path) as fh:
            ds = fh["X"]
            assert np.array_equal(ds[:], data) and np.array_equal(fh["Z"][:], labels)
            # a contiguous dataset says where its bytes lie in the file (the engine's staging threads pread them
            # there, past the library); a chunked one has no such place
            assert ds.little_endian and (ds.file_offset is None) == ("chunks" in kw)
            if ds.file_offset is not None:
                raw = np.fromfile(path, dtype=np.float32, count=data.size, offset=ds.file_offset).reshape(data.shape)
                assert np.array_equal(raw, data)
            ok = {}

            def read(a, ds=ds, ok=ok):
                ok[a] = np.array_equal(ds[a:a + 40], data[a:a + 40])

            threads = [threading.Thread(target=read, args=(a,)) for a in range(0, 120, 5)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            assert len(ok) == 24 and all(ok.values())
    bad = tmp_path / "bad.h5"
    bad.write_bytes(b"not an hdf5 file " * 64)
    with pytest.raises(OSError):
        h5.File(bad)
    with pytest.raises(FileNotFoundError):
        h5.File(tmp_path / "absent.h5")


# ----------------------------------------------------------------------------
# round 3: the memory-mapped .mat reader, the native staging threads, collective-safe failures
# ----------------------------------------------------------------------------
def test_mat_reader_returns_views_equal_to_loadmat(tmp_path):
    """amcpy_amd/matfile.py on level-5 files as scipy.io.savemat writes them (the reference's input,
    feature_extraction.py:46-48): complex128 / complex64 / real variables, compressed or not, come
    back as Fortran-ordered views of the mapping (no decode) equal to what loadmat returns; what the
    fast reader does not take on falls through to scipy; a missing variable is a KeyError."""
    import scipy.io
    from amcpy_amd.feature_extraction import SplitComplex
    from amcpy_amd.matfile import _Unsupported, load_variable, read_variable_v5
    rng = np.random.default_rng(21)
    c128 = rng.standard_normal((3, 5, 40)) + 1j * rng.standard_normal((3, 5, 40))
    box = {"signal_bpsk": c128, "signal_qpsk": c128.astype(np.complex64) * 2, "just_real": c128.real.copy(),
           "ints": np.arange(24, dtype=np.int16).reshape(2, 3, 4), "text": "hello"}
    for compress in (False, True):
        path = tmp_path / f"box_{int(compress)}.mat"
        scipy.io.savemat(str(path), box, do_compression=compress)
        ref = scipy.io.loadmat(str(path))
        for key in ("signal_bpsk", "signal_qpsk"):
            got = load_variable(path, key)
            assert isinstance(got, SplitComplex) and got.shape == (3, 5, 40) and got.dtype == ref[key].dtype
            assert got.real.flags.f_contiguous and not got.real.flags.writeable and not got.real.flags.owndata
            assert np.array_equal(got[:, :, :], ref[key]) and np.array_equal(got[1, 2:4, :7], ref[key][1, 2:4, :7])
        real = load_variable(path, "just_real")
        assert isinstance(real, SplitComplex) and real.imag is None and real.real.dtype == np.float64
        assert np.array_equal(real.real, ref["just_real"]) and np.array_equal(real[:2], ref["just_real"][:2])
        assert real.source == got.source == ("inflated" if compress else "mapped")
        # with a buffer pool an uncompressed variable is READ (preadv on threads) into arrays that are reused
        from amcpy_amd.matfile import BufferPool
        pool = BufferPool()
        first = load_variable(path, "signal_bpsk", pool)
        assert first.source == ("inflated" if compress else "read") and np.array_equal(first[:, :, :], ref["signal_bpsk"])
        if not compress:
            assert first.real.flags.f_contiguous and first.real.flags.writeable
            addr = first.real.__array_interface__["data"][0]
            first.release()
            again = load_variable(path, "signal_bpsk", pool)    # same shape: the same buffers come back
            assert again.real.__array_interface__["data"][0] in (addr, first.imag.__array_interface__["data"][0])
            assert np.array_equal(again[:, :, :], ref["signal_bpsk"])
            again.release()
        with pytest.raises(_Unsupported):
            read_variable_v5(path, "ints")                       # int16 storage: not the fast reader's business
        assert np.array_equal(load_variable(path, "ints"), ref["ints"])           # ... scipy's
        with pytest.raises(KeyError):
            load_variable(path, "signal_nope")
    # an element the fast reader cannot even name (an empty array: its inflated header is under 64 bytes) must not turn
    # "I do not know" into "no such variable": scipy reads it
    for compress in (False, True):
        path = tmp_path / f"tiny_{int(compress)}.mat"
        scipy.io.savemat(str(path), {"tiny": np.zeros((0, 0)), "signal_bpsk": c128}, do_compression=compress)
        assert load_variable(path, "tiny").shape == (0, 0)
        assert np.array_equal(load_variable(path, "signal_bpsk")[:, :, :], c128)
        with pytest.raises(KeyError):
            load_variable(path, "signal_nope")
    junk = tmp_path / "junk.mat"
    junk.write_bytes(b"not a mat file at all" * 20)
    with pytest.raises(Exception):
        load_variable(junk, "x")
    with pytest.raises(FileNotFoundError):
        load_variable(tmp_path / "absent.mat", "x")


def _stage(arr_re, arr_im, kind, S, K, N, strides, first, count, threads=3):
    import ctypes as C
    from amcpy_amd import _lib
    lib = _lib.load()
    F = S * K
    pm, inner = C.c_int32(-1), C.c_int32(-1)
    probe = lib.amcx_stage_host(arr_re.ctypes.data, None if arr_im is None else arr_im.ctypes.data, kind, S, K, N,
                                *strides, 0, 0, None, 0, threads, C.byref(pm), C.byref(inner))
    _lib.check(probe)
    unit = F if pm.value else N
    dst = np.full((count, unit), np.nan + 1j * np.nan, dtype=np.complex64)
    _lib.check(lib.amcx_stage_host(arr_re.ctypes.data, None if arr_im is None else arr_im.ctypes.data, kind, S, K, N,
                                   *strides, first, count, dst.ctypes.data, dst.nbytes, threads, C.byref(pm), C.byref(inner)))
    return dst, pm.value, inner.value


def test_native_staging_of_strided_containers():
    """amcx_stage_host (the host half of amcx_ctx_features18_strided_host; needs no GPU): sample planes
    of a Fortran-ordered container with more snr rows / frames / samples than the configuration uses,
    rows of a C-ordered one, split real / imaginary arrays, float32 and float64 -- every staged chunk
    equals numpy's own slicing + astype(complex64) (round to nearest even), whatever the thread count."""
    from amcpy_amd import _lib
    rng = np.random.default_rng(5)
    S_tot, K_tot, L = 4, 37, 300
    S, K, N = 3, 33, 256
    full = rng.standard_normal((S_tot, K_tot, L)) * 1e3 + 1j * rng.standard_normal((S_tot, K_tot, L))
    want = full[:S, :K, :N].astype(np.complex64)                     # numpy's rounding of the doubles
    es = lambda a: [st // a.itemsize for st in a.strides]
    # Fortran order (what loadmat returns): planes, snr the inner axis -> position j = k * S + s
    f = np.asfortranarray(full)
    for first, count in [(0, 1), (5, 17), (250, 6), (0, N)]:
        for threads in (1, 3, 8):
            got, pm, inner = _stage(f, None, _lib.SRC_C128, S, K, N, es(f), first, count, threads)
            assert (pm, inner) == (1, 1)
            ref = want[:, :, first:first + count].transpose(2, 1, 0).reshape(count, K * S)
            assert np.array_equal(got, ref)
    # the same container as complex64, and as two real arrays (a MATLAB v5 file's storage)
    f64 = np.asfortranarray(full.astype(np.complex64))
    got, pm, inner = _stage(f64, None, _lib.SRC_C64, S, K, N, es(f64), 3, 40)
    assert np.array_equal(got, f64[:S, :K, 3:43].transpose(2, 1, 0).reshape(40, K * S))
    re, im = np.asfortranarray(full.real), np.asfortranarray(full.imag)
    got, pm, inner = _stage(re, im, _lib.SRC_F64_SPLIT, S, K, N, es(re), 100, 50)
    assert np.array_equal(got, want[:, :, 100:150].transpose(2, 1, 0).reshape(50, K * S))
    got, _, _ = _stage(re, None, _lib.SRC_F64_SPLIT, S, K, N, es(re), 100, 50)          # a real signal
    assert np.array_equal(got, want.real[:, :, 100:150].transpose(2, 1, 0).reshape(50, K * S).astype(np.complex64))
    re32, im32 = re.astype(np.float32), im.astype(np.float32)
    got, _, _ = _stage(re32, im32, _lib.SRC_F32_SPLIT, S, K, N, es(re32), 0, 9)
    assert np.array_equal(got, (re32 + 1j * im32)[:S, :K, :9].transpose(2, 1, 0).reshape(9, K * S))
    # (L, S, K) in C order seen as (S, K, L): the frame axis is the inner one -> position j = s * K + k
    t = np.ascontiguousarray(full.transpose(2, 0, 1)).transpose(1, 2, 0)
    got, pm, inner = _stage(t, None, _lib.SRC_C128, S, K, N, es(t), 7, 20)
    assert (pm, inner) == (1, 0) and np.array_equal(got, want[:, :, 7:27].transpose(2, 0, 1).reshape(20, S * K))
    # C order: rows, unit = frame g = s * K + k
    got, pm, _ = _stage(full, None, _lib.SRC_C128, S, K, N, es(full), 30, 40)
    assert pm == 0 and np.array_equal(got, want.reshape(S * K, N)[30:70])
    # no contiguous axis at all: refused, the Python layer copies
    odd = full[:, :, ::2]
    with pytest.raises(_lib.AmcxError) as ei:
        _stage(odd[::1, ::2], None, _lib.SRC_C128, 2, 10, 128, es(odd[::1, ::2]), 0, 1)
    assert ei.value.code == _lib.ENOTSUP
    with pytest.raises(ValueError):
        _stage(f, None, _lib.SRC_C128, S, K, N, es(f), N - 2, 5)      # past the last plane


def test_publish_container_keeps_memory_order_and_checks_room(tmp_path, monkeypatch):
    """Rank 0 publishes the used part of a modulation in the order it lies in memory (no host
    transposition), and refuses BEFORE mapping when the shared directory has no room (a sparse
    tmpfs file written past capacity is a SIGBUS)."""
    from amcpy_amd import feature_extraction as fe
    rng = np.random.default_rng(8)
    full = np.asfortranarray(rng.standard_normal((3, 9, 20)) + 1j * rng.standard_normal((3, 9, 20)))
    path = fe._publish_container(full, 2, 7, 16, threads=3)
    try:
        back = np.load(path, mmap_mode="r")
        assert back.shape == (2, 7, 16) and back.flags.f_contiguous and np.array_equal(back, full[:2, :7, :16])
    finally:
        path.unlink()
    c = np.ascontiguousarray(full)
    path = fe._publish_container(c, 2, 7, 16, threads=1)
    try:
        back = np.load(path, mmap_mode="r")
        assert back.flags.c_contiguous and np.array_equal(back, c[:2, :7, :16])
    finally:
        path.unlink()
    with pytest.raises(OSError, match="free to publish"):
        fe._shared_dir(1 << 60)


_FAILING_WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, os.environ["AMCX_REPO"])
    from pathlib import Path
    from amcpy_amd.config import Config, Paths, SignalConfig
    from amcpy_amd import feature_extraction as fe
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []
    def compute(block):
        calls.append(block.shape)
        if rank == 1 and len(calls) == 2:         # the second modulation dies on rank 1 only
            raise FloatingPointError("injected engine failure")
        return np.zeros((block.shape[0], 18), dtype=np.float32)
    cfg = Config(paths=Paths(root=Path(os.environ["AMCX_ROOT"])),
                 signals=SignalConfig(snr_values={0: "0", 1: "10", 2: "20"}, num_frames=7, frame_size=16))
    try:
        fe.run_extraction(cfg, compute=compute, verbose=False)
    except RuntimeError as exc:
        print("RAISED", rank, exc)
        dist.destroy_process_group()
        sys.exit(3)
    print("NO ERROR", rank)
''')


def test_two_rank_failure_raises_on_every_rank(tmp_path):
    """An engine that raises on rank 1 only: both ranks report the same error and exit non-zero within
    the timeout (no rank is left inside a collective), the first modulation's file exists, no shared
    file is left behind.  The reference's threads swallow such failures (feature_extraction.py:33-39,74)."""
    import glob
    import scipy.io
    from amcpy_amd.config import Config, Paths, SignalConfig
    from amcpy_amd import feature_extraction as fe
    rng = np.random.default_rng(12)
    cfg = Config(paths=Paths(root=tmp_path),
                 signals=SignalConfig(snr_values={0: "0", 1: "10", 2: "20"}, num_frames=7, frame_size=16))
    cfg.paths.ensure_dirs()
    scipy.io.savemat(str(cfg.paths.mat_data / cfg.paths.mat_filename),
                     {cfg.signals.mat_info[m]: (rng.standard_normal((3, 9, 20)) + 1j * rng.standard_normal((3, 9, 20)))
                      for m in cfg.signals.modulations_with_noise})
    before = set(glob.glob(str(fe._shared_dir(0) / "amcx_frames_*")))
    script = tmp_path / "failing_worker.py"
    script.write_text(_FAILING_WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   AMCX_REPO=str(REPO), AMCX_ROOT=str(tmp_path), PYTHONDONTWRITEBYTECODE="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert [p.returncode for p in procs] == [3, 3], "\n".join(outs)
    for r in range(2):
        assert f"RAISED {r}" in outs[r] and "rank 1: FloatingPointError: injected engine failure" in outs[r], outs[r]
    assert (cfg.paths.calculated_features / "BPSK_features.mat").exists()
    assert not (cfg.paths.calculated_features / "QPSK_features.mat").exists()
    assert set(glob.glob(str(fe._shared_dir(0) / "amcx_frames_*"))) == before, "shared frame files left behind"


def test_native_staging_on_random_layouts():
    """amcx_stage_host on sixty random containers (no GPU): every axis order, padded in every dimension, complex64 /
    complex128 / split / real-only, random unit ranges and thread counts -- each staged chunk equals numpy's slicing
    + astype(complex64) in the order the layout report says (rows; planes with the snr or the frame axis inside)."""
    import itertools
    from amcpy_amd import _lib
    rng = np.random.default_rng(77)
    orders = list(itertools.permutations(range(3)))
    seen = set()
    for case in range(60):
        S, K, N = int(rng.integers(1, 6)), int(rng.integers(1, 30)), int(rng.integers(2, 90))
        pad = [int(rng.integers(0, 3)) for _ in range(3)]
        shape = (S + pad[0], K + pad[1], N + pad[2])
        order = orders[int(rng.integers(0, len(orders)))]
        full = rng.standard_normal(shape) * 100 + 1j * rng.standard_normal(shape)
        lay = lambda a: np.ascontiguousarray(a.transpose(order)).transpose(np.argsort(order))
        arr = lay(full)
        es = [st // arr.itemsize for st in arr.strides]
        kind = int(rng.integers(0, 4))
        if kind == 0:
            a64 = lay(full.astype(np.complex64))
            re, im, k, es, want = a64, None, _lib.SRC_C64, [st // 8 for st in a64.strides], a64
        elif kind == 1:
            re, im, k, want = arr, None, _lib.SRC_C128, arr.astype(np.complex64)
        elif kind == 2:
            re, im, k, want = lay(full.real), lay(full.imag), _lib.SRC_F64_SPLIT, arr.astype(np.complex64)
            es = [st // 8 for st in re.strides]
        else:
            re, im, k, want = lay(full.real), None, _lib.SRC_F64_SPLIT, arr.real.astype(np.complex64)
            es = [st // 8 for st in re.strides]
        unit_axes = [es[2] == 1, es[1] == 1 and K > 1, es[0] == 1 or S == 1]
        if not any(unit_axes):
            with pytest.raises(_lib.AmcxError):
                _stage(re, im, k, S, K, max(N, 2), es, 0, 1)
            continue
        got_probe, pm, inner = _stage(re, im, k, S, K, N, es, 0, 1, threads=1)
        total = N if pm else S * K
        first = int(rng.integers(0, total))
        count = int(rng.integers(1, total - first + 1))
        got, pm, inner = _stage(re, im, k, S, K, N, es, first, count, threads=int(rng.integers(1, 6)))
        used = want[:S, :K, :N]
        if not pm:
            ref = used.reshape(S * K, N)[first:first + count]
        elif inner:
            ref = used[:, :, first:first + count].transpose(2, 1, 0).reshape(count, K * S)
        else:
            ref = used[:, :, first:first + count].transpose(2, 0, 1).reshape(count, S * K)
        assert np.array_equal(got, ref), (case, shape, order, kind, pm, inner, first, count)
        seen.add((pm, inner))
    assert seen == {(0, 0), (1, 0), (1, 1)}, seen


def test_opting_out_of_torch_is_all_or_nothing(tmp_path):
    """Loading the library without torch -- ``_lib.load(skip_torch=True)``, what `python -m amcpy_amd` does for
    itself, or AMCX_SKIP_TORCH=1 in the environment: the library loads without importing torch, the host-container
    path works from there, and the tensor entry point refuses instead of handing torch's device pointers to a second
    HIP runtime.  The opt-out belongs to that first load: ``main([...])`` called in-process neither skips torch nor
    writes to os.environ, so tensors keep working afterwards."""
    code = textwrap.dedent("""
        import os, sys
        from amcpy_amd import _lib
        _lib.load(skip_torch=%r)
        assert "torch" not in sys.modules, "the library pulled torch in"
        from amcpy_amd.feature_extraction import _rank_world
        assert _rank_world() == (0, 1) and "torch" not in sys.modules
        from amcpy_amd.features import features18
        try:
            features18(None)
        except RuntimeError as exc:
            assert "skip_torch" in str(exc)
            print("REFUSED")
    """)
    base = {k: v for k, v in os.environ.items() if k != "AMCX_SKIP_TORCH"}
    base.update(PYTHONPATH=str(REPO), PYTHONDONTWRITEBYTECODE="1")
    for env, flag in ((dict(base, AMCX_SKIP_TORCH="1"), False), (base, True)):
        r = subprocess.run([sys.executable, "-c", code % flag], env=env, capture_output=True, text=True, timeout=300,
                           cwd=str(tmp_path))
        assert r.returncode == 0 and "REFUSED" in r.stdout, r.stdout + r.stderr
    # in-process use of the command line's entry point: no opt-out, nothing written to the environment
    code = textwrap.dedent("""
        import os, sys
        from amcpy_amd.main import main
        try:
            main(["extract", "--root", %r, "--num-frames", "1", "--frame-size", "128", "--snr-values", "0"])
        except Exception as exc:                       # no container there (and no GPU here): the point is what came before
            print("RAISED", type(exc).__name__)
        from amcpy_amd import _lib
        assert "AMCX_SKIP_TORCH" not in os.environ
        assert "torch" in sys.modules and _lib.torch_wanted()
        _lib.require_torch_runtime()
        print("TENSORS-OK")
    """) % str(tmp_path)
    r = subprocess.run([sys.executable, "-c", code], env=base, capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0 and "TENSORS-OK" in r.stdout, r.stdout + r.stderr


def test_eight_rank_run_extraction_over_gloo(tmp_path):
    """The target topology's rank count on CPU: eight ranks over gloo, 3 x 7 = 21 frames per modulation, so that
    ceil(21 / 8) = 3 frames go to ranks 0..6 and rank 7's shard is EMPTY -- every rank still takes part in every
    collective, rank 0 writes files equal to the single-process run."""
    import scipy.io
    from amcpy_amd.config import Config, Paths, SignalConfig
    from amcpy_amd import feature_extraction as fe
    rng = np.random.default_rng(13)
    cfg = Config(paths=Paths(root=tmp_path / "eight"),
                 signals=SignalConfig(snr_values={0: "0", 1: "10", 2: "20"}, num_frames=7, frame_size=16))
    cfg1 = Config(paths=Paths(root=tmp_path / "one"), signals=cfg.signals)
    container = {cfg.signals.mat_info[m]: (rng.standard_normal((3, 9, 20)) + 1j * rng.standard_normal((3, 9, 20)))
                 for m in cfg.signals.modulations_with_noise}
    for c in (cfg, cfg1):
        c.paths.ensure_dirs()
        scipy.io.savemat(str(c.paths.mat_data / c.paths.mat_filename), container)

    def compute(block):
        base = np.abs(np.asarray(block)[:, :16]).sum(axis=1, dtype=np.float64)
        return (base[:, None] * np.arange(1, 19)[None, :]).astype(np.float32)

    fe.run_extraction(cfg1, compute=compute, verbose=False)
    script = tmp_path / "extract_worker8.py"
    script.write_text(_EXTRACT_WORKER.replace(
        'assert all(s[1] == (20 if direct else 16) for s in seen) and sum(s[0] for s in seen) == 6 * (12, 9)[rank], seen',
        'assert sum(s[0] for s in seen) == (0 if rank == 7 else 6 * 3), (rank, seen)'))
    assert "6 * 3" in script.read_text()            # seven frames per snr row over eight ranks: the flattening is cut
    port = _free_port()
    procs = []
    for r in range(8):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="8", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   AMCX_REPO=str(REPO), AMCX_ROOT=str(tmp_path / "eight"), AMCX_MODE="direct", PYTHONDONTWRITEBYTECODE="1",
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "EXTRACT_OK 0 6 18" in outs[0] and "EXTRACT_OK 7 6 0" in outs[7], (outs[0], outs[7])
    for m in cfg.signals.modulations_with_noise:
        a = scipy.io.loadmat(str(cfg1.paths.calculated_features / f"{m}_features.mat"))
        b = scipy.io.loadmat(str(cfg.paths.calculated_features / f"{m}_features.mat"))
        key = cfg.signals.mat_info[m]
        assert b[key].shape == (3, 7, 18) and np.array_equal(a[key], b[key])


def _stage_file(path, re_off, im_off, kind, S, K, N, strides, first, count, threads=3):
    import ctypes as C
    import os
    from amcpy_amd import _lib
    lib = _lib.load()
    pm, inner = C.c_int32(-1), C.c_int32(-1)
    fd = os.open(str(path), os.O_RDONLY)
    try:
        _lib.check(lib.amcx_stage_file(fd, re_off, -1 if im_off is None else im_off, kind, S, K, N, *strides, 0, 0, None, 0,
                                       threads, C.byref(pm), C.byref(inner)))
        unit = S * K if pm.value else N
        dst = np.full((count, unit), np.nan + 1j * np.nan, dtype=np.complex64)
        _lib.check(lib.amcx_stage_file(fd, re_off, -1 if im_off is None else im_off, kind, S, K, N, *strides, first, count,
                                       dst.ctypes.data, dst.nbytes, threads, C.byref(pm), C.byref(inner)))
    finally:
        os.close(fd)
    return dst, pm.value, inner.value


def test_native_staging_from_a_file_equals_staging_from_memory(tmp_path):
    """amcx_stage_file (the host half of amcx_ctx_features18_strided_file; no GPU): the same containers as bytes in a
    file at odd offsets -- column-major split doubles as a level-5 .mat keeps them, split singles, interleaved
    complex64 / complex128 in both orders, real-only, blocks longer than the per-thread scratch -- staged by pread
    are bit-identical to staging the arrays from memory; a file that ends inside the variable is AMCX_EIO (OSError)."""
    from amcpy_amd import _lib
    rng = np.random.default_rng(21)
    es = lambda a: [st // a.itemsize for st in a.strides]
    for case, (S, K, N) in enumerate([(3, 33, 256), (1, 7, 64), (26, 40, 128), (2, 9000, 16)]):   # 18 000-element planes > one scratch block
        full = rng.standard_normal((S, K, N)) * 1e3 + 1j * rng.standard_normal((S, K, N))
        re, im = np.asfortranarray(full.real), np.asfortranarray(full.imag)
        path = tmp_path / f"c{case}.bin"
        pad_a, pad_b = 8 * int(rng.integers(1, 9)), 8 * int(rng.integers(0, 5))
        with open(path, "wb") as fh:
            fh.write(b"\x5a" * pad_a); fh.write(re.tobytes(order="F")); fh.write(b"\xa5" * pad_b); fh.write(im.tobytes(order="F"))
            off32 = fh.tell()
            fh.write(re.astype(np.float32).tobytes(order="F")); fh.write(im.astype(np.float32).tobytes(order="F"))
            off_c = fh.tell()
            fh.write(np.asfortranarray(full).tobytes(order="F"))
            off_rows = fh.tell()
            fh.write(full.astype(np.complex64).tobytes(order="C"))
        d_im = pad_a + re.nbytes + pad_b
        first, count = int(rng.integers(0, N - 1)), int(rng.integers(1, 6))
        count = min(count, N - first)
        for threads in (1, 4):
            want, pm, inner = _stage(re, im, _lib.SRC_F64_SPLIT, S, K, N, es(re), first, count, threads)
            got, pm2, inner2 = _stage_file(path, pad_a, d_im, _lib.SRC_F64_SPLIT, S, K, N, es(re), first, count, threads)
            assert (pm, inner) == (pm2, inner2) and np.array_equal(got, want), (case, threads)
        want, _, _ = _stage(re, None, _lib.SRC_F64_SPLIT, S, K, N, es(re), first, count)
        got, _, _ = _stage_file(path, pad_a, None, _lib.SRC_F64_SPLIT, S, K, N, es(re), first, count)
        assert np.array_equal(got, want)
        re32, im32 = np.asfortranarray(re.astype(np.float32)), np.asfortranarray(im.astype(np.float32))
        want, _, _ = _stage(re32, im32, _lib.SRC_F32_SPLIT, S, K, N, es(re32), first, count)
        got, _, _ = _stage_file(path, off32, off32 + re32.nbytes, _lib.SRC_F32_SPLIT, S, K, N, es(re32), first, count)
        assert np.array_equal(got, want)
        fc = np.asfortranarray(full)
        want, _, _ = _stage(fc, None, _lib.SRC_C128, S, K, N, es(fc), first, count)
        got, _, _ = _stage_file(path, off_c, None, _lib.SRC_C128, S, K, N, es(fc), first, count)
        assert np.array_equal(got, want)
        rows = np.ascontiguousarray(full.astype(np.complex64))
        f0 = int(rng.integers(0, S * K - 1))
        want, pm, _ = _stage(rows, None, _lib.SRC_C64, S, K, N, es(rows), f0, min(5, S * K - f0))
        got, pm2, _ = _stage_file(path, off_rows, None, _lib.SRC_C64, S, K, N, es(rows), f0, min(5, S * K - f0))
        assert pm == pm2 == 0 and np.array_equal(got, want)
        # the file ends inside the imaginary array
        short = tmp_path / f"short{case}.bin"
        short.write_bytes(path.read_bytes()[:d_im + im.nbytes // 2])
        with pytest.raises(OSError):
            _stage_file(short, pad_a, d_im, _lib.SRC_F64_SPLIT, S, K, N, es(re), 0, N)
    with pytest.raises(ValueError):
        _stage_file(path, -8, None, _lib.SRC_C64, 1, 1, 64, [64, 64, 1], 0, 1)


def test_mat_variables_located_in_the_file(tmp_path):
    """load_variable(direct=True): an uncompressed variable comes back as offsets into the file (FileComplex) whose
    indexing equals loadmat's array and whose staging from the file equals staging loadmat's array from memory; a
    compressed or integer variable falls through to the other readers."""
    import scipy.io
    from amcpy_amd import _lib
    from amcpy_amd.feature_extraction import FileComplex, FrameRows, _native_source
    from amcpy_amd.matfile import load_variable
    rng = np.random.default_rng(8)
    shape = (4, 30, 160)
    x = np.asfortranarray(rng.standard_normal(shape) + 1j * rng.standard_normal(shape))
    r = np.asfortranarray(rng.standard_normal(shape).astype(np.float32))
    path = tmp_path / "c.mat"
    scipy.io.savemat(str(path), {"first": np.arange(7.0), "x": x, "r": r, "i8": np.arange(24, dtype=np.int8).reshape(2, 3, 4)})
    want = scipy.io.loadmat(str(path))
    fx = load_variable(path, "x", direct=True)
    assert isinstance(fx, FileComplex) and fx.shape == shape and fx.dtype == np.complex128 and fx.source == "file"
    assert np.array_equal(fx[1:3, 4:20, :128], want["x"][1:3, 4:20, :128])
    src = _native_source(fx)
    assert src[3] == _lib.SRC_F64_SPLIT and src[4] == [1, 4, 120] and src[6] == fx.fileno()
    S, K, N = 3, 28, 128
    got, pm, inner = _stage_file(path, src[1], src[2], src[3], S, K, N, src[4], 5, 40)
    ref, _, _ = _stage(np.asfortranarray(want["x"].real), np.asfortranarray(want["x"].imag), _lib.SRC_F64_SPLIT, S, K, N,
                       src[4], 5, 40)
    assert (pm, inner) == (1, 1) and np.array_equal(got, ref)
    assert np.array_equal(got, want["x"][:S, :K, 5:45].astype(np.complex64).transpose(2, 1, 0).reshape(40, K * S))
    # the host gather of a FrameRows over it (injected engines, tests) goes through a mapping
    assert np.array_equal(FrameRows(fx, S, K, 10, 50).to_array()[:, :N], want["x"][:S, :K, :].reshape(S * K, -1)[10:50, :N])
    fx.release(); fx.release()
    fr = load_variable(path, "r", direct=True)                 # a real float32 variable: no imaginary array
    assert isinstance(fr, FileComplex) and fr.imag_offset is None and fr.dtype == np.complex64
    assert np.array_equal(fr[:, :, :10], want["r"][:, :, :10].astype(np.complex64))
    assert not isinstance(load_variable(path, "i8", direct=True), FileComplex)       # int8 storage: scipy
    assert not isinstance(load_variable(path, "first", direct=True), FileComplex)    # not 3-D
    zpath = tmp_path / "z.mat"
    scipy.io.savemat(str(zpath), {"x": x}, do_compression=True)
    z = load_variable(zpath, "x", direct=True)
    assert not isinstance(z, FileComplex) and z.source == "inflated" and np.array_equal(z[:2, :5, :9], x[:2, :5, :9])
    with pytest.raises(KeyError):
        load_variable(path, "absent", direct=True)


def test_frame_axis_cut_of_a_container():
    """sharding.shard_by_frames / FrameColumns / gather_frame_columns (world 1 semantics here; two ranks in
    test_two_rank_run_extraction_*): the cut is chosen when it balances within 1/8 of the flattening's, a rank's
    share is the frames [k_lo, k_hi) of every snr row in snr-major order, and the shares tile the container."""
    from amcpy_amd.feature_extraction import FrameColumns
    from amcpy_amd.sharding import gather_frame_columns, shard_by_frames, shard_range
    assert shard_by_frames(26, 4096, 8) and shard_by_frames(26, 512, 8) and shard_by_frames(3, 7, 2)
    assert not shard_by_frames(3, 7, 8)            # fewer frames per row than ranks
    assert not shard_by_frames(26, 9, 8)           # 26 * 2 = 52 frames on the busiest rank against 30: the flattening balances better
    assert not shard_by_frames(26, 4096, 1)
    rng = np.random.default_rng(4)
    S, K, L, N = 3, 11, 20, 16
    full = np.asfortranarray(rng.standard_normal((S + 1, K + 2, L)) + 1j * rng.standard_normal((S + 1, K + 2, L)))
    W = 4
    seen = np.zeros((S, K), dtype=int)
    parts = []
    for r in range(W):
        k_lo, k_hi = shard_range(K, r, W)
        share = FrameColumns(full, S, K, k_lo, k_hi)
        assert share.shape == (S * (k_hi - k_lo), L) and list(share.blocks()) == ([(0, S, k_lo, k_hi)] if k_hi > k_lo else [])
        got = share.to_array()
        assert np.array_equal(got, full[:S, k_lo:k_hi].reshape(-1, L))
        sub = np.empty((2, N), dtype=np.complex64)
        if share.shape[0] >= 3:
            share.gather(sub, 1, 3, N)
            assert np.array_equal(sub, got[1:3, :N].astype(np.complex64))
        seen[:, k_lo:k_hi] += 1
        parts.append(np.abs(got[:, :4]).astype(np.float32))
    assert (seen == 1).all()
    # gather on one rank: the block comes back as (S, K, C)
    one = gather_frame_columns(np.abs(full[:S, :K, :4]).reshape(-1, 4).astype(np.float32), S, K, 0, 1)
    assert one.shape == (S, K, 4) and np.array_equal(one, np.abs(full[:S, :K, :4]).astype(np.float32))


def test_read_ahead_depth_follows_the_container(tmp_path, monkeypatch):
    """run_extraction's reader threads: one variable ahead for an uncompressed container, up to six (bounded by the
    variables, the cores and a quarter of the available memory) for a compressed one; AMCX_READ_AHEAD overrides."""
    import scipy.io
    from amcpy_amd import feature_extraction as fe
    from amcpy_amd.matfile import compressed_variable_bytes
    x = np.asfortranarray(np.random.default_rng(0).standard_normal((2, 30, 64)) + 0j)
    plain, packed = tmp_path / "p.mat", tmp_path / "z.mat"
    scipy.io.savemat(str(plain), {"x": x})
    scipy.io.savemat(str(packed), {"x": x}, do_compression=True)
    assert compressed_variable_bytes(plain) == 0 and compressed_variable_bytes(tmp_path / "absent.mat") == 0
    inflated = compressed_variable_bytes(packed)
    assert 2 * x.real.nbytes <= inflated <= 2 * x.real.nbytes + 256
    monkeypatch.delenv("AMCX_READ_AHEAD", raising=False)
    assert fe._read_ahead(0, 6) == 1
    cores = os.cpu_count() or 2
    assert fe._read_ahead(inflated, 6) == min(6, cores - 1) and fe._read_ahead(inflated, 2) == min(2, cores - 1)
    assert fe._read_ahead(1 << 50, 6) == 1                      # a variable that does not fit: one at a time
    monkeypatch.setenv("AMCX_READ_AHEAD", "4")
    assert fe._read_ahead(0, 6) == 4


def _alive(pid: int) -> bool:
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    try:                                   # a zombie still answers kill(0)
        return open(f"/proc/{pid}/stat").read().split(")")[1].split()[0] != "Z"
    except OSError:
        return False


def test_bench_self_launch_process_handling(tmp_path):
    """`python bench.py --gpus N` without a launcher starts its own ranks (bench.self_launch; the reference's
    run_extraction forks its own workers, feature_extraction.py:89-97).  The process handling, on stand-in rank
    scripts (no GPU here): every rank gets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, only rank 0 owns stdout, a rank's
    non-zero exit becomes the job's and the ranks still running are killed, a time limit kills them all, and a
    clean run returns 0."""
    import time
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent("""
        import os, sys, time, pathlib
        r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["LOCAL_RANK"] == str(r) and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
        pathlib.Path(sys.argv[2], f"pid{r}").write_text(str(os.getpid()))
        print(f"line from rank {r} of {w}", flush=True)
        mode = sys.argv[1]
        if mode == "ok":
            sys.exit(0)
        if mode == "fail1" and r == 1:
            sys.exit(7)
        time.sleep(600)
    """))

    def run(mode, n, limit):
        d = tmp_path / mode
        d.mkdir()
        code = ("import sys; sys.path.insert(0, %r); import bench; "
                "sys.exit(bench.self_launch(%d, [%r, %r], %r, script=%r, build=False))"
                % (str(REPO), n, mode, str(d), limit, str(script)))
        t0 = time.time()
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
        pids = [int((d / f"pid{k}").read_text()) for k in range(n)]
        return r, pids, time.time() - t0

    r, pids, _ = run("ok", 3, 60.0)
    assert r.returncode == 0, r.stderr
    assert r.stdout.splitlines() == ["line from rank 0 of 3"]                       # ranks 1, 2 speak on stderr
    assert "line from rank 1 of 3" in r.stderr and "line from rank 2 of 3" in r.stderr
    r, pids, took = run("fail1", 2, 60.0)
    assert r.returncode == 7 and "rank 1 exited with 7" in r.stderr, (r.returncode, r.stderr)
    assert took < 40 and not any(_alive(p) for p in pids), pids                  # rank 0 was asleep for 600 s
    r, pids, took = run("hang", 2, 2.0)
    assert r.returncode == 124 and not any(_alive(p) for p in pids), (r.returncode, r.stderr)
    # the parent dying takes the ranks with it (PR_SET_PDEATHSIG)
    d = tmp_path / "orphan"
    d.mkdir()
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.self_launch(2, ['hang', %r], 300.0, script=%r, build=False))" % (str(REPO), str(d), str(script)))
    parent = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    deadline = time.time() + 30
    while time.time() < deadline and not ((d / "pid0").exists() and (d / "pid1").exists()):
        time.sleep(0.05)
    time.sleep(0.2)
    pids = [int((d / f"pid{k}").read_text()) for k in range(2)]
    parent.kill()
    parent.wait()
    deadline = time.time() + 10
    while time.time() < deadline and any(_alive(p) for p in pids):
        time.sleep(0.05)
    assert not any(_alive(p) for p in pids), pids


def test_bench_refuses_a_launch_that_disagrees_with_gpus(tmp_path):
    """--gpus 2 under a launcher that started one rank is an error (nothing is silently run as one rank); --gpus 2 with
    no launcher starts two ranks, which on a box without GPUs say so and exit non-zero, leaving nothing behind."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    bad = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--no-cpu-baseline"],
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), cwd=str(tmp_path),
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in bad.stderr
    import torch
    if torch.cuda.device_count() == 0:
        r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--dist-backend", "gloo",
                            "--launch-timeout", "100"], env=env, cwd=str(tmp_path), capture_output=True, text=True,
                           timeout=200)
        assert r.returncode != 0 and "no GPU of its own" in r.stderr and not r.stdout.strip(), (r.stdout, r.stderr[-1500:])


def _marker_features(block):
    """A stand-in engine: 18 columns that depend on every sample of the frame."""
    block = np.asarray(block)
    base = np.abs(block).sum(axis=1, dtype=np.float64) + np.real(block[:, 0])
    return (base[:, None] * np.arange(1, 19)[None, :]).astype(np.float32)


@pytest.mark.parametrize("n_snr,n_frames,devices", [(3, 40, [0, 1, 2, 3]), (5, 3, [0, 1, 2, 3]), (2, 9, [0, 0]),
                                                    (1, 1, [0, 1, 2])])
def test_device_fan_out_cuts_and_reassembles(n_snr, n_frames, devices):
    """feature_extraction.DeviceFanOut -- several GPUs from ONE process, what `extract --devices` drives -- with
    stand-in engines (no GPU here): a whole container is cut along its frame axis when that balances and along the
    snr-major flattening otherwise, every frame is computed exactly once, and the rows come back in the snr-major
    order of a single engine's result, from Fortran- and C-ordered containers alike."""
    from amcpy_amd import feature_extraction as fe
    from amcpy_amd.sharding import shard_by_frames
    rng = np.random.default_rng(5)
    N = 12
    full = rng.standard_normal((n_snr, n_frames + 2, N + 3)) + 1j * rng.standard_normal((n_snr, n_frames + 2, N + 3))
    for parsed in (np.asfortranarray(full), np.ascontiguousarray(full)):
        fan = fe.DeviceFanOut(N, devices, threads=2)
        seen = []

        class Stub:
            def __init__(self, dev):
                self.device, self.stats = dev, {}

            def __call__(self, rows):
                arr = rows.to_array()[:, :N]
                seen.append((self.device, type(rows).__name__, arr.shape[0]))
                return _marker_features(arr)

        fan.engines = [Stub(d) for d in devices]
        rows = fe.FrameRows(parsed, n_snr, n_frames)
        got = fan(rows)
        want = _marker_features(rows.to_array()[:, :N])
        assert got.shape == (n_snr * n_frames, 18) and np.array_equal(got, want)
        assert sum(n for _, _, n in seen) == n_snr * n_frames
        by_frames = shard_by_frames(n_snr, n_frames, len(devices))
        assert {t for _, t, n in seen if n} == ({"FrameColumns"} if by_frames else {"FrameRows"})
        # a range that starts inside an snr row goes by the flattening
        part = rows.slice(1, n_snr * n_frames)
        assert np.array_equal(fan(part), want[1:])
        fan.close()


def test_device_fan_out_reports_every_failing_device():
    from amcpy_amd import feature_extraction as fe
    fan = fe.DeviceFanOut(8, [0, 1, 2], threads=1)

    def ok(rows):
        return np.zeros((rows.shape[0], 18), np.float32)

    def boom(rows):
        raise OSError("no such device")

    fan.engines = [ok, boom, boom]
    x = np.zeros((1, 9, 8), np.complex64)
    with pytest.raises(RuntimeError) as err:
        fan(fe.FrameRows(x, 1, 9))
    assert "device 1: OSError" in str(err.value) and "device 2: OSError" in str(err.value)
    fan.close()
    with pytest.raises(ValueError):
        fe.DeviceFanOut(8, [])
    with pytest.raises(ValueError):
        fe.DeviceFanOut(8, [-1])


def test_bench_scaling_block_at_eight_ranks():
    """What a multi-rank bench line adds (bench._scaling_block) at the driver's largest N, on made-up ranks of which one is
    slow: `per_rank` names it, `scaling_efficiency` = aggregate / (8 x the best rank's own kernel-only rate), `rank_balance`
    = slowest / fastest; the block stays small enough for the whole line to fit the driver's 4 KB tail; and the measured
    FMA ceiling lands in roofline.secondary.measured with the kernel's own rate against it (bench._secondary)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", REPO / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    frames = 6 * 26 * 4096
    per_rank = [{"rank": r, "dev": r, "bus": f"0000:{0x05 + 0x10 * r:02x}:00.0",
                 "ms": [3.1234567 if r != 5 else 3.9, 3.05, 3.3 if r != 5 else 4.2], "frames": frames,
                 "wall_s": 0.0631 if r != 5 else 0.0785, "fma_G": 851.234567, "fma_GHz": 2.1234567} for r in range(8)]
    value = 8 * frames * 20 / (0.0785)                      # 20 steps, the slowest rank's wall
    blk = bench._scaling_block(per_rank, value, 8)
    best = frames / 3.1234567e-3
    assert abs(blk["scaling_efficiency"] - value / (8 * best)) < 1e-12
    assert abs(blk["rank_balance"] - 3.1234567 / 3.9) < 1e-9 and abs(blk["sum_of_rank_rates"] - (7 * best + frames / 3.9e-3)) < 1
    assert [r["rank"] for r in blk["per_rank"]] == list(range(8)) and blk["per_rank"][5]["ms"][0] == 3.9
    text = json.dumps(bench._rounded(blk))
    assert len(text) < 1300, len(text)                      # + ~1.7 KB of an N > 1 line without it + ~0.65 KB of h2d_fanout: under 4 096
    # (the replayed half of roofline.secondary: test_replayed_counters_are_bound_to_the_timed_binary)
    assert bench._secondary(200e6, {"error": "boom"}, ident={"error": "x"})["measured"] == {"error": "boom"}


def test_replayed_counters_are_bound_to_the_timed_binary(tmp_path):
    """roofline.traffic and roofline.secondary are replays of committed profiles -- and only of profiles TAKEN ON THE
    BINARY THAT IS RUNNING: every summary carries the SHA-256 of the library's gfx950 code object and of the dominant
    kernel's machine code (tools/prof_summary.py), bench.py computes the same two for the library it loaded
    (bench._binary_identity) and emits a replay only when they agree -- the kernel's digest (a change to another kernel
    leaves the profile valid), or the whole code object's.  A mismatching digest gives null and says why; so does a
    host without the LLVM tools.  (Round 5's line carried the previous binary's counters without a way to tell.)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", REPO / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    ident = bench._binary_identity("amcx_features18_wave_kernel<2048>")
    assert "error" not in ident and len(ident["code_object_sha256"]) == 64 and len(ident["kernel_sha256"]) == 64, ident
    committed = json.loads((REPO / "amcpy_amd" / "csrc" / "codeobj.json").read_text())
    assert ident["code_object_sha256"] == committed["code_object_sha256"]          # the built library is the tree's
    other = bench._binary_identity("amcx_features18_wave_kernel<4096>")
    assert other["kernel_sha256"] != ident["kernel_sha256"] and other["code_object_sha256"] == ident["code_object_sha256"]

    def pmc(name, **digests):
        (tmp_path / name).write_text(json.dumps({"frame_size": 2048, "hbm_bytes_per_frame": 16500.0, **digests}))

    frames = 6 * 26 * 4096
    # 1. a summary of ANOTHER binary (or one without any digest, as every profile before round 6): nothing is replayed
    pmc("r5_n2048_pmc_traffic.json")
    pmc("r6_n2048_pmc_traffic.json", code_object_sha256="0" * 64, kernel_sha256="1" * 64)
    traffic, why = bench._pmc_traffic(frames, 2048, ident, directory=tmp_path)
    assert traffic is None and "r6_n2048_pmc_traffic.json" in why
    # 2. the same kernel inside a library whose other kernels changed: valid
    pmc("r6b_n2048_pmc_traffic.json", code_object_sha256="0" * 64, kernel_sha256=ident["kernel_sha256"])
    traffic, src = bench._pmc_traffic(frames, 2048, ident, directory=tmp_path)
    assert traffic == 16500.0 * frames and src == "profiles/r6b_n2048_pmc_traffic.json"
    # 3. the whole code object agrees (a summary that names no kernel digest)
    (tmp_path / "r6b_n2048_pmc_traffic.json").unlink()
    pmc("r6c_n2048_pmc_traffic.json", code_object_sha256=ident["code_object_sha256"])
    assert bench._pmc_traffic(frames, 2048, ident, directory=tmp_path)[0] == 16500.0 * frames
    # 4. no identity (no llvm-objcopy on the host): null, with the reason
    traffic, why = bench._pmc_traffic(frames, 2048, {"error": "FileNotFoundError"}, directory=tmp_path)
    assert traffic is None and "digest" in why
    # the instruction budget behind roofline.secondary: the same rule
    fma = {"wave_instr_per_s": 850e9, "clock_GHz": 2.1}
    (tmp_path / "r6_wave_budget.json").write_text(json.dumps({"valu_instr_per_frame": 3400.0, "kernel_sha256": "2" * 64}))
    sec = bench._secondary(200e6, fma, ident=ident, directory=tmp_path)
    assert sec["replayed"] is None and "valu_instr_per_frame" not in sec and "ratio" not in sec["measured"]
    assert sec["measured"]["fma_Gwaveinstr_per_s"] == 850.0                          # what THIS run measured stays
    (tmp_path / "r6b_wave_budget.json").write_text(json.dumps({"valu_instr_per_frame": 3400.0, "kernel_sha256": ident["kernel_sha256"]}))
    sec = bench._secondary(200e6, fma, ident=ident, directory=tmp_path)
    assert abs(sec["measured"]["ratio"] - 3400.0 * 200e6 / 850e9) < 1e-12 and sec["source"].startswith("profiles/r6b_wave_budget.json")
    brief = bench._secondary(200e6, fma, brief=True, ident=ident, directory=tmp_path)
    assert set(brief) == {"valu_instr_per_frame", "source", "measured"} and brief["measured"]["ratio"] == sec["measured"]["ratio"]
    # the committed profiles of THIS tree: whatever bench.py would replay now names this binary
    traffic, src = bench._pmc_traffic(frames, 2048, ident)
    if traffic is not None:
        d = json.loads((REPO / src).read_text())
        assert bench._same_binary(d, ident)


def test_numa_mapper_on_a_fake_sysfs_tree(tmp_path):
    """Host placement (include/amcx.h, ABI 4): amcx_numa_place reads <sysfs>/bus/pci/devices/<bdf>/{numa_node,
    local_cpulist}.  A fake tree of two nodes and four devices (two per socket, the SMT siblings in the second range
    as on the 2-socket EPYC hosts of an MI355X node) must give each device its socket's CPU set; a platform that says -1,
    a device that is not there and a malformed list bind nothing; and staging_threads_per_device splits a node's CPUs
    among the engines on it -- those of them this process may use at all."""
    from amcpy_amd import _lib
    from amcpy_amd.feature_extraction import staging_threads_per_device
    tree = {"0000:05:00.0": (0, "0-15,32-47"), "0000:15:00.0": (0, "0-15,32-47"),
            "0000:85:00.0": (1, "16-31,48-63"), "0000:95:00.0": (1, "16-31,48-63"),
            "0000:a5:00.0": (-1, "0-63"), "0000:b5:00.0": (1, "garbage")}
    for bdf, (node, cpus) in tree.items():
        d = tmp_path / "bus" / "pci" / "devices" / bdf
        d.mkdir(parents=True)
        (d / "numa_node").write_text(f"{node}\n")
        (d / "local_cpulist").write_text(cpus + "\n")
    node0 = list(range(0, 16)) + list(range(32, 48))
    node1 = list(range(16, 32)) + list(range(48, 64))
    places = [_lib.numa_place(bdf, str(tmp_path)) for bdf in list(tree)[:4]]
    assert places == [(0, node0), (0, node0), (1, node1), (1, node1)]
    assert _lib.numa_place("0000:05:00.0".upper(), str(tmp_path)) == (0, node0)            # HIP may hand out upper case
    for bdf in ("0000:a5:00.0", "0000:b5:00.0", "0000:ff:00.0", "../../etc", ""):
        assert _lib.numa_place(bdf, str(tmp_path)) == (-1, []), bdf
    # every CPU allowed: 32 local CPUs over two engines each -> the wanted 8 (or the default 8); 12 wanted -> 12
    assert staging_threads_per_device(places, None, allowed=range(64)) == [8, 8, 8, 8]
    assert staging_threads_per_device(places, 12, allowed=range(64)) == [12, 12, 12, 12]
    assert staging_threads_per_device(places, 64, allowed=range(64)) == [16, 16, 16, 16]
    # a cpuset of 0-19: node 0 keeps 16 CPUs (8 each), node 1 only 4 (2 each)
    assert staging_threads_per_device(places, None, allowed=range(20)) == [8, 8, 2, 2]
    # nothing local allowed -> the old even split of what is allowed; unknown placement likewise; never below 1
    assert staging_threads_per_device(places[:2], None, allowed=range(16, 24)) == [4, 4]
    assert staging_threads_per_device([(-1, [])] * 4, None, allowed=range(8)) == [2, 2, 2, 2]
    assert staging_threads_per_device([(-1, [])] * 4, None, allowed=range(2)) == [1, 1, 1, 1]


def test_extract_cli_device_arguments(tmp_path):
    from amcpy_amd import main as cli
    assert cli._parse_devices("0,1,3") == [0, 1, 3] and cli._parse_devices("2") == [2] and cli._parse_devices("0,0") == [0, 0]
    for bad in ("", "a,b", "0,-1"):
        with pytest.raises(SystemExit):
            cli._parse_devices(bad)
    import torch
    if torch.cuda.device_count() == 0:
        with pytest.raises(SystemExit):
            cli._parse_devices("all")                            # says that no device is visible
    with pytest.raises(SystemExit):
        cli.main(["extract", "--root", str(tmp_path), "--device", "0", "--devices", "0,1"])
    from amcpy_amd.config import Config, Paths
    from amcpy_amd.feature_extraction import run_extraction
    with pytest.raises(ValueError):
        run_extraction(Config(paths=Paths(root=tmp_path)), devices=[0, 1], device=0)


_PLACEMENT_WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, os.environ["AMCX_REPO"])
    from pathlib import Path
    from amcpy_amd.config import Config, Paths, SignalConfig
    from amcpy_amd import feature_extraction as fe
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class OnDevice:                                   # an engine that says which GPU it computes on
        device = int(os.environ["AMCX_TEST_DEVICES"].split(",")[rank])
        def __call__(self, rows):
            return np.zeros((rows.shape[0], 18), dtype=np.float32)

    fe.default_engine = lambda *a, **k: OnDevice()
    cfg = Config(paths=Paths(root=Path(os.environ["AMCX_ROOT"])),
                 signals=SignalConfig(snr_values={0: "0", 1: "10", 2: "20"}, num_frames=7, frame_size=16))
    try:
        fe.run_extraction(cfg, verbose=False)
    except RuntimeError as exc:
        print("RAISED", rank, exc)
        dist.destroy_process_group()
        sys.exit(3)
    print("NO ERROR", rank)
    dist.destroy_process_group()
''')


def test_ranks_sharing_a_device_are_refused(tmp_path):
    """Several ranks whose engines sit on ONE device of one host (a launcher whose ranks never chose their GPU) would be
    silently correct and world-size times slow: run_extraction all-gathers (host, device) once and every rank raises,
    unless AMCX_SHARE_GPU=1 says the sharing is meant.  Distinct devices pass."""
    import scipy.io
    from amcpy_amd.config import Config, Paths, SignalConfig
    rng = np.random.default_rng(3)
    cfg = Config(paths=Paths(root=tmp_path),
                 signals=SignalConfig(snr_values={0: "0", 1: "10", 2: "20"}, num_frames=7, frame_size=16))
    cfg.paths.ensure_dirs()
    scipy.io.savemat(str(cfg.paths.mat_data / cfg.paths.mat_filename),
                     {cfg.signals.mat_info[m]: (rng.standard_normal((3, 7, 16)) + 1j * rng.standard_normal((3, 7, 16)))
                      for m in cfg.signals.modulations_with_noise})
    script = tmp_path / "placement_worker.py"
    script.write_text(_PLACEMENT_WORKER)

    def run(devices, share):
        port = _free_port()
        procs = []
        for r in range(2):
            env = {k: v for k, v in os.environ.items() if k != "AMCX_SHARE_GPU"}
            env.update(RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), AMCX_REPO=str(REPO),
                       AMCX_ROOT=str(tmp_path), AMCX_TEST_DEVICES=devices, PYTHONDONTWRITEBYTECODE="1")
            if share:
                env["AMCX_SHARE_GPU"] = "1"
            procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        outs = [p.communicate(timeout=120)[0] for p in procs]
        return [p.returncode for p in procs], outs

    codes, outs = run("0,0", share=False)
    assert codes == [3, 3], "\n".join(outs)
    assert all("ranks 0 and 1 both compute on device 0" in o for o in outs), outs
    codes, outs = run("0,0", share=True)
    assert codes == [0, 0] and all("NO ERROR" in o for o in outs), "\n".join(outs)
    codes, outs = run("0,1", share=False)
    assert codes == [0, 0] and all("NO ERROR" in o for o in outs), "\n".join(outs)


@pytest.mark.parametrize("name,flags", [
    ("asan_ubsan", ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all"]),
    ("tsan", ["-O1", "-g", "-fsanitize=thread"]),
    ("plain", ["-O2"])])
def test_staging_half_under_the_sanitizers(tmp_path, name, flags):
    """The host half of the real-data path -- amcx_upload.h: the hand-rolled fork-join Pool, stage_runs over memory
    and FILE sources, the layout classifier -- compiled with g++ alone (no HIP) and run under AddressSanitizer +
    UBSan, under ThreadSanitizer, and as the product builds it: 60 random layouts against an element-by-element
    restatement (30 of them also read from a file, and from that file cut short), 1 000 back-to-back Pool::run calls
    with resizes and sleeping workers in between, three pools staging at once (one per device: DeviceFanOut).  The
    reference's own defect on this path is a threading one (feature_extraction.py:22-39,74)."""
    exe = tmp_path / f"stage_fuzz_{name}"
    cmd = ["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-pthread", *flags,
           str(REPO / "tests" / "host_san" / "stage_fuzz.cc"), "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1", ASAN_OPTIONS="detect_leaks=1",
               UBSAN_OPTIONS="print_stacktrace=1")
    for seed in ("2026", "7"):
        r = subprocess.run([str(exe), seed], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "STAGE_FUZZ_OK" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-6000:])
        assert "WARNING: ThreadSanitizer" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr \
            and "runtime error" not in r.stderr, r.stderr[-6000:]


def test_kernel_resources_match_the_committed_table():
    """The register / spill / scratch claims made in DESIGN.md section 4.5 and in the kernel headers are read off the
    BUILT library (its gfx950 code object's metadata, tools/resource_usage.py) and held to
    amcpy_amd/csrc/kernel_resources.json: a kernel that starts to spill, grows past a wave-per-SIMD step (128 / 168 /
    256 VGPRs) or appears / disappears without the table being updated fails here.  Product kernels only: the library
    must hold no experiment (no pair kernel, no one-wave N = 8192)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("resource_usage", REPO / "tools" / "resource_usage.py")
    ru = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ru)
    got = {k: {f: v[f] for f in ("vgpr", "spill", "scratch")} for k, v in ru.read().items()}
    want = __import__("json").loads((REPO / "amcpy_amd" / "csrc" / "kernel_resources.json").read_text())
    assert sorted(got) == sorted(want), (sorted(set(got) ^ set(want)))
    drift = {k: (got[k], want[k]) for k in want if got[k] != want[k]}
    assert not drift, f"kernel resources differ from amcpy_amd/csrc/kernel_resources.json (python tools/resource_usage.py --update after a deliberate change): {drift}"
    names = " ".join(got)
    assert "pair_kernel" not in names and "wave_kernel<8192>" not in names
    # the throughput kernels' occupancy steps, as the headers state them
    assert got["amcx_features18_wave_kernel<2048>"]["vgpr"] <= 128            # 4 waves per SIMD
    assert got["amcx_features18_wave_kernel<1024>"]["vgpr"] <= 168            # 3 waves per SIMD
    assert got["amcx_features18_wave_kernel<4096>"]["vgpr"] <= 256            # 2 waves per SIMD
    assert got["amcx_features18_wave_kernel<2048>"]["scratch"] <= 24          # three lane-dependent dwords + one fp64 value in the per-batch finaliser
    # N = 32768: sixteen waves per frame = 4 per SIMD: 128 registers, and next to nothing spilled in the frame loop (a
    # spilled register is a 256-byte transaction per wave that reaches HBM: the first version moved 4x the frame's bytes)
    assert got["group::amcx_features18_group_kernel<16>"]["vgpr"] <= 128 and got["group::amcx_features18_group_kernel<16>"]["spill"] <= 16
    assert got["group::amcx_features18_group_kernel<8>"]["vgpr"] <= 256


def test_cited_profiles_exist():
    """DESIGN.md describes the tree as it is and HISTORY.md how it got there; every number in either names the
    profiles/ file it comes from.  A citation of a file that is not in the tree is a claim without evidence: every
    `profiles/<name>` and every back-quoted `r<round>_<name>.{json,jsonl,txt,csv}` in the documents must exist
    (patterns with a * are families, not files)."""
    have = {p.name for p in (REPO / "profiles").iterdir()}
    ext = r"(?:jsonl|json|txt|csv|patch)"
    for doc in ("DESIGN.md", "HISTORY.md", "README.md", "INTEGRATION.md", "tools/README.md", "profiles/README.md"):
        text = (REPO / doc).read_text()
        names = set(re.findall(rf"profiles/([A-Za-z0-9_.\-]+\.{ext})\b", text))
        names |= set(re.findall(rf"`((?:r\d+[a-z]?_)[A-Za-z0-9_.\-]+\.{ext})`", text))
        missing = sorted(n for n in names if n not in have and not (REPO / "tools" / "experiments" / n).exists())
        assert not missing, f"{doc} cites profiles that are not in the tree: {missing}"
    assert len((REPO / "DESIGN.md").read_text().splitlines()) <= 400, "DESIGN.md describes the tree in at most 400 lines; narratives go to HISTORY.md"


def test_product_sources_carry_no_laboratory():
    """Product and laboratory are separate: nothing under amcpy_amd/csrc or include/ mentions an experiment / ablation
    switch (AMCX_EXP_* / AMCX_ABL_*), the experiment kernels live in tools/experiments/, and the patch that puts the
    branches back (tools/experiments/r5_lab_branches.patch, tools/experiments/make_lab.sh) names the commit it is
    against.  AMCX_WAVE_STAMPS (tools/wave_clock.hip, wave_stamps.hip) and AMCX_PRIO_MASK are the two kept knobs."""
    import re
    hits = []
    for p in sorted((REPO / "amcpy_amd" / "csrc").iterdir()) + sorted((REPO / "include").iterdir()):
        if p.suffix in (".h", ".hip", ".py"):
            for i, line in enumerate(p.read_text().splitlines(), 1):
                if re.search(r"AMCX_(EXP|ABL)_[A-Z0-9_]+|if \(true\)|if \(false\)", line) and p.name != "build.py":
                    hits.append(f"{p.name}:{i}: {line.strip()[:100]}")
    assert not hits, hits
    assert not (REPO / "amcpy_amd" / "csrc" / "amcx_pair_kernel.h").exists()
    assert not (REPO / "amcpy_amd" / "csrc" / "amcx_fixup_kernel.h").exists()
    exp = REPO / "tools" / "experiments"
    assert (exp / "amcx_pair_kernel.h").exists() and (exp / "amcx_fixup_kernel.h").exists()
    patch = (exp / "r5_lab_branches.patch").read_text()
    for flag in ("AMCX_EXP_WAVES12", "AMCX_EXP_PK_FFT", "AMCX_ABL_NOFFT", "AMCX_ABL_FFT_TAIL_MFMA", "AMCX_EXP_PAIR4096"):
        assert flag in patch, flag
    base = re.search(r"^LAB_BASE=(\w+)", (exp / "make_lab.sh").read_text(), re.M).group(1)
    r = subprocess.run(["git", "-C", str(REPO), "cat-file", "-e", f"{base}^{{commit}}"], capture_output=True)
    if (REPO / ".git").exists():                       # (the GPU box's snapshot has no history)
        assert r.returncode == 0, f"make_lab.sh's base commit {base} is not in this history"


def test_built_library_is_the_committed_gpu_program():
    """amcpy_amd/csrc/codeobj.json holds the SHA-256 of the gfx950 code object (and of its .text) that the committed
    sources build to; the build is reproducible (fixed -cuid, build.py), so the library in the tree -- the one that
    travels to the GPU box -- is held to it.  After a deliberate kernel change: python tools/codeobj_gate.py --update.
    (Round 5's removal of the experiment branches was held to the same gate: byte-identical before and after,
    profiles/r5_codeobj_gate.txt.)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("codeobj_gate", REPO / "tools" / "codeobj_gate.py")
    gate = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gate)
    from amcpy_amd.csrc import build as b
    lib = b.build(force=False, verbose=False)
    got = gate.digests(lib)
    want = __import__("json").loads((REPO / "amcpy_amd" / "csrc" / "codeobj.json").read_text())
    assert got["text_sha256"] == want["text_sha256"], \
        "the built kernels differ from amcpy_amd/csrc/codeobj.json (python tools/codeobj_gate.py --update after a deliberate change)"
    assert got["code_object_sha256"] == want["code_object_sha256"]


def test_run_extraction_resume_skips_complete_files(tmp_path):
    """``run_extraction(cfg, resume=True)`` / ``extract --resume``: the per-modulation output file is the path's
    resume unit.  A file that is complete for THIS configuration is skipped; a missing one, one
    written for another frame count and one cut short are computed again."""
    import scipy.io
    from amcpy_amd.config import Config, Paths, SignalConfig
    from amcpy_amd import feature_extraction as fe
    rng = np.random.default_rng(8)
    cfg = Config(paths=Paths(root=tmp_path),
                 signals=SignalConfig(snr_values={0: "0", 1: "10"}, num_frames=5, frame_size=16))
    cfg.paths.ensure_dirs()
    mods = list(cfg.signals.modulations_with_noise)
    scipy.io.savemat(str(cfg.paths.mat_data / cfg.paths.mat_filename),
                     {cfg.signals.mat_info[m]: (rng.standard_normal((2, 6, 20)) + 1j * rng.standard_normal((2, 6, 20)))
                      for m in mods})
    calls = []

    def compute(block):
        calls.append(block.shape[0])
        return _marker_features(np.asarray(block)[:, :16])

    fe.run_extraction(cfg, compute=compute, verbose=False, resume=True)       # nothing there yet: everything is computed
    assert len(calls) == len(mods)
    ref = {m: scipy.io.loadmat(str(cfg.paths.calculated_features / f"{m}_features.mat"))[cfg.signals.mat_info[m]] for m in mods}
    calls.clear()
    fe.run_extraction(cfg, compute=compute, verbose=False, resume=True)       # everything there: nothing is computed
    assert calls == []
    # one missing, one of another shape (a run with fewer frames), one cut short
    (cfg.paths.calculated_features / f"{mods[0]}_features.mat").unlink()
    scipy.io.savemat(str(cfg.paths.calculated_features / f"{mods[1]}_features.mat"),
                     {"Modulation": mods[1], cfg.signals.mat_info[mods[1]]: np.zeros((2, 3, 18), np.float32)})
    p2 = cfg.paths.calculated_features / f"{mods[2]}_features.mat"
    p2.write_bytes(p2.read_bytes()[:300])
    fe.run_extraction(cfg, compute=compute, verbose=False, resume=True)
    assert len(calls) == 3
    for m in mods:
        got = scipy.io.loadmat(str(cfg.paths.calculated_features / f"{m}_features.mat"))[cfg.signals.mat_info[m]]
        assert got.dtype == np.float32 and np.array_equal(got, ref[m]), m
    calls.clear()
    # round 5 (provenance beside every file): a file of the right SHAPE that was computed for something else is stale.
    # Another frame size -- the shape (n_snr, n_frames, 18) does not say N:
    fe.run_extraction(cfg, compute=compute, verbose=False, resume=True)
    assert calls == []
    cfg12 = Config(paths=Paths(root=tmp_path), signals=SignalConfig(snr_values={0: "0", 1: "10"}, num_frames=5, frame_size=12))
    fe.run_extraction(cfg12, compute=lambda b: (calls.append(b.shape[0]), _marker_features(np.asarray(b)[:, :12]))[1],
                      verbose=False, resume=True)
    assert len(calls) == len(mods)
    calls.clear()
    fe.run_extraction(cfg, compute=compute, verbose=False, resume=True)       # ... and back: all six again
    assert len(calls) == len(mods)
    calls.clear()
    # other SNR labels of the same count
    cfg_snr = Config(paths=Paths(root=tmp_path), signals=SignalConfig(snr_values={0: "0", 1: "20"}, num_frames=5, frame_size=16))
    fe.run_extraction(cfg_snr, compute=compute, verbose=False, resume=True)
    assert len(calls) == len(mods)
    calls.clear()
    fe.run_extraction(cfg, compute=compute, verbose=False, resume=True)
    assert len(calls) == len(mods)
    calls.clear()
    # a file without a record (the reference's own output, or an older build's) is not trusted; one whose input
    # container has been replaced since is not either
    (cfg.paths.calculated_features / f"{mods[3]}_features.provenance.json").unlink()
    container = cfg.paths.mat_data / cfg.paths.mat_filename
    fe.run_extraction(cfg, compute=compute, verbose=False, resume=True)
    assert len(calls) == 1
    calls.clear()
    st = container.stat()
    os.utime(container, ns=(st.st_atime_ns, st.st_mtime_ns + 5_000_000_000))
    fe.run_extraction(cfg, compute=compute, verbose=False, resume=True)
    assert len(calls) == len(mods)
    # the .mat itself still holds exactly what the reference writes (feature_extraction.py:77-81)
    keys = {k for k in scipy.io.loadmat(str(cfg.paths.calculated_features / f"{mods[0]}_features.mat")) if not k.startswith("__")}
    assert keys == {"Modulation", cfg.signals.mat_info[mods[0]]}
    calls.clear()
    fe.run_extraction(cfg, compute=compute, verbose=False)                    # without resume: the reference's behaviour
    assert len(calls) == len(mods)
