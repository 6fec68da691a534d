#!/usr/bin/env python3
"""Headline benchmark: IQ frames/sec, all 18 features, 2048-sample complex64
frames (BASELINE.json metric), on N MI355X of one node.

    python bench.py                       # N=1, defaults finish in ~2-3 min
    python bench.py --gpus N              # starts its own N ranks (one fresh child process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

`--gpus N` without a launcher's WORLD_SIZE in the environment launches itself, as the
reference's run_extraction forks its own workers (feature_extraction.py:89-97): the parent
touches neither HIP nor torch.cuda, starts N children with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* set, passes rank 0's JSON line through, exits non-zero if any child does and takes
the children down with it (signal, time limit, or its own death).  Under an external launcher
the ranks are the launcher's.

A *step* is one pass of the hot path over one rank's resident shard: the
BASELINE configs[1] shape -- 6 modulations x 26 SNR x 4096 frames x 2048
samples of synthetic IQ (10.47 GB), generated in HBM before the timed region
(SURVEY.md section 8d).  With N > 1 every rank holds its own shard of that
shape (frames shard embarrassingly: no collective on the data path, weak
scaling); the only cross-rank traffic is the timing barrier/MAX.

One JSON line on stdout (rank 0).  Extra objects:
  roofline     achieved = algorithmic bytes per launch ((8*N+72) B x frames)
               / mean launch duration from HIP events on the launch stream,
               against the 8 TB/s HBM peak (MI355X_MICROARCH.md); `traffic`
               carries the PMC-derived HBM bytes per launch when a committed
               profile summary provides it (profiles/*.json), else null.
               roofline.secondary.measured: the board's instruction-issue ceiling under its power cap, run
               live after the timed region (amcx_probe_fma_rate, ~1 s of independent v_fma_f32), and this
               kernel's own instruction rate against it; roofline.frac_of_measured_read_peak.
  per_rank     (N > 1) one entry per rank: rank, device, PCI bus id, ms = [mean, min, max] launch, frames per
               launch, its own wall seconds, its device's FMA ceiling -- if the aggregate is short of N x, the
               line says which rank, device or clock was slow; scaling_efficiency = value / (N x the best rank's
               own kernel-only rate), rank_balance = slowest / fastest rank's kernel-only rate.
  h2d_fanout   (N > 1) the real-data path over all N devices from ONE process (DeviceFanOut): rank 0, before it
               touches a GPU, runs a fresh child that uploads one configs[1] modulation (3.49 GB complex128).
  cpu_baseline the oracle's reference-shaped per-frame evaluator on the host
               cores (kind "port"), on the configs[0] shape, rank 0 at N=1
               only; run BEFORE the GPU is initialised (worker processes).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_MODS, N_SNR, N_FRAMES, FRAME_SIZE = 6, 26, 4096, 2048
CPU_SAMPLE = (6, 2, 500, 2048)  # BASELINE configs[0]


# ----------------------------------------------------------------------------
# CPU baselines (the oracle, reference-shaped) -- run before any GPU initialisation
# ----------------------------------------------------------------------------
_BLOCK = None


def _cpu_init(jobs_by_slot, counter):
    """Pool initialiser: import numpy + the oracle and build this worker's block once, outside
    the timed region (fork/import used to be most of a 1.6 s measurement)."""
    global _BLOCK
    import numpy as np
    from amcpy_amd import synth
    from oracle import iq_features_oracle as orc
    with counter.get_lock():
        slot = counter.value
        counter.value += 1
    mod, mi, si, snr, n_frames, N = jobs_by_slot[slot % len(jobs_by_slot)]
    _BLOCK = synth.host_block(mod, snr, n_frames, N, seed=1000 + 10 * mi + si).astype(np.complex128)
    orc.calculate_features(range(1, 19), _BLOCK[0])            # warm every code path


def _cpu_spin(seconds):
    """Cycle over this worker's block for `seconds`; returns (frames, elapsed)."""
    from oracle import iq_features_oracle as orc
    n = _BLOCK.shape[0]
    t0 = time.perf_counter()
    done = 0
    acc = 0.0
    while True:
        row = orc.calculate_features(range(1, 19), _BLOCK[done % n])
        acc += row[5]
        done += 1
        if (done & 7) == 0 and time.perf_counter() - t0 >= seconds:
            break
    return done, time.perf_counter() - t0, acc


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _host_cores():
    """Cores this process may really use: the affinity mask, capped by a cgroup CPU quota if one is set
    (a 1-GPU share of a large host shows all 256 cores in the mask but schedules on 16)."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", None)):
        try:
            txt = open(path).read().strip()
            if parse is not None:
                quota, period = parse(txt)
                if quota != "max":
                    cores = max(1, min(cores, int(float(quota) / float(period) + 0.5)))
            else:
                quota = int(txt)
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip())
                if quota > 0:
                    cores = max(1, min(cores, int(quota / period + 0.5)))
            break
        except Exception:
            continue
    return cores


def cpu_baseline(max_procs: int | None = None, seconds: float = 10.0):
    """Frames/s of the oracle's per-frame, per-feature evaluator (same redundancy class as the
    reference: 9x moments, 4x instantaneous values per frame): one worker process per host core,
    each cycling over its own (modulation, SNR) block of the configs[0] shape for `seconds` of
    wall time AFTER the pool is up and warm."""
    import multiprocessing as mp
    from amcpy_amd import synth
    n_mods, n_snr, n_frames, N = CPU_SAMPLE
    procs = max(1, min(_host_cores(), max_procs or _host_cores()))
    grid = synth.snr_grid(n_snr)
    jobs = [(mod, mi, si, float(grid[si]), min(n_frames, 64), N)
            for mi, mod in enumerate(synth.MODS6[:n_mods]) for si in range(n_snr)]
    ctx = mp.get_context("fork")
    counter = ctx.Value("i", 0)
    with ctx.Pool(procs, initializer=_cpu_init, initargs=(jobs, counter)) as pool:
        pool.map(_cpu_spin, [0.05] * procs, chunksize=1)           # every worker up and warm
        solo = pool.map(_cpu_spin, [1.5], chunksize=1)[0]          # ONE worker busy: a core's own rate
        one_core = solo[0] / solo[1]                               # (value / this = the cores the pool effectively had)
        t0 = time.perf_counter()
        res = pool.map(_cpu_spin, [seconds] * procs, chunksize=1)
        wall = time.perf_counter() - t0
    frames = sum(r[0] for r in res)
    value = frames / wall
    return {
        "value": value, "unit": "frames/s", "cores": procs, "kind": "port", "cpu_model": _cpu_model(),
        "one_core_frames_per_s": one_core, "effective_cores": value / one_core,
        "sample": f"{procs} procs x {wall:.1f} s, each cycling 64 frames of one (mod,SNR) block of configs[0], oracle.calculate_features per frame: {frames} frames",
        "extrapolated_configs1_seconds": N_MODS * N_SNR * N_FRAMES / value,
    }


def _ref_shaped_child(job):
    """One modulation the way the reference's _modulation_process runs it (feature_extraction.py:42-82):
    a Queue of (signal, snr, frame) items, num_threads daemon worker threads each storing
    calculate_features(1..18) into a shared float32 matrix, queue.join()."""
    import queue
    import threading
    import numpy as np
    from amcpy_amd import synth
    from oracle import iq_features_oracle as orc
    mod, mi, n_snr, n_frames, N, n_threads = job
    grid = synth.snr_grid(n_snr)
    parsed = np.stack([synth.host_block(mod, float(grid[si]), n_frames, N, seed=1000 + 10 * mi + si)
                       for si in range(n_snr)]).astype(np.complex128)
    mat = np.zeros((n_snr, n_frames, 18), dtype=np.float32)
    q = queue.Queue()

    def worker():
        while True:
            sig, si, fi = q.get()
            try:
                mat[si, fi, :] = orc.calculate_features(range(1, 19), sig)
            finally:
                q.task_done()

    t0 = time.perf_counter()
    for _ in range(n_threads):
        threading.Thread(target=worker, daemon=True).start()
    for si in range(n_snr):
        for fi in range(n_frames):
            q.put((parsed[si, fi, 0:N], si, fi))
    q.join()
    return n_snr * n_frames, time.perf_counter() - t0, float(mat[0, 0, 5])


def cpu_baseline_reference_shaped(n_threads: int = 8):
    """The reference's own process/thread structure (feature_extraction.py:58-61,89-97;
    config.py:98): one process per modulation x num_threads=8 GIL-bound threads, on the whole
    BASELINE configs[0] container (6 000 frames).  Uses at most 6 cores however many the host has."""
    import multiprocessing as mp
    from amcpy_amd import synth
    n_mods, n_snr, n_frames, N = CPU_SAMPLE
    jobs = [(mod, mi, n_snr, n_frames, N, n_threads) for mi, mod in enumerate(synth.MODS6[:n_mods])]
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(n_mods) as pool:
        res = pool.map(_ref_shaped_child, jobs, chunksize=1)
    wall = time.perf_counter() - t0
    frames = sum(r[0] for r in res)
    compute_wall = max(r[1] for r in res)
    value = frames / compute_wall
    return {
        "value": value, "unit": "frames/s", "cores": min(n_mods, _host_cores()), "kind": "reference-shaped",
        "cpu_model": _cpu_model(),
        "sample": f"configs[0] whole: {n_mods} procs x {n_threads} threads on a Queue (feature_extraction.py:58-61,89-97); slowest {compute_wall:.1f} s",
        "extrapolated_configs1_seconds": N_MODS * N_SNR * N_FRAMES / value,
    }


# ----------------------------------------------------------------------------
def _binary_identity(kernel_name: str):
    """Which GPU program this process runs: SHA-256 of the loaded library's gfx950 code object and of the machine code of
    `kernel_name` in it (tools/codeobj_gate.py: .hip_fatbin -> clang-offload-bundler; the build is reproducible, so the
    same sources give the same digests anywhere).  Every committed profile summary carries the same two digests
    (tools/prof_summary.py): counters are replayed into the line only for the binary they were taken on."""
    try:
        import importlib.util
        from amcpy_amd import _lib
        spec = importlib.util.spec_from_file_location("codeobj_gate", REPO / "tools" / "codeobj_gate.py")
        gate = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(gate)
        lib = _lib.LIB_PATH
        whole = gate.digests(lib)["code_object_sha256"]
        per_kernel = gate.kernel_digests(lib)
        kern = gate.kernel_digest(per_kernel, kernel_name)
        return {"code_object_sha256": whole, "kernel_sha256": kern, "kernel": kernel_name}
    except Exception as exc:                               # no llvm tools on this host, ...: nothing is replayed then
        return {"error": repr(exc)[:200]}


def _same_binary(record: dict, ident) -> bool:
    """A committed measurement belongs to the running binary if the dominant kernel's machine code is the same (a change
    to another kernel of the library leaves it valid), or the whole code object is."""
    if not ident or "error" in ident or not isinstance(record, dict):
        return False
    if record.get("kernel_sha256") and record.get("kernel_sha256") == ident.get("kernel_sha256"):
        return True
    return bool(record.get("code_object_sha256")) and record.get("code_object_sha256") == ident.get("code_object_sha256")


def _pmc_traffic(frames_per_launch: int, frame_size: int, ident=None, directory=None):
    """(HBM bytes per launch, source) from the newest committed PMC summary for this frame size TAKEN ON THIS BINARY
    (profiles/*pmc*.json carry the digests of the library they were collected on) -- replayed, not measured in this
    run; (None, reason) when there is none: a kernel change invalidates the replay instead of leaving it in the line."""
    best, src, stale = None, None, None
    directory = Path(directory) if directory else REPO / "profiles"
    for p in sorted(directory.glob("*pmc*.json"), key=lambda q: (q.name.split("_")[0], q.name)):   # r1.. < r2.. < r3..
        try:
            d = json.loads(p.read_text())
        except Exception:
            continue
        if d.get("frame_size") != frame_size or not d.get("hbm_bytes_per_frame"):
            continue
        if _same_binary(d, ident):
            best, src = d["hbm_bytes_per_frame"] * frames_per_launch, f"profiles/{p.name}"
        else:
            stale = p.name
    if best is None:
        why = ("no code-object digest for the running library" if not ident or "error" in ident else
               f"no committed PMC pass was taken on this binary (newest for N = {frame_size}: {stale})")
        return None, why
    return best, src


def _parity_block():
    """Where parity is established -- not a replay of numbers: the live gate is `pytest -m gpu` (the driver runs it), whose
    test_full_snr_grid_against_oracle covers 6 modulations x 26 SNRs x 8 frames at N = 1024 / 2048 / 4096 with zero frames
    beyond the unfloored criterion."""
    return {"source": "pytest -m gpu; nothing replayed here"}


def _committed_json(name: str):
    p = REPO / "profiles" / name
    try:
        return json.loads(p.read_text())
    except Exception:
        return None


def _fortran_container(n_snr, n_frames, N, seed=7):
    """A complex128 (n_snr, n_frames, N) array in Fortran order, as scipy.io.loadmat returns the reference's
    variables; random bits, tiled from a 64 MB block (the upload rate does not depend on the values)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    total = n_snr * n_frames * N
    block = (rng.standard_normal(min(total, 1 << 22)) + 1j * rng.standard_normal(min(total, 1 << 22)))
    flat = np.empty(total, dtype=np.complex128)
    for a in range(0, total, block.size):
        flat[a:a + block.size] = block[:min(block.size, total - a)]
    return flat.reshape((n_snr, n_frames, N), order="F")


def h2d_path(dev, frame_size: int = FRAME_SIZE, big: bool = True):
    """The real-data upload path (run_extraction's engine, amcx_ctx_features18_strided_host) on complex128
    containers held the way scipy.io.loadmat returns them (Fortran order, pageable memory): host threads
    stage sample planes into pinned slots and round them to complex64 -> H2D -> device transposition ->
    feature kernel -> D2H of the (F x 18) result.  Reported beside the headline, never as `value`."""
    import shutil
    import tempfile
    import numpy as np
    from amcpy_amd.feature_extraction import FrameRows, HipEngine
    n_mods, n_snr, n_frames, N = CPU_SAMPLE[0], CPU_SAMPLE[1], CPU_SAMPLE[2], frame_size

    def timed(eng, rows, reps):
        eng(rows)                                         # warm: pinned + device slots, kernels, staging threads
        eng(rows)
        t0 = time.perf_counter()
        for _ in range(reps):
            eng(rows)
        wall = time.perf_counter() - t0
        F = rows.shape[0]
        return {"GBps": reps * F * N * 16 / wall / 1e9, "pcie_GBps": reps * eng.stats["pcie_bytes"] / wall / 1e9,
                "frames_per_s": reps * F / wall, "seconds": wall, "chunks_per_call": eng.stats["chunks"],
                "staging_threads": eng.stats["gather_threads"]}

    small = FrameRows(_fortran_container(n_snr, n_frames, N), n_snr, n_frames)
    rec = timed(HipEngine(N, dev.index), small, 2 * n_mods)  # the six modulations run_extraction loops over, twice
    rec["what"] = f"2x{n_mods}x({n_snr},{n_frames},{N}) c128 F-order: stage+round->pinned->H2D->transpose->kernel->D2H; GBps = container bytes/wall"
    rec["round_on_device"] = timed(HipEngine(N, dev.index, round_on_device=True), small, 2 * n_mods)
    if big:
        try:
            rows = FrameRows(_fortran_container(N_SNR, N_FRAMES, N), N_SNR, N_FRAMES)
            rec["configs1_modulation"] = timed(HipEngine(N, dev.index), rows, 2)
            rec["configs1_modulation"]["what"] = f"({N_SNR},{N_FRAMES},{N}) c128 = {N_SNR * N_FRAMES * N * 16 / 1e9:.2f} GB, twice"
            del rows
        except Exception as exc:                           # a host short of 3.5 GB: reported, never fatal to the headline
            rec["configs1_modulation"] = {"error": repr(exc)}
    # the whole drop-in on configs[0]: .mat in (memory-mapped, not decoded), six .mat out
    try:
        import scipy.io
        from amcpy_amd.config import Config, Paths, SignalConfig
        from amcpy_amd.feature_extraction import run_extraction
        root = Path(tempfile.mkdtemp(prefix="amcx_bench_", dir="/dev/shm" if Path("/dev/shm").is_dir() else None))
        try:
            cfg = Config(paths=Paths(root=root), signals=SignalConfig(snr_values={0: "0", 1: "10"}, num_frames=n_frames,
                                                                     frame_size=N))
            cfg.paths.ensure_dirs()
            scipy.io.savemat(str(cfg.paths.mat_data / cfg.paths.mat_filename),
                             {cfg.signals.mat_info[m]: np.asarray(small.parsed) for m in cfg.signals.modulations_with_noise})
            run_extraction(cfg, device=dev.index, verbose=False)
            t0 = time.perf_counter()
            run_extraction(cfg, device=dev.index, verbose=False)
            rec["run_extraction_configs0_seconds"] = time.perf_counter() - t0
        finally:
            shutil.rmtree(root, ignore_errors=True)
    except Exception as exc:                               # reported, never fatal to the headline
        rec["run_extraction_configs0_seconds"] = f"failed: {exc!r}"
    return rec


def fanout_child(n_devices: int, share_gpu: bool, frame_size: int, n_frames: int) -> int:
    """`bench.py --fanout-child N`: the one-process multi-GPU real-data path, in a process of its own.  One configs[1]
    modulation -- (26, n_frames, frame_size) complex128, Fortran-ordered as loadmat returns it -- through
    feature_extraction.DeviceFanOut over N devices (device r %% visible with --share-gpu), once warm, once timed.
    Prints ONE JSON object: GB/s of container bytes, frames/s, per-device seconds / frames / placement."""
    import numpy as np  # noqa: F401
    from amcpy_amd import _lib
    from amcpy_amd.feature_extraction import DeviceFanOut, FrameRows
    lib = _lib.load(skip_torch=True)                       # host-buffer path: no torch in this process
    have = lib.amcx_device_count()
    if have <= 0:
        print(json.dumps({"error": "no gfx950 device visible"}))
        return 0
    devices = [r % have for r in range(n_devices)] if share_gpu else list(range(n_devices))
    if max(devices) >= have:
        print(json.dumps({"error": f"{n_devices} devices wanted, {have} visible"}))
        return 0
    rows = FrameRows(_fortran_container(N_SNR, n_frames, frame_size), N_SNR, n_frames)
    fan = DeviceFanOut(frame_size, devices)
    try:
        fan(rows)                                          # warm: contexts, pinned slots, staging threads, kernels
        t0 = time.perf_counter()
        out = fan(rows)
        wall = time.perf_counter() - t0
        assert np.isfinite(out[:: max(1, out.shape[0] // 64)]).all()
        F = rows.shape[0]
        rec = {"GBps": F * frame_size * 16 / wall / 1e9, "frames_per_s": F / wall, "seconds": wall,
               "devices": devices, "staging_threads": [e.threads for e in fan.engines],
               "per_device_seconds": [e.stats.get("seconds") for e in fan.engines],
               "per_device_frames": fan.stats.get("frames_per_device"),
               "pcie_GBps": fan.stats.get("bytes_uploaded", 0) / wall / 1e9,
               # placement, one entry per engine: PCI bus id, NUMA node, CPUs bound (of those this process may use)
               "bus": [pl["pci_bus_id"] for pl in fan.placement()], "numa": [pl["numa_node"] for pl in fan.placement()],
               "cpus": [pl["n_cpus_allowed"] for pl in fan.placement()],
               "what": f"1 process, ({N_SNR},{n_frames},{frame_size}) c128 F-order, {F * frame_size * 16 / 1e9:.2f} GB"}
    finally:
        fan.close()
    print(json.dumps(_rounded(rec)), flush=True)
    return 0


def run_fanout_leg(n_devices: int, share_gpu: bool, frame_size: int, n_frames: int, time_limit: float = 300.0):
    """Start the fan-out child from a process that has not touched the GPU and return its JSON object (or an
    {"error": ...} -- never fatal to the headline)."""
    import subprocess
    cmd = [sys.executable, str(Path(__file__).resolve()), "--fanout-child", str(n_devices), "--frame-size", str(frame_size),
           "--frames", str(n_frames)] + (["--share-gpu"] if share_gpu else [])
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "AMCX_BENCH_SELF_LAUNCHED"):
        env.pop(k, None)                                  # the child is nobody's rank
    try:
        r = subprocess.run(cmd, env=env, stdin=subprocess.DEVNULL, capture_output=True, text=True, timeout=time_limit)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": f"fan-out child exited {r.returncode}: {r.stderr.strip()[-300:]}"}
        return json.loads(lines[-1])
    except Exception as exc:
        return {"error": repr(exc)}


def _valu_note(frames_per_s: float, ident=None, directory=None):
    """Secondary bounds of the N = 2048 kernel from the newest committed budget TAKEN ON THIS BINARY
    (profiles/r*_wave_budget.json: instructions per frame, in-kernel clock, issue slots -- DESIGN.md 4.6); None when the
    running kernel is not the one the budget was measured on."""
    directory = Path(directory) if directory else REPO / "profiles"
    for p in sorted(directory.glob("r*_wave_budget.json"), key=lambda q: q.name, reverse=True):
        try:
            b = json.loads(p.read_text())
        except Exception:
            continue
        if not _same_binary(b, ident):
            continue
        per_frame = b.get("valu_instr_per_frame")
        out = {"bound": b.get("bound", "board power, then VALU issue"), "valu_instr_per_frame": per_frame,
               "source": f"profiles/{p.name} (committed, same kernel digest; replayed)"}
        if per_frame:
            out["achieved_Gwaveinstr_per_s"] = per_frame * frames_per_s / 1e9
        for k in ("in_kernel_clock_GHz", "in_kernel_clock_zeros_GHz", "simd_cycles_per_frame", "valu_issue_slot_use"):
            if k in b:
                out[k] = b[k]
        return out
    return None


def _ensure_library(local_rank: int):
    """libamcx.so is a build artefact (not in git): if this checkout lacks it, local rank 0
    builds it (as __graft_entry__.build() does) and the other ranks wait for the file."""
    from amcpy_amd.csrc import build as b
    if local_rank == 0:
        if Path(b.HIPCC).exists():
            b.build(force=False, verbose=False)          # rebuilds only if sources are newer (build.stale)
        elif not b.LIB.exists():
            raise SystemExit("libamcx.so is missing and hipcc is not available")
        b.LIB.with_suffix(".ready").touch()
        return
    # other ranks: wait until local rank 0 has finished (a stale library must not be loaded while it is
    # being replaced): rank 0 touches the stamp file when its build() has returned
    deadline = time.time() + 300
    stamp = b.LIB.with_suffix(".ready")
    while time.time() < deadline:
        if b.LIB.exists() and stamp.exists() and stamp.stat().st_mtime >= b.LIB.stat().st_mtime and not b.stale():
            return
        time.sleep(0.5)
    raise SystemExit("libamcx.so was not (re)built by local rank 0 within 300 s")


def _config_label(frame_size, n_frames, n_mods, world):
    """Which BASELINE.json config the chosen per-GPU shape is (the default run is configs[1])."""
    if n_mods == N_MODS and n_frames == N_FRAMES and frame_size == FRAME_SIZE:
        return "BASELINE configs[1]"
    if n_mods == N_MODS and n_frames == N_FRAMES and frame_size == 4096:
        return "BASELINE configs[2]"
    if n_mods == N_MODS and n_frames * world == 65536 and frame_size == FRAME_SIZE:
        return f"BASELINE configs[3], frames sharded x{world}"
    if n_mods * world == 24 and n_frames == N_FRAMES and frame_size == 1024:
        return f"BASELINE configs[4], modulations sharded x{world}"
    return "not a BASELINE config"


def _free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


_PARENT_PID = None


def _die_with_parent():
    """preexec of a rank (its own session comes from start_new_session, so that the parent can take the whole rank
    down, helper processes included): SIGKILL when the parent goes away, however that happens -- including between
    the fork and the prctl (then the parent pid seen here is no longer the one that forked)."""
    import ctypes
    import signal
    try:
        ctypes.CDLL(None, use_errno=True).prctl(1, signal.SIGKILL, 0, 0, 0)      # PR_SET_PDEATHSIG
    except Exception:
        pass
    if _PARENT_PID is not None and os.getppid() != _PARENT_PID:
        os._exit(1)


def self_launch(n: int, argv, time_limit: float, script=None, build: bool = True) -> int:
    """`bench.py --gpus N` started by hand: N fresh child ranks of this script (the reference's caller runs ONE
    command and the parallelism happens inside, feature_extraction.py:89-97).  This process never initialises HIP or
    torch.cuda -- it builds the library if it is stale (hipcc only), starts the children with the launcher variables
    set, lets rank 0 write the one JSON line straight to this process's stdout (the other ranks' stdout goes to
    stderr), and returns the first non-zero exit code, killing whatever is still running ten seconds after a rank
    failed, at `time_limit`, or on SIGTERM / SIGINT / SIGHUP."""
    import signal
    import subprocess
    if build:
        _ensure_library(0)                                 # before the ranks exist: none of them waits for a build
    env = dict(os.environ)
    env.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               AMCX_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    script = str(Path(__file__).resolve() if script is None else script)      # (another script: the process-handling tests)
    procs = []
    global _PARENT_PID
    _PARENT_PID = os.getpid()

    def kill_all(sig=signal.SIGKILL):
        for pr in procs:
            if pr.poll() is None:
                try:
                    os.killpg(pr.pid, sig)
                except (ProcessLookupError, PermissionError):
                    pass

    def reap(timeout=5.0):
        for pr in procs:                                   # no zombies, and what a killed rank still wrote is flushed
            try:
                pr.wait(timeout=timeout)
            except Exception:
                pass

    def on_signal(signum, _frame):
        kill_all()
        reap()
        raise SystemExit(128 + signum)

    for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sg, on_signal)
    sys.stdout.flush()
    try:
        for r in range(n):
            e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen([sys.executable, script, *argv], env=e, stdin=subprocess.DEVNULL,
                                          stdout=None if r == 0 else sys.stderr, start_new_session=True,
                                          preexec_fn=_die_with_parent))
        deadline = time.time() + time_limit
        first_bad, bad_at = 0, None
        while any(pr.poll() is None for pr in procs):
            for r, pr in enumerate(procs):
                rc = pr.poll()
                if rc not in (None, 0) and first_bad == 0:
                    first_bad, bad_at = rc, time.time()
                    print(f"bench.py: rank {r} exited with {rc}; stopping the others", file=sys.stderr)
            if bad_at is not None and time.time() - bad_at > 10.0:
                kill_all()
            if time.time() > deadline:
                print(f"bench.py: ranks still running after {time_limit:.0f} s; killing them", file=sys.stderr)
                kill_all()
                first_bad = first_bad or 124
            time.sleep(0.05)
        for pr in procs:
            rc = pr.wait()
            if rc != 0 and first_bad == 0:
                first_bad = rc
        return first_bad if first_bad >= 0 else 128 - first_bad      # a rank killed by a signal
    finally:
        kill_all()
        reap()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=700,
                    help="untimed steps first; the default (~2.5 s of launches) lets the board's power cap settle "
                         "the clock before timing (profiles/r2_sustained.txt)")
    ap.add_argument("--frames", type=int, default=N_FRAMES, help="frames per (mod, SNR) block")
    ap.add_argument("--mods", type=int, default=N_MODS,
                    help="modulations per GPU (configs[4]: 24 over 8 GPUs = 3 with --frame-size 1024)")
    ap.add_argument("--variant", default="auto", choices=["auto", "block", "wave"])
    ap.add_argument("--frame-size", type=int, default=FRAME_SIZE,
                    help="samples per frame (the BASELINE metric is quoted at 2048; 1024/4096 are the other configs)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="label only: 'strong' when --frames / --mods were divided by the GPU count (tools/scale.sh)")
    ap.add_argument("--no-cpu-baseline", action="store_true",
                    help="skip the host baseline (use under rocprofv3: no worker processes)")
    ap.add_argument("--cpu-procs", type=int, default=None)
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="timed CPU work per worker after warm-up")
    ap.add_argument("--no-h2d", action="store_true", help="skip the real-data upload-path measurement")
    ap.add_argument("--no-d2h", action="store_true",
                    help="skip the result-to-host leg (under rocprofv3: its overlapped copies stretch the kernels' "
                         "durations in the trace, which should show the timed launches only)")
    ap.add_argument("--no-h2d-big", action="store_true", help="skip the 3.5 GB configs[1]-sized modulation of it")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for the timing barrier / MAX (gloo: rehearsals where RCCL cannot "
                         "run, e.g. two ranks sharing one GPU)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal: ranks take device local_rank %% device_count instead of one GPU each")
    ap.add_argument("--no-fma-probe", action="store_true", help="skip the ~1 s instruction-issue ceiling probe")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the configs[2] (N = 4096) and configs[4] (N = 1024) legs behind the timed region")
    ap.add_argument("--no-fanout", action="store_true", help="N > 1: skip the one-process fan-out upload leg")
    ap.add_argument("--fanout-child", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="self-launched ranks (--gpus N without a launcher) are killed after this many seconds")
    args = ap.parse_args()
    FS = args.frame_size
    if args.fanout_child > 0:
        raise SystemExit(fanout_child(args.fanout_child, args.share_gpu, FS, args.frames))
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ      # torch.distributed.run, or our own parent
    world = int(os.environ.get("WORLD_SIZE", "1")) if launched else 1
    rank = int(os.environ.get("RANK", "0")) if launched else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if launched else 0
    if launched and args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} under a launcher that started WORLD_SIZE={world} ranks: the two must agree "
                         f"(python bench.py --gpus N starts its own ranks when no launcher did)")
    if not launched and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:], args.launch_timeout))

    cpu = cpu_ref_shaped = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.cpu_procs, args.cpu_seconds)      # before the GPU is touched
        cpu_ref_shaped = cpu_baseline_reference_shaped()

    if os.environ.get("AMCX_BENCH_SELF_LAUNCHED") != "1":
        _ensure_library(local_rank)
    # N > 1: the one-process fan-out over all N devices, by a fresh child of rank 0 while no rank has touched its GPU
    # yet (the others wait in the rendezvous below)
    h2d_fanout = None
    if world > 1 and rank == 0 and not args.no_fanout and not args.no_h2d:
        h2d_fanout = run_fanout_leg(world, args.share_gpu, FS, min(args.frames, N_FRAMES))
    import torch
    import torch.distributed as dist
    from amcpy_amd import _lib, synth
    from amcpy_amd.features import features18

    dev_index = local_rank % max(1, torch.cuda.device_count()) if args.share_gpu else local_rank
    if dev_index >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU of its own ({torch.cuda.device_count()} visible); "
                         "--share-gpu rehearses several ranks on one device")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = launched
    # stdout carries exactly one JSON line: RCCL prints a version banner to fd 1 when the
    # communicator comes up, so everything until the final print goes to stderr instead
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rccl_ranks = 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        # the world size as the first collective sees it (every rank contributes 1)
        one = torch.ones(1, dtype=torch.int32, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(one)
        rccl_ranks = int(one.item())

    # ---- resident shard: (mods, snr, frames, N) complex64 arena in HBM --------
    n_mods = args.mods
    arena = torch.empty((n_mods, N_SNR, args.frames, FS), dtype=torch.complex64, device=dev)
    for mi in range(n_mods):
        synth.device_frames(synth.MODS6[mi % 6], N_SNR, args.frames, FS, device=dev, rank=rank,
                            mod_idx=mi, out=arena[mi])
    frames_per_launch = n_mods * N_SNR * args.frames
    out = torch.empty((n_mods, N_SNR, args.frames, 18), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    def step():
        features18(arena, out=out, variant=args.variant)   # one launch over the whole shard

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:                       # events sit on the stream the kernel is launched on
        a.record()
        step()
        b.record()
    fence()
    wall = wall_own = time.perf_counter() - t0
    launch_ms = [a.elapsed_time(b) for a, b in ev]
    if use_dist:
        t = torch.tensor([wall], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    # the same step with the (F x 18) result brought to the host every step: a ring of three result buffers, the
    # copy of step k on a second stream while steps k+1, k+2 compute (46 MB at ~28 GB/s is 1.7 ms against a 3.4 ms
    # launch; the runtime copies with a blit kernel, which the dispatcher may hold back behind the persistent
    # feature kernel -- with two buffers that delay stalled the next launch in some runs, with three it has a
    # whole launch of slack)
    RING = 3
    outs = [out] + [torch.empty_like(out) for _ in range(RING - 1)]
    hosts = [torch.empty(out.shape, dtype=torch.float32, pin_memory=True) for _ in range(RING)]
    main_stream, copy_stream = torch.cuda.current_stream(), torch.cuda.Stream(device=dev)
    computed = [torch.cuda.Event() for _ in range(RING)]
    copied = [torch.cuda.Event() for _ in range(RING)]

    def step_with_d2h(k):
        b = k % RING
        main_stream.wait_event(copied[b])                 # the copy that last read this buffer is done
        features18(arena, out=outs[b], variant=args.variant)
        computed[b].record(main_stream)
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(computed[b])
            hosts[b].copy_(outs[b], non_blocking=True)
            copied[b].record(copy_stream)

    wall_d2h_ms = None
    if not args.no_d2h:
        # The first overlapped steps after an idle queue are a transient: the first copy into a pinned buffer holds the
        # host for ~8 ms, and the ~28 launches after such a gap run up to 60 % long next to the copies before the
        # overlap settles at no cost at all (tools/d2h_overlap_probe.py, profiles/r3_d2h_overlap_probe.txt).  The
        # steady state of the pipelined loop is what is reported: warm steps and timed steps run back to back
        # without a synchronisation in between, bracketed by events (first timed launch .. last copy done).
        n_warm = 16 * RING
        e_first, e_last = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for k in range(n_warm):
            step_with_d2h(k)
        e_first.record(main_stream)
        for k in range(n_warm, n_warm + args.steps):
            step_with_d2h(k)
        e_last.record(copy_stream)
        torch.cuda.synchronize()
        wall_d2h_ms = e_first.elapsed_time(e_last) / args.steps
        assert torch.equal(hosts[(n_warm + args.steps - 1) % RING].view(torch.int32), out.cpu().view(torch.int32)), \
            "double-buffered D2H delivered a different result"
    del outs, hosts

    # sanity: the timed output is finite
    assert torch.isfinite(out[0, N_SNR // 2, :64]).all(), "non-finite features in the timed output"

    # measured streaming-read ceiling next to the nominal peak (rank 0)
    read_peak = None
    if rank == 0:
        lib = _lib.load()
        part = torch.empty(4096, dtype=torch.float32, device=dev)
        nbytes = arena.numel() * 8
        stream = torch.cuda.current_stream().cuda_stream
        for _ in range(2):
            _lib.check(lib.amcx_probe_read_bw(arena.data_ptr(), nbytes, part.data_ptr(), stream))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            _lib.check(lib.amcx_probe_read_bw(arena.data_ptr(), nbytes, part.data_ptr(), stream))
        e1.record()
        torch.cuda.synchronize()
        read_peak = nbytes * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9

    # the bound that binds: the board's instruction-issue rate under its power cap, measured now (every rank: a slow
    # device shows here), against the kernel's own rate
    fma = None
    if not args.no_fma_probe:
        try:
            fma = _lib.probe_fma_rate(1.0, torch.cuda.current_stream().cuda_stream)
        except Exception as exc:
            fma = {"error": repr(exc)}

    h2d = None
    if rank == 0 and world == 1 and not args.no_h2d:
        del arena
        arena = None
        torch.cuda.empty_cache()
        try:
            h2d = h2d_path(dev, big=not args.no_h2d_big)
        except Exception as exc:                               # the upload-path leg never takes the headline down with it
            h2d = {"error": repr(exc)}

    # the other BASELINE frame sizes, driver-timed in the same run (N = 1, the default shape only): configs[2] whole
    # (6 x 26 x 4096 frames x 4096 samples, 20.9 GB) and one GPU's eighth of configs[4] (3 of 24 modulations x 26 x 4096 x
    # 1024 samples, 2.6 GB) -- 5 warm + 20 timed launches each, events on the launch stream
    other = None
    # (AFTER the upload-path leg: freeing the 21 GB configs[2] arena right before it left the driver clearing VRAM on the copy
    #  engines the uploads use -- 83 instead of 105 GB/s for the configs[1]-sized modulation, 17 instead of 57 for configs[0])
    if rank == 0 and world == 1 and not args.no_other_configs and FS == FRAME_SIZE and args.frames == N_FRAMES and n_mods == N_MODS:
        arena = None
        torch.cuda.empty_cache()
        other = {}
        for label, fs2, mods2 in (("configs[2]", 4096, 6), ("configs[4]/8", 1024, 3)):
            try:
                other[str(fs2)] = _timed_config(torch, synth, features18, dev, rank, fs2, mods2, N_FRAMES, label, _lib)
            except Exception as exc:                           # never fatal to the headline
                other[str(fs2)] = {"error": repr(exc)[:200]}
            torch.cuda.empty_cache()

    # N > 1: the one cross-rank step of the real path -- rank 0 collecting every rank's (F x 18) block
    # (amcpy_amd/sharding.py: one padded float32 tensor gather; device tensors over RCCL) -- timed once, outside
    # the timed region, so that a multi-GPU run also shows what the gather costs.  Never fatal to the headline, and
    # never a hang: every rank reports whether it can take part BEFORE the collective (as run_extraction does), and
    # all of them skip it if one cannot.
    gather = None
    if use_dist and world > 1:
        from amcpy_amd.sharding import gather_rows
        local, problem = None, None
        try:
            local = out.reshape(-1, 18).cpu().numpy()
        except Exception as exc:
            problem = f"rank {rank}: {exc!r}"
        problems = [None] * world
        dist.all_gather_object(problems, problem)
        problems = [q for q in problems if q]
        if problems:
            gather = {"error": "; ".join(problems)}
        else:
            dist.barrier()
            t_g = time.perf_counter()
            full = gather_rows(local, local.shape[0] * world, rank, world)      # a failure here is a bug: let it propagate
            dist.barrier()
            gather = {"ms": (time.perf_counter() - t_g) * 1e3, "bytes_per_rank": int(local.nbytes),
                      "rows_on_rank0": None if full is None else int(full.shape[0]),
                      "what": "gather_rows of one step's result"}
    per_rank = None
    if use_dist:
        try:
            bus = _lib.device_pci_bus_id(dev_index)
        except Exception:
            bus = None
        # compact: eight of these and the fan-out block must leave the line under 4 KB (the driver keeps its tail only)
        mine = {"rank": rank, "dev": dev_index, "bus": bus,
                "ms": [sum(launch_ms) / len(launch_ms), min(launch_ms), max(launch_ms)],          # launch mean / min / max
                "frames": frames_per_launch, "wall_s": wall_own,
                "fma_G": None if not fma or "error" in fma else fma["wave_instr_per_s"] / 1e9,
                "fma_GHz": None if not fma or "error" in fma else fma["clock_GHz"]}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return

    total_frames = frames_per_launch * world * args.steps
    value = total_frames / wall
    mean_launch_s = sum(launch_ms) / len(launch_ms) * 1e-3
    srt = sorted(launch_ms)
    alg_bytes = (8 * FS + 72) * frames_per_launch
    kname = _lib.kernel_name(FS, _lib.VARIANTS[args.variant])
    ident = _binary_identity(kname)
    traffic, traffic_src = _pmc_traffic(frames_per_launch, FS, ident)
    achieved = alg_bytes / mean_launch_s / 1e9
    short = lambda h: h[:16] if isinstance(h, str) else h     # noqa: E731  (64 bits of a SHA-256 name a build)
    # about 3.5 KB: the driver keeps only the tail of a long line, and round 3's 7 KB `parity` block pushed h2d /
    # wall_incl_d2h_ms out of its record
    rec = {
        "metric": f"IQ frames/sec (18 features, {FS}-sample complex64)",
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"{n_mods} mods x {N_SNR} SNR x {args.frames} frames x {FS} samples "
                        f"complex64 per GPU ({_config_label(FS, args.frames, n_mods, world)}), resident in HBM; "
                        f"one launch per step",
            "frames_per_gpu_per_step": frames_per_launch, "frame_size": FS,
            "kernel": kname,
            "sharding": f"frames x{world}, no collective on the data path",
        },
        # the GPU program that was timed (tools/codeobj_gate.py --print; amcpy_amd/csrc/codeobj.json holds the tree's)
        "binary": ({"code_object_sha256": short(ident.get("code_object_sha256")), "kernel_sha256": short(ident.get("kernel_sha256"))}
                   if "error" not in ident else ident),
        "rccl_ranks": rccl_ranks,
        "launcher": ("none" if not launched else
                     "self" if os.environ.get("AMCX_BENCH_SELF_LAUNCHED") == "1" else "external") + f"/{args.dist_backend}",
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": alg_bytes, "mean_launch_ms": mean_launch_s * 1e3,
            # the spread tells a slow box (all three shift) from a disturbed run (max far from median)
            "launch_ms_min": srt[0], "launch_ms_median": srt[len(srt) // 2], "launch_ms_max": srt[-1],
            "frac_at_min": alg_bytes / (srt[0] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "measured_read_peak_GBps": read_peak,
            "frac_of_measured_read_peak": None if not read_peak else achieved / read_peak,
            # flat, so that a reader that keeps only this object's scalars still has the three BASELINE frame sizes
            **({} if not other else {f"frac_n{k}": v.get("frac") for k, v in other.items()}),
            "secondary": _secondary(value / world, fma, brief=world > 1, ident=ident) if FS == FRAME_SIZE else None,
        },
        **({} if other is None else {"other_configs": other}),
        "wall_incl_d2h_ms": wall_d2h_ms,
        "h2d": h2d,
        "gather": gather,
        **({} if per_rank is None else _scaling_block(per_rank, value, world)),
        **({} if h2d_fanout is None else {"h2d_fanout": h2d_fanout}),
        "cpu_baseline": cpu,
        "cpu_baseline_reference_shaped": cpu_ref_shaped,
        "parity": _parity_block(),
    }
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    print(json.dumps(_rounded(rec)), flush=True)


def _timed_config(torch, synth, features18, dev, rank, frame_size, n_mods, n_frames, label, _lib, warm=5, steps=20):
    """One more BASELINE shape, resident in HBM, `warm` untimed + `steps` timed launches bracketed by events on the launch
    stream: frames/s and the fraction of the HBM roofline, the way the headline's roofline.achieved is formed."""
    arena = torch.empty((n_mods, N_SNR, n_frames, frame_size), dtype=torch.complex64, device=dev)
    for mi in range(n_mods):
        synth.device_frames(synth.MODS6[mi % 6], N_SNR, n_frames, frame_size, device=dev, rank=rank, mod_idx=mi, out=arena[mi])
    out = torch.empty((n_mods, N_SNR, n_frames, 18), dtype=torch.float32, device=dev)
    for _ in range(warm):
        features18(arena, out=out)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in ev:
        a.record()
        features18(arena, out=out)
        b.record()
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in ev]
    assert torch.isfinite(out[0, N_SNR // 2, :64]).all()
    frames = n_mods * N_SNR * n_frames
    mean_s = sum(ms) / len(ms) * 1e-3
    gbps = (8 * frame_size + 72) * frames / mean_s / 1e9
    return {"config": label, "frames_per_s": frames / mean_s, "frac": gbps / HBM_PEAK_GBPS, "achieved_GBps": gbps,
            "mean_launch_ms": mean_s * 1e3, "frames_per_launch": frames, "kernel": _lib.kernel_name(frame_size),
            "warm": warm, "steps": steps}


def _secondary(frames_per_s_per_gpu: float, fma, brief: bool = False, ident=None, directory=None):
    """What THIS run measured -- the device's FMA ceiling under the power cap (amcx_probe_fma_rate: 4 waves / SIMD of
    independent v_fma_f32, 0.5 s to settle + 0.5 s timed) -- and, when a committed budget of the running kernel exists
    (same machine-code digest), the kernel's own instruction rate against it."""
    out = _valu_note(frames_per_s_per_gpu, ident, directory)
    if out is not None and brief:                  # a multi-rank line carries per_rank and h2d_fanout instead of the replayed budget
        out = {k: out[k] for k in ("valu_instr_per_frame", "source") if k in out}
    if out is None:
        out = {"replayed": None, "why": "no committed budget of this kernel's machine code"}
    if fma is None:
        return out
    if "error" in fma:
        out["measured"] = fma
        return out
    ceil = fma["wave_instr_per_s"] / 1e9
    meas = {"fma_Gwaveinstr_per_s": ceil, "fma_clock_GHz": fma["clock_GHz"]}
    if out.get("valu_instr_per_frame"):
        kern = out["valu_instr_per_frame"] * frames_per_s_per_gpu / 1e9
        meas.update(kernel_Gwaveinstr_per_s=kern, ratio=kern / ceil if ceil else None)
    out["measured"] = meas
    return out


def _scaling_block(per_rank, value, world):
    """Which rank was slow?  Every rank's own kernel-only rate (frames per launch / mean launch time); the aggregate
    against N x the best of them; the spread."""
    def short(v):
        if isinstance(v, float):
            return float(f"{v:.5g}")
        if isinstance(v, list):
            return [short(x) for x in v]
        return v
    rates = [r["frames"] / (r["ms"][0] * 1e-3) for r in per_rank if r and r.get("ms") and r["ms"][0]]
    out = {"per_rank": [{k: short(v) for k, v in r.items()} for r in per_rank]}
    if rates:
        out["scaling_efficiency"] = value / (world * max(rates))
        out["rank_balance"] = min(rates) / max(rates)
        out["sum_of_rank_rates"] = sum(rates)
    return out


def _rounded(x):
    """Six significant digits are more than any of these measurements has: a shorter line."""
    if isinstance(x, float):
        return float(f"{x:.6g}")
    if isinstance(x, dict):
        return {k: _rounded(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_rounded(v) for v in x]
    return x


if __name__ == "__main__":
    main()
