"""Batch feature extraction: the MI355X drop-in for the reference's
``run_extraction(cfg)`` (src/amcpy/feature_extraction.py:85-99).

Same call, same files: reads ``cfg.paths.mat_data / cfg.paths.mat_filename``
(one variable per modulation, ``cfg.signals.mat_info[mod]``, shaped
``(n_snr, n_frames, >= frame_size)`` complex; feature_extraction.py:46-48) and
writes ``cfg.paths.calculated_features / f"{mod}_features.mat"`` holding exactly
``"Modulation"`` and ``mat_info[mod]`` -> float32 ``(n_snr, n_frames, 18)``
(feature_extraction.py:56,77-81), so preprocessing / plotting / training code
that loads those files is untouched.

What is different underneath:
* every variable of the container is decoded ONCE, by one process, one
  modulation at a time (the reference decodes the whole file in each of its six
  child processes, feature_extraction.py:46-47);
* there are no worker processes or threads on the compute side: a modulation's
  frames go to the GPU in chunks through two pinned staging buffers, the gather
  of chunk k+1 (host threads) and its upload (copy stream) overlapping the
  kernel on chunk k.  ``scipy.io.loadmat`` hands back Fortran-ordered arrays, so
  the ``[0:frame_size]`` slice of a frame (feature_extraction.py:68) is gathered
  straight into the staging buffer -- no full transposed copy of the container
  is ever made.  ``cfg.signals.num_threads`` is the number of gather threads;
* with several ranks (one process per GPU of one node, ``torch.distributed``
  initialised by the caller) rank 0 alone decodes the container and publishes
  each modulation's packed frames as a memory-mapped file in shared memory;
  every rank uploads its own contiguous frame range from it over its own PCIe
  link, and rank 0 gathers the (F x 18) rows and writes the files
  (amcpy_amd/sharding.py).  No collective touches the IQ data;
* a failure raises: the reference's worker threads swallow exceptions and leave
  zero rows behind (feature_extraction.py:33-39).
"""
from __future__ import annotations

import os
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Callable, Optional

import numpy as np

from .config import Config
from .sharding import gather_rows, shard_range, sharded_features


def _rank_world():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except Exception:
        pass
    return 0, 1


# ----------------------------------------------------------------------------
# frame sources: rows of the C-order flattening (snr-major) of a container array
# ----------------------------------------------------------------------------
class FrameRows:
    """Frames ``[lo, hi)`` of ``parsed[:n_snr, :n_frames]`` flattened snr-major
    (frame g = snr * n_frames + k, the order feature_extraction.py:64-72 enqueues
    them in), WITHOUT materialising the flattening: for the Fortran-ordered arrays
    ``loadmat`` returns, ``reshape`` would be a full transposing copy."""

    def __init__(self, parsed: np.ndarray, n_snr: int, n_frames: int, lo: int = 0, hi: Optional[int] = None):
        self.parsed, self.n_snr, self.n_frames = parsed, n_snr, n_frames
        self.lo = lo
        self.hi = n_snr * n_frames if hi is None else hi
        self.dtype = parsed.dtype

    @property
    def shape(self):
        return (self.hi - self.lo, self.parsed.shape[2])

    def slice(self, lo: int, hi: int) -> "FrameRows":
        return FrameRows(self.parsed, self.n_snr, self.n_frames, self.lo + lo, self.lo + hi)

    def gather(self, dst: np.ndarray, g0: int, g1: int, n: int) -> None:
        """dst[(g1-g0), n] <- the first n samples of frames [g0, g1) of this range."""
        a, b = self.lo + g0, self.lo + g1
        row = 0
        while a < b:
            s, k = divmod(a, self.n_frames)
            take = min(b - a, self.n_frames - k)
            np.copyto(dst[row:row + take], self.parsed[s, k:k + take, :n], casting="same_kind")
            row += take
            a += take

    def to_array(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        self.gather(out, 0, self.shape[0], self.shape[1])
        return out


class _ArrayRows:
    """The same interface over a plain (F, L) array or memmap."""

    def __init__(self, arr: np.ndarray):
        self.arr, self.dtype, self.shape = arr, arr.dtype, arr.shape

    def gather(self, dst, g0, g1, n):
        np.copyto(dst, self.arr[g0:g1, :n], casting="same_kind")


_POOL: Optional[ThreadPoolExecutor] = None
_POOL_SIZE = 0


def _gather_parallel(rows, dst: np.ndarray, g0: int, g1: int, n: int, threads: int) -> None:
    """rows.gather split over host threads (numpy releases the GIL inside copyto)."""
    global _POOL, _POOL_SIZE
    count = g1 - g0
    threads = max(1, min(threads, count // 64 if count >= 128 else 1))
    if threads == 1:
        rows.gather(dst, g0, g1, n)
        return
    if _POOL is None or _POOL_SIZE < threads:
        if _POOL is not None:
            _POOL.shutdown(wait=True)
        _POOL, _POOL_SIZE = ThreadPoolExecutor(max_workers=threads, thread_name_prefix="amcx-gather"), threads
    per = -(-count // threads)
    futs = [_POOL.submit(rows.gather, dst[a:min(count, a + per)], g0 + a, g0 + min(count, a + per), n)
            for a in range(0, count, per)]
    for f in futs:
        f.result()


# ----------------------------------------------------------------------------
# host frames -> HBM -> features: the double-buffered upload pipeline
# ----------------------------------------------------------------------------
class HipEngine:
    """``engine(frames) -> (F, 18) float32`` through device memory.

    ``frames`` is an (F, L) complex array / memmap or a :class:`FrameRows`.  Two slots,
    each one pinned host buffer + one device buffer, are reused for every chunk and every
    call: while the kernel runs on slot k the host threads gather chunk k+1 into the other
    slot's pinned buffer and the copy stream uploads it.  Ordering is by events, not by
    allocator stream tracking: a slot's pinned buffer is rewritten only after its last upload
    has finished, its device buffer only after the kernel that read it has.  A complex128
    container (MATLAB doubles) is uploaded as is and rounded to complex64 on the GPU
    (PCIe moves 16 B/sample faster than a host ``astype`` produces 8 B/sample).
    ``stats`` of the last call: bytes uploaded, seconds, frames."""

    def __init__(self, frame_size: int, device: Optional[int] = None, chunk_bytes: int = 256 << 20,
                 threads: Optional[int] = None):
        import torch
        self.N = int(frame_size)
        self.dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.chunk_bytes = int(chunk_bytes)
        self.threads = max(1, min(threads or 8, os.cpu_count() or 1))
        self._slots = {}            # torch dtype -> list of slot dicts
        self._copy_stream = None
        self.stats = {}

    def _ring(self, tdtype, rows: int):
        import torch
        ring = self._slots.get(tdtype)
        if ring is None or ring[0]["host"].shape[0] < rows:
            ring = []
            for _ in range(2):
                slot = {"host": torch.empty((rows, self.N), dtype=tdtype, pin_memory=True),
                        "dev": torch.empty((rows, self.N), dtype=tdtype, device=self.dev),
                        "uploaded": torch.cuda.Event(), "consumed": torch.cuda.Event()}
                if tdtype != torch.complex64:
                    slot["c64"] = torch.empty((rows, self.N), dtype=torch.complex64, device=self.dev)
                ring.append(slot)
            self._slots[tdtype] = ring
        return ring

    def __call__(self, frames) -> np.ndarray:
        import torch
        from .features import features18

        rows = frames if hasattr(frames, "gather") else _ArrayRows(np.asarray(frames))
        F, L = rows.shape
        if L < self.N:
            raise ValueError(f"rows of {L} samples are shorter than frame_size {self.N}")
        if F == 0:
            return np.empty((0, 18), dtype=np.float32)
        if rows.dtype == np.complex64:
            tdtype = torch.complex64
        else:                       # doubles, or real / integer samples: staged as complex128
            tdtype = torch.complex128
        itemsize = 8 if tdtype == torch.complex64 else 16
        per = max(1, min(F, self.chunk_bytes // (self.N * itemsize)))
        with torch.cuda.device(self.dev):
            ring = self._ring(tdtype, per)
            if self._copy_stream is None:
                self._copy_stream = torch.cuda.Stream(device=self.dev)
            copy_stream, main = self._copy_stream, torch.cuda.current_stream(self.dev)
            out = torch.empty((F, 18), dtype=torch.float32, device=self.dev)
            t0 = time.perf_counter()
            for k, f0 in enumerate(range(0, F, per)):
                f1 = min(F, f0 + per)
                n = f1 - f0
                slot = ring[k & 1]
                slot["uploaded"].synchronize()             # its pinned buffer is free again
                _gather_parallel(rows, slot["host"][:n].numpy(), f0, f1, self.N, self.threads)
                with torch.cuda.stream(copy_stream):
                    copy_stream.wait_event(slot["consumed"])   # the kernel that read the device buffer is done
                    slot["dev"][:n].copy_(slot["host"][:n], non_blocking=True)
                    slot["uploaded"].record(copy_stream)
                main.wait_event(slot["uploaded"])
                x = slot["dev"][:n]
                if tdtype != torch.complex64:
                    slot["c64"][:n].copy_(x)               # round-to-nearest-even, on the GPU
                    x = slot["c64"][:n]
                features18(x, out=out[f0:f1])
                slot["consumed"].record(main)
            host_out = out.cpu().numpy()                    # synchronises the main stream
            self.stats = {"frames": F, "seconds": time.perf_counter() - t0,
                          "bytes_uploaded": F * self.N * itemsize, "chunks": -(-F // per),
                          "gather_threads": self.threads}
        return host_out


def _hip_compute(frame_size: int, device: Optional[int], chunk_bytes: int = 256 << 20,
                 threads: Optional[int] = None) -> Callable:
    """The production engine (kept under its round-1 name for callers that pass it on)."""
    return HipEngine(frame_size, device, chunk_bytes, threads)


def _check_container(parsed: np.ndarray, cfg: Config):
    n_snr = len(cfg.signals.snr_values)
    n_frames = cfg.signals.num_frames
    N = cfg.signals.frame_size
    if len(cfg.features.all_features) != 18:
        raise ValueError("the extraction engine always produces the 18 features of FeatureConfig.all_features")
    if parsed.ndim != 3 or parsed.shape[0] < n_snr or parsed.shape[1] < n_frames or parsed.shape[2] < N:
        raise ValueError(f"container array has shape {parsed.shape}, config needs "
                         f"(>={n_snr}, >={n_frames}, >={N})")
    return n_snr, n_frames, N


def extract_modulation(parsed: np.ndarray, cfg: Config, *, compute=None, device: Optional[int] = None,
                       group=None) -> Optional[np.ndarray]:
    """All 18 features of one modulation's ``(n_snr, n_frames, L)`` array, which every rank
    holds (``run_extraction`` itself decodes on rank 0 only).  Returns float32
    ``(n_snr, n_frames, 18)`` on rank 0 (None on other ranks)."""
    n_snr, n_frames, N = _check_container(parsed, cfg)
    rank, world = _rank_world()
    rows = FrameRows(parsed, n_snr, n_frames)
    if compute is None:
        engine = HipEngine(N, device, threads=cfg.signals.num_threads)
        lo, hi = shard_range(n_snr * n_frames, rank, world)
        local = engine(rows.slice(lo, hi)) if hi > lo else np.empty((0, 18), dtype=np.float32)
        mat = gather_rows(local, n_snr * n_frames, rank, world, group)
    else:                           # injected engine (tests): plain (F, L) arrays
        mat = sharded_features(rows.to_array(), N, compute, rank, world, group)
    return None if mat is None else mat.reshape(n_snr, n_frames, 18)


def extract_raw_stream(path, frame_size: int, *, skip_samples: int = 0, max_frames: Optional[int] = None,
                       compute=None, device: Optional[int] = None) -> np.ndarray:
    """Features of a raw complex64 sample stream on disk (GNU Radio file sink: interleaved
    float32 I/Q, no header -- what the reference's legacy reader takes with
    ``np.fromfile(..., dtype=np.complex64)`` and a fixed number of leading samples dropped,
    old/read_binary_stream.py:28,48,54-56).  The file is memory-mapped and cut into
    consecutive ``frame_size``-sample frames (a trailing partial frame is dropped); frames go
    up chunk by chunk through the same pinned, overlapped path as ``run_extraction``, so
    the file never has to fit in host memory.  Returns ``(n_frames, 18)`` float32."""
    if frame_size < 2:
        raise ValueError("frame_size must be >= 2")
    if skip_samples < 0:
        raise ValueError("skip_samples must be >= 0")
    n_total = Path(path).stat().st_size // 8 - skip_samples
    n_frames = max(0, n_total // frame_size)
    if max_frames is not None:
        n_frames = min(n_frames, int(max_frames))
    if n_frames == 0:
        return np.empty((0, 18), dtype=np.float32)
    frames = np.memmap(path, dtype=np.complex64, mode="r", offset=8 * skip_samples,
                       shape=(n_frames, frame_size))
    fn = compute or HipEngine(frame_size, device)
    return np.asarray(fn(frames), dtype=np.float32)


class _PairRows:
    """Frame source over a dataset of float32 (I, Q) pairs shaped (F, L, 2) that supports slicing --
    a numpy array / memmap or an ``h5py.Dataset`` (RadioML-2018.01A's ``X``: 2 555 904 x 1024 x 2,
    reference old/dataset.py:50-56).  The pairs are bit-identical to complex64, so a chunk is read
    (h5py decodes it from the file) and re-viewed, never converted."""

    def __init__(self, dataset, lo: int = 0, hi: Optional[int] = None):
        shape = tuple(dataset.shape)
        if len(shape) != 3 or shape[2] != 2:
            raise ValueError(f"expected an (F, L, 2) dataset of (I, Q) pairs, got shape {shape}")
        if np.dtype(dataset.dtype) != np.float32:
            raise TypeError(f"(I, Q) pairs must be float32, got {dataset.dtype}")
        self.ds, self.lo = dataset, lo
        self.hi = shape[0] if hi is None else hi
        self.dtype = np.dtype(np.complex64)
        self.shape = (self.hi - self.lo, shape[1])

    def gather(self, dst, g0, g1, n):
        blk = np.ascontiguousarray(self.ds[self.lo + g0:self.lo + g1])          # (g, L, 2) float32
        np.copyto(dst, blk.view(np.complex64)[..., 0][:, :n])


def extract_iq_pairs(dataset, frame_size: Optional[int] = None, *, first_frame: int = 0,
                     max_frames: Optional[int] = None, compute=None, device: Optional[int] = None) -> np.ndarray:
    """Features of frames stored as float32 (I, Q) pairs, ``dataset[f, n] = (I, Q)`` -- RadioML's
    ``(F, 1024, 2)`` layout (reference old/dataset.py:50-56, old/dataset_analysis.py:22).  ``dataset``
    is anything sliceable with ``.shape`` and ``.dtype`` (numpy array, memmap, h5py.Dataset); frames go
    up chunk by chunk through the pinned, overlapped path, so the set never has to fit in host memory.
    Returns ``(n_frames, 18)`` float32."""
    F, L = int(dataset.shape[0]), int(dataset.shape[1])
    N = L if frame_size is None else int(frame_size)
    if N < 2 or N > L:
        raise ValueError(f"frame_size {N} outside 2 .. {L}")
    lo = min(max(0, int(first_frame)), F)
    hi = F if max_frames is None else min(F, lo + int(max_frames))
    rows = _PairRows(dataset, lo, hi)
    if hi <= lo:
        return np.empty((0, 18), dtype=np.float32)
    if compute is not None:                       # injected engine (tests): plain (F, N) arrays
        block = np.empty((hi - lo, N), dtype=np.complex64)
        rows.gather(block, 0, hi - lo, N)
        return np.asarray(compute(block), dtype=np.float32)
    return HipEngine(N, device)(rows)


def extract_radioml_hdf5(path, *, key: str = "X", frame_size: Optional[int] = None, first_frame: int = 0,
                         max_frames: Optional[int] = None, device: Optional[int] = None) -> np.ndarray:
    """``extract_iq_pairs`` on dataset ``key`` of a RadioML-style HDF5 file (``GOLD_XYZ_OSC.0001_1024.hdf5``:
    ``X`` float32 (2 555 904, 1024, 2), reference old/dataset.py:43-56).  Needs ``h5py``, which the reference
    lists for its legacy scripts; raises ImportError with that hint where it is not installed."""
    try:
        import h5py
    except ImportError as exc:                    # not a silent fallback: say what is missing
        raise ImportError("extract_radioml_hdf5 needs h5py (pip install h5py); any sliceable (F, L, 2) float32 "
                          "dataset can be passed to extract_iq_pairs instead") from exc
    with h5py.File(str(path), "r") as fh:
        return extract_iq_pairs(fh[key], frame_size, first_frame=first_frame, max_frames=max_frames, device=device)


# ----------------------------------------------------------------------------
# run_extraction
# ----------------------------------------------------------------------------
def _shared_dir() -> Path:
    """Where rank 0 publishes packed frames for the other ranks of the node: /dev/shm
    (page cache, no disk) when it exists, the temp dir otherwise."""
    shm = Path("/dev/shm")
    return shm if shm.is_dir() and os.access(shm, os.W_OK) else Path(tempfile.gettempdir())


def _publish_packed(rows: FrameRows, N: int, threads: int) -> Path:
    """Rank 0: frames -> a packed (F, N) .npy in shared memory, source dtype kept."""
    F = rows.shape[0]
    fd, name = tempfile.mkstemp(prefix="amcx_frames_", suffix=".npy", dir=str(_shared_dir()))
    os.close(fd)
    dtype = rows.dtype if rows.dtype in (np.complex64, np.complex128) else np.dtype(np.complex128)
    mm = np.lib.format.open_memmap(name, mode="w+", dtype=dtype, shape=(F, N))
    step = max(1, (64 << 20) // (N * dtype.itemsize))
    for g0 in range(0, F, step):
        g1 = min(F, g0 + step)
        _gather_parallel(rows, mm[g0:g1], g0, g1, N, threads)
    mm.flush()
    del mm
    return Path(name)


def _load_variable(mat_path: Path, key: str) -> np.ndarray:
    import scipy.io
    data = scipy.io.loadmat(str(mat_path), variable_names=[key])
    if key not in data:
        raise KeyError(f"{mat_path} has no variable {key!r}")
    return np.asarray(data[key])


def run_extraction(cfg: Config, *, compute=None, device: Optional[int] = None, verbose: bool = True) -> None:
    """Drop-in for the reference's ``run_extraction(cfg)``: writes one
    ``{mod}_features.mat`` per entry of ``cfg.signals.modulations_with_noise``."""
    import scipy.io

    rank, world = _rank_world()
    cfg.paths.ensure_dirs()
    mat_path = cfg.paths.mat_data / cfg.paths.mat_filename
    N = cfg.signals.frame_size
    threads = max(1, int(cfg.signals.num_threads))
    engine = compute if compute is not None else HipEngine(N, device, threads=threads)
    for mod in cfg.signals.modulations_with_noise:
        t0 = time.perf_counter()
        key = cfg.signals.mat_info[mod]
        feats = None
        if world == 1:
            parsed = _load_variable(mat_path, key)          # one variable at a time: bounded host memory
            n_snr, n_frames, _ = _check_container(parsed, cfg)
            rows = FrameRows(parsed, n_snr, n_frames)
            mat = engine(rows) if compute is None else compute(rows.to_array())
            feats = np.asarray(mat, dtype=np.float32).reshape(n_snr, n_frames, 18)
            del parsed, rows
        else:
            import torch.distributed as dist
            meta = [None]
            if rank == 0:
                try:
                    parsed = _load_variable(mat_path, key)
                    n_snr, n_frames, _ = _check_container(parsed, cfg)
                    shared = _publish_packed(FrameRows(parsed, n_snr, n_frames), N, threads)
                    del parsed
                    meta = [("ok", str(shared), n_snr, n_frames)]
                except Exception as exc:                    # every rank must leave the collective
                    meta = [("error", repr(exc), 0, 0)]
            dist.broadcast_object_list(meta, src=0)
            status, shared, n_snr, n_frames = meta[0]
            if status != "ok":
                raise RuntimeError(f"rank 0 could not read {key!r} from {mat_path}: {shared}")
            try:
                packed = np.load(shared, mmap_mode="r")
                F = n_snr * n_frames
                lo, hi = shard_range(F, rank, world)
                local = (np.asarray(engine(packed[lo:hi]), dtype=np.float32) if hi > lo
                         else np.empty((0, 18), dtype=np.float32))
                del packed
                mat = gather_rows(local, F, rank, world)
            finally:
                dist.barrier()                              # everyone has unmapped the file
                if rank == 0:
                    Path(shared).unlink(missing_ok=True)
            if rank == 0:
                feats = mat.reshape(n_snr, n_frames, 18)
        if rank == 0:
            out_path = cfg.paths.calculated_features / f"{mod}_features.mat"
            scipy.io.savemat(str(out_path), {"Modulation": mod, key: feats})
            if verbose:
                print(f"[{mod}] {feats.shape[0] * feats.shape[1]} frames in "
                      f"{time.perf_counter() - t0:.2f}s -> {out_path}")
    if verbose and rank == 0:
        print("All feature calculations complete!")
