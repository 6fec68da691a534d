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
  child processes, feature_extraction.py:46-47), and the three stages overlap:
  a reader thread decodes modulation k+1 while modulation k is on the GPU and a
  writer thread saves k-1;
* there are no worker processes or threads on the compute side.  A modulation goes
  to the GPU AS IT LIES in host memory (``amcx_ctx_features18_strided_host``,
  include/amcx.h): ``scipy.io.loadmat`` hands back Fortran-ordered arrays, in which a
  frame's samples are ``n_snr * n_frames`` elements apart but a sample PLANE is
  contiguous, so planes are staged into pinned memory by a few host threads
  (``cfg.signals.num_threads``; doubles are rounded to float32 on the way, so PCIe
  carries 8 bytes per sample), uploaded while the next planes are staged, and
  transposed to frame-major by a device kernel.  No transposed copy of the
  container is ever made on the host and the ``[0:frame_size]`` slice of a frame
  (feature_extraction.py:68) is just the first ``frame_size`` planes;
* with several ranks (one process per GPU of one node, ``torch.distributed``
  initialised by the caller) a variable the fast reader can map (an
  uncompressed level-5 variable of doubles or singles) is mapped by every rank
  for itself -- the page cache is shared, nothing is decoded or copied; anything
  else is decoded by rank 0 alone, which publishes the part the configuration
  uses, in the order it lies in memory, as a memory-mapped file in shared
  memory.  Either way every rank uploads its own contiguous frame range over its
  own PCIe link, and rank 0 gathers the (F x 18) rows and writes the files
  (amcpy_amd/sharding.py).  No collective touches the IQ data;
* several GPUs need no launcher: ``run_extraction(cfg, devices=[0, 1, ...])`` (``python -m amcpy_amd extract
  --devices all``) drives one engine per device from one host thread each inside ONE process
  (:class:`DeviceFanOut`): every device takes its share of the frame axis, reads it from the file over its own
  staging threads and PCIe link, and writes its rows straight into the result -- no process group, no gather;
* a failure raises, on every rank: the reference's worker threads swallow
  exceptions and leave zero rows behind (feature_extraction.py:33-39), and its
  parent ignores the children's exit codes (:96-97).
"""
from __future__ import annotations

import functools
import os
import socket
import tempfile
import threading
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Callable, Iterator, List, Optional, Tuple

import numpy as np

from . import _lib
from .config import Config
from .sharding import collectives_forced, gather_frame_columns, gather_rows, shard_by_frames, shard_range, sharded_features


def _process_group_up() -> bool:
    try:
        import torch.distributed as dist
        return _lib.torch_wanted() and dist.is_available() and dist.is_initialized()
    except Exception:
        return False


def _rank_world():
    if not _lib.torch_wanted():          # a process that opted out of torch has no process group
        return 0, 1
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
class SplitComplex:
    """A complex ``(n_snr, n_frames, L)`` container held as two real arrays of equal shape, strides
    and dtype (float32 / float64) -- how a MATLAB v5 file stores a complex variable, so a
    memory-mapped .mat goes to the GPU without a complex array being built (amcpy_amd/matfile.py).
    ``imag`` may be None (a real signal).  Indexing returns an ordinary complex ndarray."""

    def __init__(self, real: np.ndarray, imag: Optional[np.ndarray]):
        if imag is not None and (imag.shape != real.shape or imag.strides != real.strides or imag.dtype != real.dtype):
            raise ValueError("real and imaginary parts must agree in shape, strides and dtype")
        if real.dtype not in (np.float32, np.float64):
            raise TypeError(f"split containers hold float32 or float64, got {real.dtype}")
        self.real, self.imag = real, imag
        self.source = None            # "mapped": views of a memory-mapped file every process can map for itself
        self.shape, self.ndim = real.shape, real.ndim
        self.dtype = np.dtype(np.complex64 if real.dtype == np.float32 else np.complex128)

    def __getitem__(self, idx) -> np.ndarray:
        out = np.asarray(self.real[idx]).astype(self.dtype)
        if self.imag is not None:
            out.imag = self.imag[idx]
        return out


class FileComplex:
    """A complex ``(n_snr, n_frames, L)`` container that is still in its FILE: the byte offsets of its real and
    imaginary arrays (column-major float32 / float64, how a level-5 .mat stores an uncompressed complex variable;
    ``imag_offset`` None: a real signal), or of ONE interleaved complex array (``interleaved=True``: a raw
    complex64 / complex128 stream, C-ordered; with ``order="F"`` the {real, imag} compound dataset of a MATLAB -v7.3
    file, whose bytes are the column-major variable).  Nothing is read or mapped here: the engine's staging threads pread
    the file block by block on their way to the pinned slots (``amcx_ctx_features18_strided_file``), so the
    variable never exists in host memory outside the page cache.  Indexing (tests, injected engines) goes
    through a memory mapping.  ``release()`` closes the descriptor."""

    def __init__(self, path, store_dtype, shape, real_offset: int, imag_offset: Optional[int] = None, *,
                 interleaved: bool = False, order: Optional[str] = None):
        self.path = Path(path)
        self.store = np.dtype(store_dtype)
        self.interleaved = bool(interleaved)
        if not self.interleaved and self.store not in (np.float32, np.float64):
            raise TypeError(f"split containers hold float32 or float64, got {self.store}")
        if self.interleaved and self.store not in (np.complex64, np.complex128):
            raise TypeError(f"interleaved containers hold complex64 or complex128, got {self.store}")
        self.shape, self.ndim = tuple(int(x) for x in shape), len(shape)
        self.real_offset, self.imag_offset = int(real_offset), (None if imag_offset is None else int(imag_offset))
        self.dtype = self.store if self.interleaved else \
            np.dtype(np.complex64 if self.store == np.float32 else np.complex128)
        # element strides: column-major for the split arrays of a .mat (and a -v7.3 compound), row-major for a raw stream
        self.order = order if order is not None else ("C" if self.interleaved else "F")
        if self.order not in ("C", "F"):
            raise ValueError("order is 'C' or 'F'")
        st, acc = [], 1
        for n in (self.shape if self.order == "F" else self.shape[::-1]):
            st.append(acc)
            acc *= n
        self.strides_elems = tuple(st if self.order == "F" else st[::-1])
        self.source = "file"
        self._fd, self._view, self._lock = None, None, threading.Lock()

    def fileno(self) -> int:
        with self._lock:
            if self._fd is None:
                self._fd = os.open(str(self.path), os.O_RDONLY)
            return self._fd

    def release(self) -> None:
        with self._lock:
            if self._fd is not None:
                os.close(self._fd)
                self._fd = None
            self._view = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def _mapped(self):
        if self._view is None:
            re = np.memmap(self.path, dtype=self.store, mode="r", offset=self.real_offset, shape=self.shape, order=self.order)
            if self.interleaved:
                self._view = re
            else:
                im = None if self.imag_offset is None else \
                    np.memmap(self.path, dtype=self.store, mode="r", offset=self.imag_offset, shape=self.shape, order=self.order)
                self._view = SplitComplex(re, im)
        return self._view

    def __getitem__(self, idx) -> np.ndarray:
        return np.asarray(self._mapped()[idx])


class FrameRows:
    """Frames ``[lo, hi)`` of ``parsed[:n_snr, :n_frames]`` flattened snr-major
    (frame g = snr * n_frames + k, the order feature_extraction.py:64-72 enqueues
    them in), WITHOUT materialising the flattening: for the Fortran-ordered arrays
    ``loadmat`` returns, ``reshape`` would be a full transposing copy."""

    def __init__(self, parsed, n_snr: int, n_frames: int, lo: int = 0, hi: Optional[int] = None):
        self.parsed, self.n_snr, self.n_frames = parsed, n_snr, n_frames
        self.lo = lo
        self.hi = n_snr * n_frames if hi is None else hi
        self.dtype = parsed.dtype

    @property
    def shape(self):
        return (self.hi - self.lo, self.parsed.shape[2])

    def slice(self, lo: int, hi: int) -> "FrameRows":
        return FrameRows(self.parsed, self.n_snr, self.n_frames, self.lo + lo, self.lo + hi)

    def blocks(self) -> Iterator[Tuple[int, int, int, int]]:
        """``(s0, s1, k0, k1)`` rectangles of the (snr, frame) grid that tile ``[lo, hi)`` in order:
        at most a partial first snr row, a run of whole rows, a partial last row."""
        a = self.lo
        while a < self.hi:
            s, k = divmod(a, self.n_frames)
            if k == 0 and self.hi - a >= self.n_frames:
                m = (self.hi - a) // self.n_frames
                yield s, s + m, 0, self.n_frames
                a += m * self.n_frames
            else:
                take = min(self.hi - a, self.n_frames - k)
                yield s, s + 1, k, k + take
                a += take

    def gather(self, dst: np.ndarray, g0: int, g1: int, n: int) -> None:
        """dst[(g1-g0), n] <- the first n samples of frames [g0, g1) of this range (host copy:
        tests and injected engines; the production engine never calls it)."""
        row = 0
        for s0, s1, k0, k1 in self.slice(g0, g1).blocks():
            for s in range(s0, s1):
                np.copyto(dst[row:row + k1 - k0], self.parsed[s, k0:k1, :n], casting="same_kind")
                row += k1 - k0

    def to_array(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        self.gather(out, 0, self.shape[0], self.shape[1])
        return out


class FrameColumns(FrameRows):
    """Frames ``[k_lo, k_hi)`` of EVERY snr row of ``parsed[:n_snr, :n_frames]``, snr-major (row
    ``s * (k_hi - k_lo) + (k - k_lo)``): a rank's share when a container is cut along its frame axis
    (``sharding.shard_by_frames``).  In a column-major container that is one contiguous run of every sample
    plane -- what the staging threads read from the file or copy at full rate."""

    def __init__(self, parsed, n_snr: int, n_frames: int, k_lo: int, k_hi: int):
        super().__init__(parsed, n_snr, n_frames, 0, n_snr * max(0, k_hi - k_lo))
        self.k_lo, self.k_hi = k_lo, max(k_lo, k_hi)

    def slice(self, lo: int, hi: int) -> "FrameRows":
        if (lo, hi) != (0, self.hi):
            raise NotImplementedError("a frame-axis share is taken whole")
        return self

    def blocks(self) -> Iterator[Tuple[int, int, int, int]]:
        if self.k_hi > self.k_lo and self.n_snr:
            yield 0, self.n_snr, self.k_lo, self.k_hi

    def gather(self, dst: np.ndarray, g0: int, g1: int, n: int) -> None:
        w = self.k_hi - self.k_lo
        for g in range(g0, g1):                           # host copy: tests and injected engines only
            s, k = divmod(g, w)
            np.copyto(dst[g - g0], self.parsed[s, self.k_lo + k, :n], casting="same_kind")


def _native_source(arr):
    """(keepalive, re_ptr, im_ptr, kind, element strides, bytes per element, fd) of a container the
    native engine can read in place -- pointers, or byte offsets into the file ``fd`` -- or None if it has to be
    copied first."""
    if isinstance(arr, FileComplex):
        if arr.interleaved:
            kind = _lib.SRC_C64 if arr.store == np.complex64 else _lib.SRC_C128
        else:
            kind = _lib.SRC_F32_SPLIT if arr.store == np.float32 else _lib.SRC_F64_SPLIT
        return arr, arr.real_offset, arr.imag_offset, kind, list(arr.strides_elems), arr.store.itemsize, arr.fileno()
    if isinstance(arr, SplitComplex):
        re, im = arr.real, arr.imag
        kind = _lib.SRC_F32_SPLIT if re.dtype == np.float32 else _lib.SRC_F64_SPLIT
    elif isinstance(arr, np.ndarray) and arr.dtype in (np.complex64, np.complex128, np.float32, np.float64):
        re, im = arr, None
        kind = {np.dtype(np.complex64): _lib.SRC_C64, np.dtype(np.complex128): _lib.SRC_C128,
                np.dtype(np.float32): _lib.SRC_F32_SPLIT, np.dtype(np.float64): _lib.SRC_F64_SPLIT}[arr.dtype]
    else:
        return None
    item = re.itemsize
    if any(st < 0 or st % item for st in re.strides):
        return None
    return (re, im), re.ctypes.data, (None if im is None else im.ctypes.data), kind, [st // item for st in re.strides], item, None


# ----------------------------------------------------------------------------
# host container -> HBM -> features: the native upload pipeline
# ----------------------------------------------------------------------------
class HipEngine:
    """``engine(frames) -> (F, 18) float32`` through device memory.

    ``frames`` is an (F, L) array / memmap, or a :class:`FrameRows` over an (n_snr, n_frames, L)
    container in any memory order (ndarray or :class:`SplitComplex`).  The container is read where it
    lies by ``amcx_ctx_features18_strided_host``: host threads stage contiguous runs -- sample planes
    of a Fortran-ordered container, rows of a C-ordered one -- into three pinned slots (rounding
    doubles to float32 on the way), the copy engine drains them, a device kernel transposes planes to
    frame-major, the feature kernel runs, the (F x 18) result comes back.  ``chunk_bytes`` is the size
    of one pinned slot, ``threads`` the staging threads (the reference's ``num_threads``).
    ``stats`` of the last call: frames, seconds, bytes over PCIe, source bytes, chunks, threads."""

    def __init__(self, frame_size: int, device: Optional[int] = None, chunk_bytes: int = 32 << 20,
                 threads: Optional[int] = None, round_on_device: bool = False):
        self.N = int(frame_size)
        if device is None:
            device = 0
            if _lib.torch_wanted():
                try:
                    import torch
                    if torch.cuda.is_available():
                        device = torch.cuda.current_device()
                except Exception:
                    pass
        self.device = int(device)
        self.chunk_bytes = int(chunk_bytes)
        self.threads = max(1, min(int(threads or 8), os.cpu_count() or 1))
        self.round_on_device = bool(round_on_device)
        self._ctx: Optional[_lib.HostContext] = None
        self.stats = {}

    def _context(self) -> "_lib.HostContext":
        if self._ctx is None:
            self._ctx = _lib.HostContext(self.device)
            self._ctx.configure(self.threads, self.chunk_bytes, int(self.round_on_device))
        return self._ctx

    def _run_block(self, src, base_elems: int, n_snr: int, n_frames: int, strides, out: np.ndarray) -> None:
        keep, re_ptr, im_ptr, kind, _, item, fd = src
        if fd is not None:
            # a file is read run by run (one pread each): worth it for whole planes or rows, not for the few snr
            # values per sample that a shard cut across the snr axis leaves of a column-major plane
            ss, sk, sn = strides
            if sn == 1:
                run = self.N * (n_frames if sk == self.N else 1)       # rows that follow each other are one read
            elif sk == 1 and n_frames > 1:
                run = n_snr * n_frames if ss == n_frames else n_frames
            else:
                run = n_snr * n_frames if sk == n_snr else n_snr
            if run * item < (8 << 10):
                src = _native_source(keep._mapped())
                keep, re_ptr, im_ptr, kind, _, item, fd = src
        ctx = self._context()
        run = ctx.run_strided if fd is None else functools.partial(ctx.run_file, fd)
        run(re_ptr + base_elems * item, None if im_ptr is None else im_ptr + base_elems * item,
            kind, n_snr, n_frames, self.N, strides, out)
        st = ctx.upload_stats()
        for k in ("frames", "source_bytes", "pcie_bytes", "chunks", "seconds_staging", "seconds_waiting",
                  "seconds_prepare", "seconds_tail", "seconds"):
            self.stats[k] = self.stats.get(k, 0) + st[k]
        self.stats["gather_threads"], self.stats["plane_major"] = st["threads"], st["plane_major"]
        self.stats["from_file"] = max(self.stats.get("from_file", 0), st["from_file"])

    def __call__(self, frames) -> np.ndarray:
        if isinstance(frames, FrameRows):
            rows = frames
        else:
            arr = frames if isinstance(frames, SplitComplex) else np.asarray(frames)
            if arr.ndim != 2:
                raise ValueError(f"expected (F, L) frames, got shape {arr.shape}")
            rows = FrameRows(arr[None] if not isinstance(arr, SplitComplex) else
                             SplitComplex(arr.real[None], None if arr.imag is None else arr.imag[None]),
                             1, arr.shape[0])
        F, L = rows.shape
        if L < self.N:
            raise ValueError(f"rows of {L} samples are shorter than frame_size {self.N}")
        if F == 0:
            return np.empty((0, 18), dtype=np.float32)
        t0 = time.perf_counter()
        self.stats = {}
        parsed = rows.parsed
        src = _native_source(parsed)
        if src is not None:
            st = src[4]
            unit = [st[2] == 1, st[1] == 1 and rows.n_frames > 1, st[0] == 1 or rows.n_snr == 1]
            if not any(unit):
                src = None
        if src is None:
            # integer / half / exotic dtypes, negative or sub-element strides, no contiguous axis: one
            # C-ordered copy of the part that is used, then the row path
            block = np.ascontiguousarray(rows.to_array(), dtype=np.complex64 if rows.dtype == np.complex64 else np.complex128)
            rows = FrameRows(block[None], 1, F)
            src = _native_source(rows.parsed)
        ss, sk, sn = src[4]
        out = np.empty((F, 18), dtype=np.float32)
        row = 0
        for s0, s1, k0, k1 in rows.blocks():
            n = (s1 - s0) * (k1 - k0)
            self._run_block(src, s0 * ss + k0 * sk, s1 - s0, k1 - k0, (ss, sk, sn), out[row:row + n])
            row += n
        self.stats["seconds_native"] = self.stats.pop("seconds", 0.0)
        self.stats["seconds"] = time.perf_counter() - t0
        self.stats["bytes_uploaded"] = self.stats.get("pcie_bytes", 0)
        return out

    def close(self) -> None:
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None


_ENGINES = {}


def default_engine(frame_size: int, device: Optional[int] = None, threads: Optional[int] = None) -> HipEngine:
    """The process's engine for (frame_size, device, threads), created on first use and kept: a context is two
    streams, three pinned slots and device scratch (~10 ms to set up, ~8 ms to tear down -- as long as a whole
    BASELINE configs[0] run takes).  :func:`release_engines` frees them."""
    probe = HipEngine(frame_size, device, threads=threads)
    key = (probe.N, probe.device, probe.threads)
    eng = _ENGINES.get(key)
    if eng is None:
        eng = _ENGINES[key] = probe
    return eng


def release_engines() -> None:
    """Free the contexts :func:`default_engine` keeps (pinned host memory, device scratch, staging threads)."""
    while _ENGINES:
        _ENGINES.popitem()[1].close()


def staging_threads_per_device(places, want: Optional[int] = None, allowed=None):
    """Staging threads for each engine of a one-process fan-out.  ``places[i] = (numa node, [cpus local to device i])``
    (``_lib.numa_place``; node -1: unknown).  An engine whose device has a known node shares that node's CPUs -- those
    of them this process may run on (``allowed``, default ``os.sched_getaffinity(0)``) -- with the other engines on the
    node; the rest share all allowed CPUs evenly, as before round 5.  At most ``want`` (default 8) and at least 1 each."""
    allowed = set(os.sched_getaffinity(0)) if allowed is None else set(allowed)
    want = max(1, int(want or 8))
    on_node = {}
    for node, _ in places:
        on_node[node] = on_node.get(node, 0) + 1
    out = []
    for node, cpus in places:
        local = len(allowed.intersection(cpus)) if node >= 0 else 0
        share = local // on_node[node] if local else len(allowed) // len(places)
        out.append(max(1, min(want, share)))
    return out


class DeviceFanOut:
    """``engine(rows) -> (F, 18) float32`` over SEVERAL devices from one process: one :class:`HipEngine` (context,
    streams, pinned slots, staging threads) per entry of ``devices`` and one host thread each -- the native call
    releases the GIL, ``hipSetDevice`` is per thread (the C ABI's host entries take the device index).  Where the
    reference forks a process per modulation and threads inside it (feature_extraction.py:58-61,89-97), this cuts
    every modulation over the devices: along its FRAME axis when that balances (``sharding.shard_by_frames`` -- one
    contiguous run per sample plane of a column-major .mat), along the snr-major flattening otherwise.  Each device's
    rows land directly in the result; no process group and no gather are involved.  A device may be listed twice
    (two contexts on it: how the one-GPU test box exercises the path)."""

    def __init__(self, frame_size: int, devices, threads: Optional[int] = None, chunk_bytes: int = 32 << 20):
        devices = [int(d) for d in devices]
        if not devices:
            raise ValueError("DeviceFanOut needs at least one device")
        have = _lib.load().amcx_device_count()
        bad = [d for d in devices if d < 0 or (have > 0 and d >= have)]
        if bad:
            raise ValueError(f"device index {bad[0]} out of range: {have} gfx950 device(s) visible")
        self.N, self.devices = int(frame_size), devices
        # the staging threads of all devices share the host's cores: a device's engine gets its share of the CPUs LOCAL to
        # it (the socket it hangs off: _lib.numa_place reads the kernel's PCI tree; the context binds its threads and
        # pinned slots there by itself), or of all CPUs when the platform does not say
        self.places = [self._place(d) for d in devices]
        per_dev = staging_threads_per_device(self.places, threads)
        self.engines = []
        try:
            for d, n in zip(devices, per_dev):
                self.engines.append(HipEngine(frame_size, d, chunk_bytes, threads=n))
        except BaseException:
            for e in self.engines:                          # a later device failed: the earlier ones' contexts, pinned slots
                e.close()                                   # and staging threads are released, not leaked
            raise
        self.threads = min(per_dev)
        self._pool = ThreadPoolExecutor(max_workers=len(devices), thread_name_prefix="amcx-device")
        self.stats = {}

    @staticmethod
    def _place(device: int):
        """(numa node, [local cpus]) of a device, (-1, []) if unknown or switched off (AMCX_NUMA=0)."""
        if os.environ.get("AMCX_NUMA", "1")[:1] == "0":
            return -1, []
        try:
            return _lib.numa_place(_lib.device_pci_bus_id(device), os.environ.get("AMCX_SYSFS_ROOT", ""))
        except Exception:
            return -1, []

    def placement(self):
        """What each engine's context bound itself to (amcx_ctx_placement), in device order."""
        return [e._context().placement() for e in self.engines]

    def shares(self, rows: FrameRows):
        """[(rows of device i, where they go)]: ``("columns", k_lo, k_hi)`` or ``("rows", lo, hi)``."""
        W = len(self.engines)
        whole = type(rows) is FrameRows and rows.lo == 0 and rows.hi == rows.n_snr * rows.n_frames
        if whole and shard_by_frames(rows.n_snr, rows.n_frames, W):
            cuts = [shard_range(rows.n_frames, r, W) for r in range(W)]
            return [(FrameColumns(rows.parsed, rows.n_snr, rows.n_frames, a, b), ("columns", a, b)) for a, b in cuts]
        F = rows.shape[0]
        cuts = [shard_range(F, r, W) for r in range(W)]
        return [(rows.slice(a, b), ("rows", a, b)) for a, b in cuts]

    def __call__(self, frames) -> np.ndarray:
        if not isinstance(frames, FrameRows):
            arr = np.asarray(frames)
            if arr.ndim != 2:
                raise ValueError(f"expected (F, L) frames, got shape {arr.shape}")
            frames = FrameRows(arr[None], 1, arr.shape[0])
        rows = frames
        F = rows.shape[0]
        out = np.empty((F, 18), dtype=np.float32)
        if F == 0:
            return out
        t0 = time.perf_counter()
        shares = self.shares(rows)

        def one(i):
            part, where = shares[i]
            return self.engines[i](part) if part.shape[0] else np.empty((0, 18), dtype=np.float32)

        futs = [self._pool.submit(one, i) for i in range(len(shares))]
        failures = []
        for i, fut in enumerate(futs):                      # every device finishes (or fails) before anything is raised
            try:
                blk = fut.result()
            except Exception as exc:
                failures.append(f"device {self.devices[i]}: {type(exc).__name__}: {exc}")
                continue
            kind, a, b = shares[i][1]
            if kind == "rows":
                out[a:b] = blk
            elif b > a:
                out.reshape(rows.n_snr, rows.n_frames, 18)[:, a:b] = blk.reshape(rows.n_snr, b - a, 18)
        if failures:
            raise RuntimeError("feature extraction failed on " + "; ".join(failures))
        self.stats = {"seconds": time.perf_counter() - t0, "devices": list(self.devices),
                      "frames_per_device": [sh[0].shape[0] for sh in shares],
                      "bytes_uploaded": sum(e.stats.get("bytes_uploaded", 0) for e in self.engines),
                      "source_bytes": sum(e.stats.get("source_bytes", 0) for e in self.engines)}
        return out

    def close(self) -> None:
        self._pool.shutdown(wait=True)
        for e in self.engines:
            getattr(e, "close", lambda: None)()


def default_fanout(frame_size: int, devices, threads: Optional[int] = None) -> DeviceFanOut:
    """The process's :class:`DeviceFanOut` for (frame_size, devices, threads), kept like :func:`default_engine`'s."""
    key = (int(frame_size), tuple(int(d) for d in devices), max(1, int(threads or 8)))
    eng = _ENGINES.get(key)
    if eng is None:
        eng = _ENGINES[key] = DeviceFanOut(frame_size, devices, threads)
    return eng


def _check_container(parsed, cfg: Config):
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
        if shard_by_frames(n_snr, n_frames, world):          # frames [k_lo, k_hi) of every snr row (sharding.py)
            k_lo, k_hi = shard_range(n_frames, rank, world)
            local = engine(FrameColumns(parsed, n_snr, n_frames, k_lo, k_hi)) if k_hi > k_lo else \
                np.empty((0, 18), dtype=np.float32)
            return gather_frame_columns(local, n_snr, n_frames, rank, world, group)
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
    consecutive ``frame_size``-sample frames (a trailing partial frame is dropped); the staging
    threads read the file slot by slot, so it never has to fit in host memory.
    Returns ``(n_frames, 18)`` float32."""
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
    if compute is None:                 # the staging threads read the file themselves, part by part
        stream = FileComplex(path, np.complex64, (1, n_frames, frame_size), 8 * skip_samples, interleaved=True)
        try:
            return np.asarray(HipEngine(frame_size, device)(FrameRows(stream, 1, n_frames)), dtype=np.float32)
        finally:
            stream.release()
    frames = np.memmap(path, dtype=np.complex64, mode="r", offset=8 * skip_samples,
                       shape=(n_frames, frame_size))
    return np.asarray(compute(frames), dtype=np.float32)


def _pairs_as_complex(block: np.ndarray) -> np.ndarray:
    """(g, L, 2) float32 -> (g, L) complex64 view (bit-identical layouts); copies only if the pairs
    are not interleaved in memory."""
    if block.strides[-1] != 4 or block.strides[-2] != 8:
        block = np.ascontiguousarray(block)
    return block.view(np.complex64)[..., 0]


def extract_iq_pairs(dataset, frame_size: Optional[int] = None, *, first_frame: int = 0,
                     max_frames: Optional[int] = None, compute=None, device: Optional[int] = None,
                     chunk_frames: Optional[int] = None) -> np.ndarray:
    """Features of frames stored as float32 (I, Q) pairs, ``dataset[f, n] = (I, Q)`` -- RadioML's
    ``(F, 1024, 2)`` layout (reference old/dataset.py:50-56, old/dataset_analysis.py:22).  ``dataset``
    is anything sliceable with ``.shape`` and ``.dtype``.  A numpy array or memmap is re-viewed as
    complex64 and goes up in one native call; any other dataset (an ``h5py.Dataset``, which decodes
    chunks from the file as they are sliced) is read ``chunk_frames`` at a time by a reader thread one
    chunk ahead of the upload, so the set never has to fit in host memory.  Returns ``(n_frames, 18)``
    float32."""
    shape = tuple(dataset.shape)
    if len(shape) != 3 or shape[2] != 2:
        raise ValueError(f"expected an (F, L, 2) dataset of (I, Q) pairs, got shape {shape}")
    if np.dtype(dataset.dtype) != np.float32:
        raise TypeError(f"(I, Q) pairs must be float32, got {dataset.dtype}")
    F, L = int(shape[0]), int(shape[1])
    N = L if frame_size is None else int(frame_size)
    if N < 2 or N > L:
        raise ValueError(f"frame_size {N} outside 2 .. {L}")
    lo = min(max(0, int(first_frame)), F)
    hi = F if max_frames is None else min(F, lo + int(max_frames))
    if hi <= lo:
        return np.empty((0, 18), dtype=np.float32)
    if compute is not None:                       # injected engine (tests): plain (F, N) arrays
        block = np.ascontiguousarray(_pairs_as_complex(np.asarray(dataset[lo:hi]))[:, :N])
        return np.asarray(compute(block), dtype=np.float32)
    from . import hdf5_min
    if (isinstance(dataset, hdf5_min.Dataset) and N == L and dataset.file_offset is not None and dataset.little_endian):
        # a contiguous dataset is a raw interleaved complex64 stream at a known place in the file: the staging threads
        # read it themselves, slot by slot (as extract_raw_stream does), and libhdf5 is not on the data path at all
        stream = FileComplex(dataset.file_path, np.complex64, (1, hi - lo, L), dataset.file_offset + lo * L * 8, interleaved=True)
        try:
            return np.asarray(HipEngine(N, device)(FrameRows(stream, 1, hi - lo)), dtype=np.float32)
        finally:
            stream.release()
    engine = HipEngine(N, device)
    if isinstance(dataset, np.ndarray):
        return engine(_pairs_as_complex(dataset[lo:hi]))
    step = int(chunk_frames or max(1, (256 << 20) // (L * 8)))
    spans = [(a, min(hi, a + step)) for a in range(lo, hi, step)]
    out = np.empty((hi - lo, 18), dtype=np.float32)
    for (a, b), fut in _prefetched(spans, lambda ab: _pairs_as_complex(np.asarray(dataset[ab[0]:ab[1]]))):
        out[a - lo:b - lo] = engine(fut.result())
    return out


def extract_radioml_hdf5(path, *, key: str = "X", frame_size: Optional[int] = None, first_frame: int = 0,
                         max_frames: Optional[int] = None, device: Optional[int] = None, compute=None,
                         chunk_frames: Optional[int] = None) -> np.ndarray:
    """``extract_iq_pairs`` on dataset ``key`` of a RadioML-style HDF5 file (``GOLD_XYZ_OSC.0001_1024.hdf5``:
    ``X`` float32 (2 555 904, 1024, 2), reference old/dataset.py:43-56).  The file is opened with the HDF5 C library
    through ``amcpy_amd.hdf5_min`` (ctypes) where one is found, otherwise with ``h5py``, which the reference lists for
    its legacy scripts (this image ships libhdf5 1.10.6 but no h5py for its interpreter).  A CONTIGUOUS ``X`` at its full
    frame length is then a raw complex64 stream at a known file offset and the engine's staging threads read it
    themselves; a chunked or compressed one is decoded by the library ``chunk_frames`` rows at a time on a reader thread
    ahead of the upload.  Neither library there: ImportError that says so."""
    from . import hdf5_min
    kw = dict(first_frame=first_frame, max_frames=max_frames, device=device, compute=compute, chunk_frames=chunk_frames)
    if hdf5_min.available():                      # the C library itself: a contiguous X then bypasses it altogether
        with hdf5_min.File(path) as fh:
            return extract_iq_pairs(fh[key], frame_size, **kw)
    try:
        import h5py
    except ImportError as exc:                    # not a silent fallback: say what is missing
        raise ImportError("extract_radioml_hdf5 needs an HDF5 C library >= 1.10 (AMCX_LIBHDF5=/path/to/libhdf5.so) or h5py "
                          "(pip install h5py); any sliceable (F, L, 2) float32 dataset can be passed to extract_iq_pairs "
                          "instead") from exc
    with h5py.File(str(path), "r") as fh:
        return extract_iq_pairs(fh[key], frame_size, **kw)


# ----------------------------------------------------------------------------
# run_extraction
# ----------------------------------------------------------------------------
def _prefetched(items: List, fn: Callable, depth: int = 3):
    """``(item, future)`` pairs with ``fn(item)`` running on reader threads up to ``depth`` items ahead of the
    consumer: while the caller works on item k, items k+1 .. k+depth are being read / decoded (a memory-mapped
    variable costs nothing to "read"; a compressed one is a zlib inflate of hundreds of megabytes, which
    releases the GIL -- MATLAB's default `save` compresses).  At most ``depth + 1`` items are alive."""
    depth = max(1, int(depth))
    with ThreadPoolExecutor(max_workers=depth, thread_name_prefix="amcx-reader") as ex:
        futs = [ex.submit(fn, it) for it in items[:depth]]
        for i, it in enumerate(items):
            cur = futs[i]
            if i + depth < len(items):
                futs.append(ex.submit(fn, items[i + depth]))
            yield it, cur
            futs[i] = None                                  # drop the reference: the consumer is done with it


def _shared_dir(need_bytes: int) -> Path:
    """Where rank 0 publishes a modulation for the other ranks of the node: /dev/shm (page cache, no
    disk) when it has room, the temp dir otherwise.  Writing a sparse tmpfs file past the mount's
    capacity raises SIGBUS, not an exception, so room is checked BEFORE the file is mapped (a
    container's default /dev/shm is 64 MB; a configs[1] modulation is 3.5 GB)."""
    for cand in (Path("/dev/shm"), Path(tempfile.gettempdir())):
        try:
            if cand.is_dir() and os.access(cand, os.W_OK):
                st = os.statvfs(cand)
                if st.f_bavail * st.f_frsize >= need_bytes + (64 << 20):
                    return cand
        except OSError:
            continue
    raise OSError(f"neither /dev/shm nor {tempfile.gettempdir()} has {need_bytes / 1e9:.2f} GB free to publish a "
                  "modulation to the other ranks")


def _copy_parallel(dst: np.ndarray, src, threads: int) -> None:
    """dst <- src over host threads, split along the slowest axis of dst (numpy releases the GIL in copyto)."""
    axis = int(np.argmax(dst.strides))
    n = dst.shape[axis]
    threads = max(1, min(threads, n))
    idx = [slice(None)] * dst.ndim

    def part(a, b):
        sl = list(idx)
        sl[axis] = slice(a, b)
        np.copyto(dst[tuple(sl)], src[tuple(sl)], casting="same_kind")

    if threads == 1:
        part(0, n)
        return
    per = -(-n // threads)
    with ThreadPoolExecutor(max_workers=threads, thread_name_prefix="amcx-publish") as ex:
        for f in [ex.submit(part, a, min(n, a + per)) for a in range(0, n, per)]:
            f.result()


def _publish_container(parsed, n_snr: int, n_frames: int, N: int, threads: int) -> Path:
    """Rank 0: the part of the container the configuration uses -> an .npy in shared memory, IN THE
    MEMORY ORDER IT HAS (Fortran for what loadmat returns: the copy is a run of contiguous planes,
    no transposition), source dtype kept."""
    used = parsed[:n_snr, :n_frames, :N]                 # a view (ndarray) or the assembled part (SplitComplex)
    dtype = used.dtype if used.dtype in (np.complex64, np.complex128) else np.dtype(np.complex128)
    fortran = used.strides[0] < used.strides[2]
    need = n_snr * n_frames * N * dtype.itemsize
    fd, name = tempfile.mkstemp(prefix="amcx_frames_", suffix=".npy", dir=str(_shared_dir(need)))
    os.close(fd)
    try:
        mm = np.lib.format.open_memmap(name, mode="w+", dtype=dtype, shape=(n_snr, n_frames, N), fortran_order=fortran)
        _copy_parallel(mm, used, threads)
        mm.flush()
        del mm
    except BaseException:
        Path(name).unlink(missing_ok=True)
        raise
    return Path(name)


def _load_variable(mat_path: Path, key: str, pool=None, direct: bool = False):
    from .matfile import load_variable
    return load_variable(mat_path, key, pool, direct)


def _read_ahead(inflated_bytes: int, n_variables: int) -> int:
    """How many variables the reader threads decode ahead of the GPU.  An uncompressed container: 1 (it is only
    located, or read at link rate).  A compressed one is inflate-bound, one deflate stream per variable, so every
    variable ahead is a core at work (2.2 / 0.9 / 0.62 s for the 2.6 GB container at 1 / 3 / 6,
    profiles/r3_extract_ab_readahead.txt): up to six, as many as fit twice over in a quarter of the memory the host
    has available.  AMCX_READ_AHEAD overrides."""
    if "AMCX_READ_AHEAD" in os.environ:
        return max(1, int(os.environ["AMCX_READ_AHEAD"]))
    if inflated_bytes <= 0:
        return 1
    avail = 0
    try:
        with open("/proc/meminfo") as fh:
            for line in fh:
                if line.startswith("MemAvailable:"):
                    avail = int(line.split()[1]) * 1024
                    break
    except OSError:
        pass
    fit = (avail // 4) // (2 * inflated_bytes) if avail else 3
    return int(max(1, min(6, n_variables, (os.cpu_count() or 2) - 1, fit)))


def _placement(world: int, device: Optional[int]) -> bool:
    """One all-gather of (host name, device index) per run: True when every rank is on one host (rank 0 then decodes
    for all).  Two ranks on the SAME device of the same host -- a launcher whose ranks never called
    ``set_device(LOCAL_RANK)`` -- would be silently correct and ``world`` times slow: that raises on every rank
    unless AMCX_SHARE_GPU=1 says it is meant (rehearsals on a one-GPU box).  ``device`` None: an injected engine."""
    import torch.distributed as dist
    where = [None] * world
    dist.all_gather_object(where, (socket.gethostname(), device))
    if device is not None and os.environ.get("AMCX_SHARE_GPU", "0") != "1":
        seen = {}
        for r, hd in enumerate(where):
            if hd[1] is not None and hd in seen:
                raise RuntimeError(f"ranks {seen[hd]} and {r} both compute on device {hd[1]} of {hd[0]}: give every rank "
                                   f"its own GPU (torch.cuda.set_device(LOCAL_RANK) before run_extraction, or device=), "
                                   f"or set AMCX_SHARE_GPU=1 if sharing is intended")
            seen[hd] = r
    return len({h for h, _ in where}) == 1


def _provenance(cfg: Config, mat_path: Path, key: str) -> dict:
    """What a feature file was computed FROM and FOR: the shape (n_snr, n_frames, 18) says neither the frame size nor
    which SNR labels or which input container -- a file written for another ``--frame-size``, or for a container that
    has since been replaced, has the right shape and stale numbers."""
    try:
        st = Path(mat_path).stat()
        src = {"input_size": st.st_size, "input_mtime_ns": st.st_mtime_ns}
    except OSError:
        src = {"input_size": None, "input_mtime_ns": None}
    return {"frame_size": int(cfg.signals.frame_size), "num_frames": int(cfg.signals.num_frames),
            "snr_values": [[int(k), str(v)] for k, v in cfg.signals.snr_values.items()],
            "input": str(Path(mat_path).name), "variable": key, **src}


def _provenance_path(out_path: Path) -> Path:
    """Beside the feature file, not inside it: the .mat keeps exactly the two variables the reference writes
    (feature_extraction.py:77-81), which is what its downstream loaders see."""
    return out_path.with_name(out_path.stem + ".provenance.json")


def _already_extracted(out_path: Path, key: str, shape, provenance: Optional[dict] = None) -> bool:
    """True if ``out_path`` is a complete feature file for this configuration: it holds ``key`` as a float32 array
    of ``shape`` (and the ``Modulation`` string) -- only the variable headers are read (``scipy.io.whosmat``) -- AND the
    provenance record written beside it equals ``provenance`` (frame size, SNR labels, frame count, the input
    container's name, size and modification time).  A file that is cut short, was written for another frame size / SNR
    grid / container, or has no record (written before records existed, or by the reference) does not qualify."""
    import json
    import scipy.io
    try:
        seen = {name: (tuple(shp), cls) for name, shp, cls in scipy.io.whosmat(str(out_path))}
        if seen.get(key) != (tuple(shape), "single") or "Modulation" not in seen:
            return False
        size = out_path.stat().st_size
        if size < 4 * int(np.prod(shape)):                  # the array's bytes are really there
            return False
        if provenance is None:
            return True
        return json.loads(_provenance_path(out_path).read_text()) == provenance
    except Exception:
        return False


def run_extraction(cfg: Config, *, compute=None, device: Optional[int] = None, devices=None,
                   verbose: bool = True, resume: bool = False) -> None:
    """Drop-in for the reference's ``run_extraction(cfg)``: writes one
    ``{mod}_features.mat`` per entry of ``cfg.signals.modulations_with_noise``.
    ``devices``: several GPU indices driven from THIS process (:class:`DeviceFanOut`; not together with a process
    group of several ranks, where every rank has its one ``device``).
    ``resume``: modulations whose feature file is already there, complete, of this configuration's shape AND recorded
    (``{mod}_features.provenance.json`` beside it) as computed for this frame size, these SNR labels and this very input
    container (name, size, modification time) are skipped -- the per-modulation file is the path's natural resume unit (the reference recomputes everything,
    all-or-nothing per file; a file is written by ONE savemat call at the end of its modulation, here as there)."""
    import scipy.io

    rank, world = _rank_world()
    cfg.paths.ensure_dirs()
    mat_path = cfg.paths.mat_data / cfg.paths.mat_filename
    N = cfg.signals.frame_size
    threads = max(1, int(cfg.signals.num_threads))
    if devices is not None:
        devices = [int(d) for d in devices]
        if compute is not None or device is not None:
            raise ValueError("devices= stands in for device= / compute=")
        if world > 1:
            raise ValueError("devices= drives several GPUs from one process; with a process group of several ranks "
                             "every rank takes its one device=")
        if len(devices) == 1:
            device, devices = devices[0], None
    if compute is not None:
        engine = compute
    elif devices:
        engine = default_fanout(N, devices, threads)
    else:
        engine = default_engine(N, device, threads)
    mods = list(cfg.signals.modulations_with_noise)
    if resume:
        todo = mods
        if rank == 0:
            shape = (len(cfg.signals.snr_values), cfg.signals.num_frames, 18)
            todo = [m for m in mods if not _already_extracted(cfg.paths.calculated_features / f"{m}_features.mat",
                                                              cfg.signals.mat_info[m], shape,
                                                              _provenance(cfg, mat_path, cfg.signals.mat_info[m]))]
            if verbose and len(todo) < len(mods):
                print(f"resume: {len(mods) - len(todo)} of {len(mods)} feature files are complete, computing {todo}")
        if world > 1:                                       # every rank loops over the same modulations
            import torch.distributed as dist
            box = [todo]
            dist.broadcast_object_list(box, src=0)
            todo = box[0]
        mods = todo
    t_start = time.perf_counter()

    def run(rows: FrameRows) -> np.ndarray:
        if rows.shape[0] == 0:
            return np.empty((0, 18), dtype=np.float32)
        mat = engine(rows) if compute is None else compute(rows.to_array())
        return np.asarray(mat, dtype=np.float32)

    def save(mod: str, key: str, feats: np.ndarray, t0: float) -> None:
        out_path = cfg.paths.calculated_features / f"{mod}_features.mat"
        # written aside and renamed: an interrupted run never leaves a partial file under the final name (what
        # resume= and every downstream loader look at)
        tmp_path = out_path.with_name(f"{out_path.stem}.{os.getpid()}.tmp.mat")
        record = _provenance_path(out_path)
        try:
            record.unlink(missing_ok=True)                  # never a fresh record beside an old file, or an old one beside a new
            scipy.io.savemat(str(tmp_path), {"Modulation": mod, key: feats})
            os.replace(tmp_path, out_path)
            record.write_text(__import__("json").dumps(provenance[mod]) + "\n")
        finally:
            tmp_path.unlink(missing_ok=True)
        if verbose:
            print(f"[{mod}] {feats.shape[0] * feats.shape[1]} frames in "
                  f"{time.perf_counter() - t0:.2f}s -> {out_path}")

    # uncompressed variables go from the file to the pinned slots inside the native engine (AMCX_DIRECT_FILE=0: read /
    # map them in Python first, the round-3 path kept for A/B runs); an injected engine gets arrays
    direct = compute is None and os.environ.get("AMCX_DIRECT_FILE", "1") != "0"
    # what each file is computed from, taken BEFORE the container is read: a container replaced mid-run leaves a record
    # that no longer matches it
    provenance = {m: _provenance(cfg, mat_path, cfg.signals.mat_info[m]) for m in mods}
    writer = ThreadPoolExecutor(max_workers=1, thread_name_prefix="amcx-writer") if rank == 0 else None
    writes = []
    published: List[Path] = []          # rank 0: shared files not yet removed
    feed = None
    try:
        if world == 1 and not (collectives_forced() and _process_group_up()):
            from .matfile import BufferPool, compressed_variable_bytes
            # A compressed container is inflate-bound: three reader threads run ahead (zlib releases the GIL).  An
            # uncompressed variable is only LOCATED here: the native engine's staging threads read it from the file
            # on their way to the pinned slots.  With an injected engine (tests) it is read with preadv into two
            # pairs of buffers that take turns: reading is faster than first-touching fresh pages, mapped or allocated.
            pool = BufferPool()
            depth = _read_ahead(compressed_variable_bytes(mat_path), len(mods))
            feed = _prefetched(mods, lambda m: _load_variable(mat_path, cfg.signals.mat_info[m], pool, direct),
                               depth)
            for mod, fut in feed:
                t0 = time.perf_counter()
                key = cfg.signals.mat_info[mod]
                parsed = fut.result()                   # the reader threads are up to three variables ahead
                try:
                    n_snr, n_frames, _ = _check_container(parsed, cfg)
                    feats = run(FrameRows(parsed, n_snr, n_frames)).reshape(n_snr, n_frames, 18)
                finally:
                    getattr(parsed, "release", lambda: None)()
                del parsed
                writes.append(writer.submit(save, mod, key, feats, t0))
        else:
            import torch.distributed as dist
            shared_host = _placement(world, getattr(engine, "device", None))
            mapped = {}                                 # rank 0: variables its reader thread has already mapped

            def decode_and_publish(mod):                # rank 0's reader thread
                parsed = _load_variable(mat_path, cfg.signals.mat_info[mod], None, direct)
                n_snr, n_frames, _ = _check_container(parsed, cfg)
                if getattr(parsed, "source", None) in ("mapped", "file"):
                    mapped[mod] = parsed                # every rank reads / maps the variable itself: nothing to publish
                    return "", n_snr, n_frames
                path = _publish_container(parsed, n_snr, n_frames, N, threads)
                published.append(path)
                return str(path), n_snr, n_frames

            def decode_locally(mod):                    # ranks on different hosts: as the reference's children do
                parsed = _load_variable(mat_path, cfg.signals.mat_info[mod], None, direct)
                n_snr, n_frames, _ = _check_container(parsed, cfg)
                return parsed, n_snr, n_frames

            if not shared_host:
                feed = _prefetched(mods, decode_locally)
            elif rank == 0:
                feed = _prefetched(mods, decode_and_publish)
            else:
                feed = ((m, None) for m in mods)
            for mod, fut in feed:
                t0 = time.perf_counter()
                key = cfg.signals.mat_info[mod]
                # 1. the modulation: every rank learns where it is, or that rank 0 could not read it
                meta, parsed = [None], None
                if shared_host:
                    if rank == 0:
                        try:
                            meta = [("ok",) + fut.result()]
                        except Exception as exc:        # every rank must leave the collective
                            meta = [("error", repr(exc), 0, 0)]
                    dist.broadcast_object_list(meta, src=0)
                    status, shared, n_snr, n_frames = meta[0]
                    if status != "ok":
                        raise RuntimeError(f"rank 0 could not read {key!r} from {mat_path}: {shared}")
                # 2. this rank's frame range; a failure is kept until every rank has reported
                local, failure, by_frames = None, None, False
                try:
                    if not shared_host:
                        parsed, n_snr, n_frames = fut.result()
                    elif shared:
                        parsed = np.load(shared, mmap_mode="r")
                    else:                               # mapped straight from the container, by every rank
                        parsed = mapped.pop(mod, None) if rank == 0 else None
                        if parsed is None:
                            parsed = _load_variable(mat_path, key, None, direct)
                        _check_container(parsed, cfg)
                    F = n_snr * n_frames
                    by_frames = shard_by_frames(n_snr, n_frames, world)     # the same answer on every rank
                    if by_frames:
                        k_lo, k_hi = shard_range(n_frames, rank, world)
                        local = run(FrameColumns(parsed, n_snr, n_frames, k_lo, k_hi))
                        expect = n_snr * (k_hi - k_lo)
                    else:
                        lo, hi = shard_range(F, rank, world)
                        local = run(FrameRows(parsed, n_snr, n_frames, lo, hi))
                        expect = hi - lo
                    # a wrong row count would raise inside the gather on THIS rank only and leave the others in the
                    # collective: it is reported with the status word instead
                    if local.shape != (expect, 18):
                        raise RuntimeError(f"engine returned {local.shape} for {expect} frames")
                except Exception as exc:
                    failure = f"{type(exc).__name__}: {exc}"
                del parsed
                # 3. one status word per rank BEFORE the data collective: all ranks raise together
                # (the all-gather is also the point after which nobody maps the shared file any more)
                statuses = [None] * world
                dist.all_gather_object(statuses, failure)
                if shared_host and rank == 0 and shared:
                    Path(shared).unlink(missing_ok=True)
                    published.remove(Path(shared))
                bad = [(r, s) for r, s in enumerate(statuses) if s is not None]
                if bad:
                    raise RuntimeError(f"feature extraction of {mod!r} failed on " +
                                       "; ".join(f"rank {r}: {s}" for r, s in bad))
                mat = gather_frame_columns(local, n_snr, n_frames, rank, world) if by_frames else \
                    gather_rows(local, F, rank, world)
                if rank == 0:
                    writes.append(writer.submit(save, mod, key, mat.reshape(n_snr, n_frames, 18), t0))
        for w in writes:
            w.result()                                  # a failed savemat raises here
    finally:
        if feed is not None and hasattr(feed, "close"):
            feed.close()                                # waits for a decode / publish still in flight
        if writer is not None:
            writer.shutdown(wait=True)
        for path in list(published):
            Path(path).unlink(missing_ok=True)
    if verbose and rank == 0:
        print(f"All feature calculations complete! ({time.perf_counter() - t_start:.2f}s)")
