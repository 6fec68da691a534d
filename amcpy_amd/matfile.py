"""Reading one variable of the reference's input container without decoding it.

The reference calls ``scipy.io.loadmat(path)`` -- the whole file, in each of its six child
processes (feature_extraction.py:46-48) -- and gets, per modulation, a Fortran-ordered complex128
array that scipy assembles from the file's two real arrays (``real + 1j * imag``: several passes
over 16 bytes per sample).  A MATLAB level-5 file already holds the data the GPU path wants:
the real parts and the imaginary parts of a variable are two contiguous little-endian arrays in
column-major order.  :func:`load_variable` therefore memory-maps the file, walks the element tags
to the variable asked for and returns the two arrays as views of the mapping
(:class:`~amcpy_amd.feature_extraction.SplitComplex`); the upload path's staging threads read them
straight out of the page cache, interleave and round them on their way to pinned memory.  Nothing
is decoded, copied or allocated per sample on the way.

Level-5 layout restated from the published MAT-file format (MathWorks "MAT-File Format", R2019b,
ch. 1): 128-byte header (bytes 124-125 version 0x0100, 126-127 endian indicator "IM" when the file
is little-endian); then data elements, each an 8-byte tag (uint32 type, uint32 byte count; "small"
form when the upper 16 bits of the first word are non-zero: count there, type below, <= 4 data bytes
in the second word) and data padded to 8 bytes.  miMATRIX (14) holds sub-elements: array flags
(class in byte 0, complex = bit 0x0800), dimensions (int32), name (int8), real part, imaginary
part.  miCOMPRESSED (15) wraps one zlib-deflated miMATRIX.  Anything this reader does not take on
-- big-endian files, integer-compressed numeric data, sparse / cell / struct / char variables, level
4 files -- goes to ``scipy.io.loadmat``, whose result for the variable is returned as it is.

``-v7.3`` files are HDF5 behind a 512-byte MATLAB header, and the only form MATLAB has for a variable above 2 GB -- one
BASELINE configs[1] modulation is 3.49 GB of complex128.  ``scipy.io.loadmat`` refuses them ("Please use HDF reader"),
and so does the reference (feature_extraction.py:46-47).  :func:`load_variable_v73` reads them through
:mod:`amcpy_amd.hdf5_min`: a variable is a root-level dataset with its dimensions reversed (the bytes are the
column-major variable), a complex one the compound ``{real, imag}``, the class in the ``MATLAB_class`` attribute.  A
CONTIGUOUS dataset is only located (``H5Dget_offset``) and read from the file by the engine's staging threads as
interleaved complex128 / complex64; a chunked or compressed one is decoded by libhdf5.  Classes other than ``double`` /
``single`` and files without a usable libhdf5 fall through to ``scipy.io.loadmat`` and its own error.

Mapping is free, but the first touch of every mapped page is a minor fault, and those do not scale: 13-18 GB/s
from one thread, 22 GB/s from eight (`profiles/r3_read_probe.txt`) -- 30 ms for a 436 MB variable whose upload
takes 4.5 ms.  ``preadv`` into a buffer that is reused from variable to variable does 51 GB/s from eight
threads, so a single-process ``run_extraction`` reads uncompressed variables that way (:func:`load_variable`
with a :class:`BufferPool`) and keeps the mapping for the cases where only part of a variable is wanted (a
rank's frame range) or nothing can be reused.
"""
from __future__ import annotations

import mmap
import os
import struct
import threading
import zlib
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Optional, Tuple

import numpy as np

MI_INT8, MI_INT32, MI_UINT32, MI_SINGLE, MI_DOUBLE, MI_MATRIX, MI_COMPRESSED = 1, 5, 6, 7, 9, 14, 15
MX_DOUBLE, MX_SINGLE = 6, 7
_STORAGE = {MI_DOUBLE: np.dtype("<f8"), MI_SINGLE: np.dtype("<f4")}


class _Unsupported(Exception):
    """The fast reader does not take this file / variable on: scipy decodes it."""


def _tag(buf, pos: int) -> Tuple[int, int, int, int]:
    """(type, byte count, offset of the data, offset of the next element) of the element at pos."""
    w0, w1 = struct.unpack_from("<II", buf, pos)
    if w0 >> 16:                                   # small element: count and type share the first word
        return w0 & 0xFFFF, w0 >> 16, pos + 4, pos + 8
    return w0, w1, pos + 8, pos + 8 + ((w1 + 7) & ~7)


def _matrix_header(buf, pos: int, end: int):
    """Of the miMATRIX body at [pos, end): (class, is_complex, dims, name, offset of the real-part element)."""
    t, n, d, nxt = _tag(buf, pos)
    if t != MI_UINT32 or n < 8:
        raise _Unsupported("array flags")
    flags = struct.unpack_from("<I", buf, d)[0]
    t, n, d, nxt2 = _tag(buf, nxt)
    if t != MI_INT32:
        raise _Unsupported("dimensions")
    dims = struct.unpack_from(f"<{n // 4}i", buf, d)
    t, n, d, nxt3 = _tag(buf, nxt2)
    if t != MI_INT8:
        raise _Unsupported("name")
    name = bytes(buf[d:d + n]).decode("latin-1")
    return flags & 0xFF, bool(flags & 0x0800), dims, name, nxt3


def _numeric_layout(buf, pos: int, dims, is_complex: bool):
    """(storage dtype, element count, offset of the real data, offset of the imaginary data or None)."""
    t, n, d, nxt = _tag(buf, pos)
    count = int(np.prod(dims, dtype=np.int64))
    if t not in _STORAGE or n != count * _STORAGE[t].itemsize:
        raise _Unsupported(f"numeric data stored as MAT type {t}")
    d2 = None
    if is_complex:
        t2, n2, d2, _ = _tag(buf, nxt)
        if t2 != t or n2 != n:
            raise _Unsupported("imaginary part stored differently from the real part")
    return _STORAGE[t], count, d, d2


def _numeric_parts(buf, pos: int, end: int, dims, is_complex: bool):
    dt, count, d, d2 = _numeric_layout(buf, pos, dims, is_complex)
    real = np.frombuffer(buf, dtype=dt, count=count, offset=d).reshape(dims, order="F")
    imag = None if d2 is None else np.frombuffer(buf, dtype=dt, count=count, offset=d2).reshape(dims, order="F")
    return real, imag


class BufferPool:
    """Reusable host buffers for :func:`load_variable`: a variable read with ``preadv`` lands in arrays that the
    previous variables of the same shape used, so their pages are already there (thread-safe)."""

    def __init__(self):
        self._free, self._lock = {}, threading.Lock()

    def take(self, dtype, count: int) -> np.ndarray:
        with self._lock:
            lst = self._free.get((np.dtype(dtype).str, count))
            if lst:
                return lst.pop()
        return np.empty(count, dtype=dtype)

    def give(self, arr: np.ndarray) -> None:
        with self._lock:
            self._free.setdefault((arr.dtype.str, arr.size), []).append(arr)


_PREAD_THREADS = 8
_pread_pool = None
_pread_lock = threading.Lock()


def _pread_executor() -> ThreadPoolExecutor:
    global _pread_pool
    with _pread_lock:
        if _pread_pool is None:
            _pread_pool = ThreadPoolExecutor(max_workers=_PREAD_THREADS, thread_name_prefix="amcx-pread")
        return _pread_pool


def _pread_into(path, jobs) -> None:
    """jobs: (destination uint8 view, file offset) pairs; pieces of <= 16 MiB spread over the pread threads
    (os.preadv releases the GIL; 51 GB/s from eight threads out of the page cache)."""
    piece = 16 << 20
    fd = os.open(str(path), os.O_RDONLY)
    try:
        def one(view, off):
            done, n = 0, len(view)
            while done < n:
                got = os.preadv(fd, [view[done:]], off + done)
                if got <= 0:
                    raise OSError(f"short read from {path} at offset {off + done}")
                done += got
        futs = []
        ex = _pread_executor()
        for dst, off in jobs:
            mv = memoryview(dst)
            for a in range(0, len(mv), piece):
                futs.append(ex.submit(one, mv[a:a + piece], off + a))
        for f in futs:
            f.result()
    finally:
        os.close(fd)


def _peek(comp: memoryview) -> Tuple[Optional[str], int]:
    """(name of the variable inside a miCOMPRESSED element, its inflated size), from the first bytes of its stream."""
    try:
        head = zlib.decompressobj().decompress(bytes(comp[:512]), 256)
        if len(head) < 64 or struct.unpack_from("<I", head, 0)[0] != MI_MATRIX:
            return None, 0
        return _matrix_header(head, 8, len(head))[3], 8 + struct.unpack_from("<I", head, 4)[0]
    except Exception:
        return None, 0


def _open_v5(path):
    """(memory mapping, size) of a little-endian level-5 file, or ``_Unsupported``."""
    with open(path, "rb") as fh:
        size = fh.seek(0, 2)
        if size < 136:
            raise _Unsupported("not a level-5 file")
        mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
    head = mm[:128]
    if head[:10] != b"MATLAB 5.0" or head[126:128] != b"IM":
        raise _Unsupported("not a little-endian level-5 file")
    return mm, size


def _find(mm, size: int, key: str, path):
    """Walk the top-level elements for the variable ``key``: ``("matrix", class, is_complex, dims, data offset, end)``
    for a plain miMATRIX element, ``("compressed", offset, byte count, inflated size)`` for a compressed one."""
    pos = 128
    unnamed = 0                                                 # elements whose name could not be read
    while pos + 8 <= size:
        t, n, d, nxt = _tag(mm, pos)
        if t == MI_COMPRESSED:
            nxt = d + n                                         # compressed elements are not padded
            name, inflated = _peek(memoryview(mm)[d:d + n])
            if name == key:
                return "compressed", d, n, inflated
            unnamed += name is None
        elif t == MI_MATRIX and n >= 48:
            cls, cplx, dims, name, data_pos = _matrix_header(mm, d, d + n)
            if name == key:
                return "matrix", cls, cplx, dims, data_pos, d + n
        elif t == MI_MATRIX:
            unnamed += 1                                        # too short for a header this reader knows (an empty array)
        if nxt <= pos:
            raise _Unsupported("corrupt element tag")
        pos = nxt
    if unnamed:                                                 # it may be one of those: scipy.io.loadmat decides
        raise _Unsupported(f"{unnamed} element(s) of {path} could not be identified")
    raise KeyError(f"{path} has no variable {key!r}")


def locate_variable_v5(path, key: str):
    """Where the variable ``key`` lies in a little-endian level-5 MAT file, without touching its data:
    ``(storage dtype, dims, byte offset of the real array, byte offset of the imaginary array or None)`` -- both
    arrays column-major.  ``_Unsupported`` for a compressed or non-float variable, ``KeyError`` if there is none."""
    mm, size = _open_v5(path)
    found = _find(mm, size, key, path)
    if found[0] != "matrix":
        raise _Unsupported("compressed variable")
    _, cls, cplx, dims, data_pos, _ = found
    if cls not in (MX_DOUBLE, MX_SINGLE):
        raise _Unsupported(f"array class {cls}")
    dt, _, d_re, d_im = _numeric_layout(mm, data_pos, dims, cplx)
    return dt, tuple(int(x) for x in dims), d_re, d_im


def read_variable_v5(path, key: str, pool: Optional[BufferPool] = None):
    """The variable ``key`` of a little-endian level-5 MAT file: ``(real, imag, how)`` with Fortran-ordered
    float32 / float64 arrays (``imag`` None for a real variable) that are views of the file's memory mapping
    (``how == "mapped"``), of its inflated bytes for a compressed variable (``"inflated"``), or -- with a ``pool`` --
    arrays from the pool that the data was read into with ``preadv`` (``"read"``; uncompressed variables only).
    Raises ``_Unsupported`` for what the module docstring lists, ``KeyError`` if there is no such variable."""
    mm, size = _open_v5(path)
    found = _find(mm, size, key, path)
    if found[0] == "compressed":
        _, d, n, inflated = found
        # one allocation of the final size instead of a buffer that doubles its way up (zlib releases the GIL:
        # run_extraction inflates the next variables on reader threads meanwhile)
        body = zlib.decompress(memoryview(mm)[d:d + n], zlib.MAX_WBITS, max(inflated, 1 << 16))
        t2, n2, d2, _ = _tag(body, 0)
        if t2 != MI_MATRIX:
            raise _Unsupported("compressed element is not a matrix")
        cls, cplx, dims, _, data_pos = _matrix_header(body, d2, d2 + n2)
        if cls not in (MX_DOUBLE, MX_SINGLE):
            raise _Unsupported(f"array class {cls}")
        return _numeric_parts(body, data_pos, d2 + n2, dims, cplx) + ("inflated",)
    _, cls, cplx, dims, data_pos, end = found
    if cls not in (MX_DOUBLE, MX_SINGLE):
        raise _Unsupported(f"array class {cls}")
    if pool is not None:                            # read it instead of mapping it
        dt, count, d_re, d_im = _numeric_layout(mm, data_pos, dims, cplx)
        real = pool.take(dt, count)
        imag = pool.take(dt, count) if d_im is not None else None
        jobs = [(real.view(np.uint8), d_re)] + ([(imag.view(np.uint8), d_im)] if imag is not None else [])
        try:
            _pread_into(path, jobs)
        except BaseException:
            pool.give(real)
            if imag is not None:
                pool.give(imag)
            raise
        return (real.reshape(dims, order="F"), None if imag is None else imag.reshape(dims, order="F"), "read")
    return _numeric_parts(mm, data_pos, end, dims, cplx) + ("mapped",)


def stores_compressed(mat_path) -> bool:
    """True if the first variable of a level-5 file is a miCOMPRESSED element (MATLAB's default ``save``): its
    variables have to be inflated -- worth several reader threads -- rather than read or mapped.  False for
    anything else, including files this reader does not take on."""
    try:
        with open(mat_path, "rb") as fh:
            head = fh.read(136)
        return len(head) == 136 and head[:10] == b"MATLAB 5.0" and head[126:128] == b"IM" and \
            struct.unpack_from("<I", head, 128)[0] == MI_COMPRESSED
    except OSError:
        return False


def compressed_variable_bytes(mat_path) -> int:
    """Inflated size of the first variable of a level-5 file if it is stored compressed, else 0: what one reader
    thread of ``run_extraction`` holds while it decodes ahead of the GPU."""
    if not stores_compressed(mat_path):
        return 0
    try:
        mm, size = _open_v5(mat_path)
        t, n, d, _ = _tag(mm, 128)
        return max(1, _peek(memoryview(mm)[d:d + n])[1])
    except Exception:
        return 1


_HDF5_MAGIC = b"\x89HDF\r\n\x1a\n"


def is_v73(mat_path) -> bool:
    """True for an HDF5 file: the signature at offset 0, or behind the 512-byte header a MATLAB -v7.3 file carries (the
    format allows a user block of 512, 1024, ... bytes; MATLAB writes 512)."""
    try:
        with open(mat_path, "rb") as fh:
            head = fh.read(520)
    except OSError:
        return False
    return head[:8] == _HDF5_MAGIC or (head[:10] == b"MATLAB 7.3" and head[512:520] == _HDF5_MAGIC)


def load_variable_v73(mat_path, key: str, direct: bool = False):
    """The variable ``key`` of a MATLAB -v7.3 (HDF5) file in MATLAB's shape: a :class:`FileComplex` when ``direct`` and
    the dataset is contiguous (nothing read: offsets into the file, interleaved complex or one real array, column-major),
    otherwise a Fortran-ordered complex / real ndarray decoded by libhdf5 -- what ``loadmat`` returns for a level-5 file
    of the same data.  ``KeyError``: no such variable; ``_Unsupported``: not a double / single numeric array, or no
    libhdf5 to read it with."""
    from . import hdf5_min
    from .feature_extraction import FileComplex
    if not hdf5_min.available():
        raise _Unsupported("no HDF5 C library")
    with hdf5_min.File(mat_path) as fh:
        if key not in fh:
            raise KeyError(f"{mat_path} has no variable {key!r}")
        try:
            ds = fh[key]
        except (TypeError, ValueError) as exc:              # a group (struct / cell), a compound that is not {real, imag}
            raise _Unsupported(str(exc))
        cls = ds.attr_string("MATLAB_class")
        want = {"double": 8, "single": 4}.get(cls)
        part = ds.dtype.itemsize // (2 if ds.complex_pair else 1)
        if want is None or ds.dtype.kind not in "fc" or part != want:
            raise _Unsupported(f"MATLAB_class {cls!r} stored as {ds.dtype}")
        dims = tuple(ds.shape[::-1])                         # MATLAB's shape
        if direct and len(dims) == 3 and ds.file_offset is not None and ds.little_endian and ds.chunks is None:
            if ds.complex_pair:
                return FileComplex(mat_path, ds.dtype, dims, ds.file_offset, interleaved=True, order="F")
            return FileComplex(mat_path, ds.dtype, dims, ds.file_offset, None)
        return ds[:].T                                       # (L, K, S) C-ordered -> (S, K, L) Fortran-ordered view


def load_variable(mat_path, key: str, pool: Optional[BufferPool] = None, direct: bool = False):
    """One variable of the container.  ``direct``: a :class:`FileComplex` -- offsets into the file, nothing read;
    the engine's staging threads read it (``source == "file"``; uncompressed variables only, anything else falls
    through to the forms below).  Otherwise a :class:`SplitComplex` over the memory-mapped file (its ``source``
    says ``"mapped"``; ``"inflated"`` for a compressed variable; ``"read"`` when a ``pool`` was given and the data
    was read into its buffers -- call ``release()`` on the result when done with it) when the fast reader applies,
    otherwise what ``scipy.io.loadmat`` returns for it."""
    from .feature_extraction import FileComplex, SplitComplex
    if is_v73(mat_path):
        try:
            return load_variable_v73(mat_path, key, direct)
        except _Unsupported:
            import scipy.io                                     # its own words for a file / class it does not read
            data = scipy.io.loadmat(str(mat_path), variable_names=[key])
            return np.asarray(data[key])
    if direct:
        try:
            dt, dims, d_re, d_im = locate_variable_v5(Path(mat_path), key)
            if len(dims) == 3:
                return FileComplex(mat_path, dt, dims, d_re, d_im)
        except (_Unsupported, struct.error, ValueError, zlib.error):
            pass
    try:
        real, imag, how = read_variable_v5(Path(mat_path), key, pool)
        out = SplitComplex(real, imag)
        out.source = how
        if how == "read":
            def release(real=real, imag=imag):
                pool.give(real.reshape(-1, order="F"))
                if imag is not None:
                    pool.give(imag.reshape(-1, order="F"))
            out.release = release
        return out
    except _Unsupported:
        pass
    except (struct.error, ValueError, zlib.error):             # a malformed file: let scipy name the problem
        pass
    import scipy.io
    data = scipy.io.loadmat(str(mat_path), variable_names=[key])
    if key not in data:
        raise KeyError(f"{mat_path} has no variable {key!r}")
    return np.asarray(data[key])
