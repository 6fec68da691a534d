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
4 or 7.3 (HDF5) files -- goes to ``scipy.io.loadmat``, whose result for the variable is returned
as it is.
"""
from __future__ import annotations

import mmap
import struct
import zlib
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


def _numeric_parts(buf, pos: int, end: int, dims, is_complex: bool):
    t, n, d, nxt = _tag(buf, pos)
    count = int(np.prod(dims, dtype=np.int64))
    if t not in _STORAGE or n != count * _STORAGE[t].itemsize:
        raise _Unsupported(f"numeric data stored as MAT type {t}")
    dt = _STORAGE[t]
    real = np.frombuffer(buf, dtype=dt, count=count, offset=d).reshape(dims, order="F")
    imag = None
    if is_complex:
        t2, n2, d2, _ = _tag(buf, nxt)
        if t2 != t or n2 != n:
            raise _Unsupported("imaginary part stored differently from the real part")
        imag = np.frombuffer(buf, dtype=dt, count=count, offset=d2).reshape(dims, order="F")
    return real, imag


def _peek(comp: memoryview) -> Tuple[Optional[str], int]:
    """(name of the variable inside a miCOMPRESSED element, its inflated size), from the first bytes of its stream."""
    try:
        head = zlib.decompressobj().decompress(bytes(comp[:512]), 256)
        if len(head) < 64 or struct.unpack_from("<I", head, 0)[0] != MI_MATRIX:
            return None, 0
        return _matrix_header(head, 8, len(head))[3], 8 + struct.unpack_from("<I", head, 4)[0]
    except Exception:
        return None, 0


def read_variable_v5(path, key: str):
    """The variable ``key`` of a little-endian level-5 MAT file: ``(real, imag, how)`` with Fortran-ordered
    float32 / float64 arrays (``imag`` None for a real variable) that are views of the file's memory mapping
    (``how == "mapped"``) or, for a compressed variable, of its inflated bytes (``"inflated"``).
    Raises ``_Unsupported`` for what the module docstring lists, ``KeyError`` if there is no such variable."""
    with open(path, "rb") as fh:
        size = fh.seek(0, 2)
        if size < 136:
            raise _Unsupported("not a level-5 file")
        mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
    head = mm[:128]
    if head[:10] != b"MATLAB 5.0" or head[126:128] != b"IM":
        raise _Unsupported("not a little-endian level-5 file")
    pos = 128
    while pos + 8 <= size:
        t, n, d, nxt = _tag(mm, pos)
        if t == MI_COMPRESSED:
            nxt = d + n                                         # compressed elements are not padded
            name, inflated = _peek(memoryview(mm)[d:d + n])
            if name == key:
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
        elif t == MI_MATRIX and n >= 48:
            cls, cplx, dims, name, data_pos = _matrix_header(mm, d, d + n)
            if name == key:
                if cls not in (MX_DOUBLE, MX_SINGLE):
                    raise _Unsupported(f"array class {cls}")
                return _numeric_parts(mm, data_pos, d + n, dims, cplx) + ("mapped",)
        if nxt <= pos:
            raise _Unsupported("corrupt element tag")
        pos = nxt
    raise KeyError(f"{path} has no variable {key!r}")


def load_variable(mat_path, key: str):
    """One variable of the container: a :class:`SplitComplex` over the memory-mapped file (its ``source``
    says ``"mapped"``, or ``"inflated"`` for a compressed variable) when the fast reader applies, otherwise
    what ``scipy.io.loadmat`` returns for it."""
    from .feature_extraction import SplitComplex
    try:
        real, imag, how = read_variable_v5(Path(mat_path), key)
        out = SplitComplex(real, imag)
        out.source = how
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
