"""The few calls of the HDF5 C library this path needs, through ctypes: open a file read-only, open a
dataset, ask for its shape and element type, read a range of rows.  For RadioML-style containers
(``GOLD_XYZ_OSC.0001_1024.hdf5``: ``X`` float32 (2 555 904, 1024, 2); reference old/dataset.py:43-56,
old/dataset_analysis.py:17-22) where ``h5py`` -- which the reference's legacy scripts import -- is not installed but the
C library is (this image: /opt/conda/lib/libhdf5.so.103, HDF5 1.10.6).  libhdf5 does the decoding, so contiguous,
chunked and deflate-compressed datasets all read the same way.

    with hdf5_min.File(path) as fh:
        x = fh["X"]                  # .shape, .dtype, x[a:b] -> numpy array of rows a .. b-1
        feats = extract_iq_pairs(x)

Not a general binding: integer and IEEE float element types, plus the one compound MATLAB's ``-v7.3`` files use for
complex arrays -- ``{real, imag}`` of two equal floats, read as complex64 / complex128 (amcpy_amd/matfile.py) --, whole
rows along axis 0, string attributes (``MATLAB_class``).  `create_dataset` exists for the tests, which need a genuine
HDF5 file to read back.

The library is found through AMCX_LIBHDF5 (a path), ctypes.util.find_library("hdf5"), then the usual install
locations; `available()` says whether one loaded.  Builds of libhdf5 are usually NOT thread-safe: every call into it
is made under one process-wide lock (ctypes drops the GIL around foreign calls, and `extract_iq_pairs` slices a
dataset from reader threads)."""
from __future__ import annotations

import ctypes
import ctypes.util
import glob
import os
import threading
from typing import Optional, Tuple

import numpy as np

_LOCK = threading.RLock()
_LIB = None
_LIB_PATH: Optional[str] = None
_TRIED = False

hid_t = ctypes.c_int64          # HDF5 >= 1.10
hsize_t = ctypes.c_uint64
herr_t = ctypes.c_int

H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5P_DEFAULT, H5S_ALL = 0, 0
H5S_SELECT_SET = 0
H5T_INTEGER, H5T_FLOAT, H5T_STRING, H5T_COMPOUND = 0, 1, 3, 6
H5T_SGN_NONE = 0
H5T_ORDER_LE, H5T_ORDER_BE = 0, 1
H5D_CHUNKED = 2
H5Z_FILTER_DEFLATE, H5Z_FILTER_SHUFFLE = 1, 2


def _candidates():
    env = os.environ.get("AMCX_LIBHDF5")
    if env:
        yield env
    found = ctypes.util.find_library("hdf5") or ctypes.util.find_library("hdf5_serial")
    if found:
        yield found
    for pat in ("/usr/lib/x86_64-linux-gnu/libhdf5_serial.so*", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*",
                "/usr/lib64/libhdf5.so*", "/usr/local/lib/libhdf5.so*", "/opt/conda/lib/libhdf5.so*"):
        for p in sorted(glob.glob(pat)):
            yield p


def _load():
    global _LIB, _LIB_PATH, _TRIED
    with _LOCK:
        if _TRIED:
            return _LIB
        _TRIED = True
        for cand in _candidates():
            try:
                lib = ctypes.CDLL(cand)
                major, minor, rel = ctypes.c_uint(), ctypes.c_uint(), ctypes.c_uint()
                lib.H5get_libversion.restype = herr_t
                if lib.H5get_libversion(ctypes.byref(major), ctypes.byref(minor), ctypes.byref(rel)) < 0:
                    continue
                if (major.value, minor.value) < (1, 10):        # hid_t was 32 bits before 1.10
                    continue
                _declare(lib)
                if lib.H5open() < 0:
                    continue
                # the library prints an error stack to stderr by default; failures are raised here instead
                lib.H5Eset_auto2(hid_t(0), None, None)
                _LIB, _LIB_PATH = lib, cand
                break
            except (OSError, AttributeError):
                continue
        return _LIB


def _declare(lib):
    P = ctypes.POINTER
    sigs = {
        "H5open": (herr_t, []),
        "H5Eset_auto2": (herr_t, [hid_t, ctypes.c_void_p, ctypes.c_void_p]),
        "H5Fopen": (hid_t, [ctypes.c_char_p, ctypes.c_uint, hid_t]),
        "H5Fcreate": (hid_t, [ctypes.c_char_p, ctypes.c_uint, hid_t, hid_t]),
        "H5Fclose": (herr_t, [hid_t]),
        "H5Lexists": (ctypes.c_int, [hid_t, ctypes.c_char_p, hid_t]),
        "H5Dopen2": (hid_t, [hid_t, ctypes.c_char_p, hid_t]),
        "H5Dcreate2": (hid_t, [hid_t, ctypes.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
        "H5Dclose": (herr_t, [hid_t]),
        "H5Dget_space": (hid_t, [hid_t]),
        "H5Dget_type": (hid_t, [hid_t]),
        "H5Dget_offset": (ctypes.c_uint64, [hid_t]),
        "H5Dget_create_plist": (hid_t, [hid_t]),
        "H5Pget_layout": (ctypes.c_int, [hid_t]),
        "H5Pget_chunk": (ctypes.c_int, [hid_t, ctypes.c_int, P(hsize_t)]),
        "H5Pget_nfilters": (ctypes.c_int, [hid_t]),
        "H5Pget_filter2": (ctypes.c_int, [hid_t, ctypes.c_uint, P(ctypes.c_uint), P(ctypes.c_size_t), P(ctypes.c_uint),
                                           ctypes.c_size_t, ctypes.c_char_p, P(ctypes.c_uint)]),
        "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, ctypes.c_void_p]),
        "H5Dwrite": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, ctypes.c_void_p]),
        "H5Sclose": (herr_t, [hid_t]),
        "H5Screate_simple": (hid_t, [ctypes.c_int, P(hsize_t), P(hsize_t)]),
        "H5Sget_simple_extent_ndims": (ctypes.c_int, [hid_t]),
        "H5Sget_simple_extent_dims": (ctypes.c_int, [hid_t, P(hsize_t), P(hsize_t)]),
        "H5Sselect_hyperslab": (herr_t, [hid_t, ctypes.c_int, P(hsize_t), P(hsize_t), P(hsize_t), P(hsize_t)]),
        "H5Tclose": (herr_t, [hid_t]),
        "H5Tget_class": (ctypes.c_int, [hid_t]),
        "H5Tget_size": (ctypes.c_size_t, [hid_t]),
        "H5Tget_sign": (ctypes.c_int, [hid_t]),
        "H5Tget_order": (ctypes.c_int, [hid_t]),
        "H5Tcreate": (hid_t, [ctypes.c_int, ctypes.c_size_t]),
        "H5Tinsert": (herr_t, [hid_t, ctypes.c_char_p, ctypes.c_size_t, hid_t]),
        "H5Tget_nmembers": (ctypes.c_int, [hid_t]),
        "H5Tget_member_name": (ctypes.c_void_p, [hid_t, ctypes.c_uint]),
        "H5Tget_member_type": (hid_t, [hid_t, ctypes.c_uint]),
        "H5Tget_member_offset": (ctypes.c_size_t, [hid_t, ctypes.c_uint]),
        "H5Tis_variable_str": (ctypes.c_int, [hid_t]),
        "H5free_memory": (herr_t, [ctypes.c_void_p]),
        "H5Aexists": (ctypes.c_int, [hid_t, ctypes.c_char_p]),
        "H5Aopen": (hid_t, [hid_t, ctypes.c_char_p, hid_t]),
        "H5Aget_type": (hid_t, [hid_t]),
        "H5Aread": (herr_t, [hid_t, hid_t, ctypes.c_void_p]),
        "H5Aclose": (herr_t, [hid_t]),
        "H5Tcopy": (hid_t, [hid_t]),
        "H5Tset_size": (herr_t, [hid_t, ctypes.c_size_t]),
        "H5Acreate2": (hid_t, [hid_t, ctypes.c_char_p, hid_t, hid_t, hid_t, hid_t]),
        "H5Awrite": (herr_t, [hid_t, hid_t, ctypes.c_void_p]),
        "H5Screate": (hid_t, [ctypes.c_int]),
        "H5Pset_userblock": (herr_t, [hid_t, hsize_t]),
        "H5Pcreate": (hid_t, [hid_t]),
        "H5Pclose": (herr_t, [hid_t]),
        "H5Pset_chunk": (herr_t, [hid_t, ctypes.c_int, P(hsize_t)]),
        "H5Pset_deflate": (herr_t, [hid_t, ctypes.c_uint]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    # raw chunk access (1.10.3+): optional -- without it chunked datasets are read through H5Dread only
    optional = {
        "H5Dget_chunk_storage_size": (herr_t, [hid_t, P(hsize_t), P(hsize_t)]),
        "H5Dread_chunk": (herr_t, [hid_t, hid_t, P(hsize_t), P(ctypes.c_uint32), ctypes.c_void_p]),
    }
    lib._amcx_raw_chunks = True
    for name, (res, args) in optional.items():
        fn = getattr(lib, name, None)
        if fn is None:
            lib._amcx_raw_chunks = False
        else:
            fn.restype, fn.argtypes = res, args


def write_matlab_header(path, text: str = "MATLAB 7.3 MAT-file, Platform: GLNXA64, Created by amcpy_amd.hdf5_min HDF5 schema 1.00 .") -> None:
    """Fill the 512-byte user block of a file created with ``File(path, "w", userblock=512)`` with MATLAB's -v7.3 header:
    116 bytes of text, 8 of subsystem offset, version 0x0200 and the endian indicator "IM"."""
    block = bytearray(b" " * 116 + b"\0" * 8 + bytes([0x00, 0x02]) + b"IM" + b"\0" * 384)
    head = text.encode("ascii")[:116]
    block[:len(head)] = head
    with open(path, "r+b") as fh:
        fh.write(bytes(block))


def available() -> bool:
    """True when an HDF5 C library (>= 1.10) could be loaded."""
    return _load() is not None


def library_path() -> Optional[str]:
    _load()
    return _LIB_PATH


def _lib():
    lib = _load()
    if lib is None:
        raise ImportError("no HDF5 C library (libhdf5 >= 1.10) found: set AMCX_LIBHDF5 to its path, or install h5py")
    return lib


def _native_type(lib, dtype: np.dtype) -> int:
    """The library's predefined native type for a numpy dtype (the H5T_NATIVE_* macros are global variables)."""
    names = {"f4": "H5T_NATIVE_FLOAT_g", "f8": "H5T_NATIVE_DOUBLE_g", "i1": "H5T_NATIVE_SCHAR_g", "u1": "H5T_NATIVE_UCHAR_g",
             "i2": "H5T_NATIVE_SHORT_g", "u2": "H5T_NATIVE_USHORT_g", "i4": "H5T_NATIVE_INT_g", "u4": "H5T_NATIVE_UINT_g",
             "i8": "H5T_NATIVE_LLONG_g", "u8": "H5T_NATIVE_ULLONG_g"}
    key = np.dtype(dtype).newbyteorder("=").str[1:]
    if key not in names:
        raise TypeError(f"element type {dtype} is not handled by hdf5_min")
    return hid_t.in_dll(lib, names[key]).value


def _dims(seq) -> "ctypes.Array":
    return (hsize_t * len(seq))(*[int(v) for v in seq])


class Dataset:
    """A dataset of a simple element type: ``shape``, ``dtype`` and row slices ``ds[a:b]`` (axis 0, step 1), each read
    with one H5Dread of a hyperslab into a fresh C-ordered numpy array."""

    def __init__(self, file: "File", name: str):
        lib = _lib()
        self._file, self.name = file, name
        with _LOCK:
            self._id = lib.H5Dopen2(file._id, name.encode(), H5P_DEFAULT)
            if self._id < 0:
                raise KeyError(f"{file.path}: no dataset {name!r}")
            try:
                self._describe(lib, name)
            except Exception:
                lib.H5Dclose(self._id)            # a dataset this binding does not handle: no handle is left behind
                self._id = -1
                raise

    def _describe(self, lib, name: str) -> None:
        space = lib.H5Dget_space(self._id)
        try:
            rank = lib.H5Sget_simple_extent_ndims(space)
            if rank < 1:
                raise ValueError(f"{name!r} is not a simple array (rank {rank})")
            dims = (hsize_t * rank)()
            lib.H5Sget_simple_extent_dims(space, dims, None)
            self.shape: Tuple[int, ...] = tuple(int(d) for d in dims)
        finally:
            lib.H5Sclose(space)
        tid = lib.H5Dget_type(self._id)
        try:
            cls, size = lib.H5Tget_class(tid), int(lib.H5Tget_size(tid))
            self.complex_pair = False
            if cls == H5T_FLOAT and size in (4, 8):
                self.dtype = np.dtype(f"f{size}")
                self.little_endian = lib.H5Tget_order(tid) == H5T_ORDER_LE
            elif cls == H5T_INTEGER and size in (1, 2, 4, 8):
                self.dtype = np.dtype(("u" if lib.H5Tget_sign(tid) == H5T_SGN_NONE else "i") + str(size))
                self.little_endian = lib.H5Tget_order(tid) == H5T_ORDER_LE
            elif cls == H5T_COMPOUND:
                self._describe_pair(lib, tid, size, name)
            else:
                raise TypeError(f"{name!r}: element class {cls} of {size} bytes is not handled by hdf5_min")
        finally:
            lib.H5Tclose(tid)
        # a CONTIGUOUS dataset's bytes lie in one run of the file (no chunks, no filters): where, or None.  The engine's
        # staging threads can then pread them straight into their pinned slots, past libhdf5 (extract_iq_pairs).
        off = int(lib.H5Dget_offset(self._id))
        self.file_offset: Optional[int] = None if off == 0xFFFFFFFFFFFFFFFF else off
        self.file_path = self._file.path
        # chunk shape and filter pipeline (ids in the order they were applied on write)
        self.chunks: Optional[Tuple[int, ...]] = None
        self.filters: Tuple[int, ...] = ()
        dcpl = lib.H5Dget_create_plist(self._id)
        if dcpl >= 0:
            try:
                if lib.H5Pget_layout(dcpl) == H5D_CHUNKED:
                    cd = (hsize_t * len(self.shape))()
                    if lib.H5Pget_chunk(dcpl, len(self.shape), cd) >= 0:
                        self.chunks = tuple(int(v) for v in cd)
                    ids = []
                    for i in range(max(0, lib.H5Pget_nfilters(dcpl))):
                        flags, cfg = ctypes.c_uint(), ctypes.c_uint()
                        n_cd = ctypes.c_size_t(0)
                        ids.append(int(lib.H5Pget_filter2(dcpl, i, ctypes.byref(flags), ctypes.byref(n_cd), None, 0, None,
                                                          ctypes.byref(cfg))))
                    self.filters = tuple(ids)
            finally:
                lib.H5Pclose(dcpl)

    def _describe_pair(self, lib, tid: int, size: int, name: str) -> None:
        """The compound MATLAB -v7.3 stores a complex array as: members ``real`` at 0 and ``imag`` behind it, two equal
        IEEE floats, nothing else -- element bytes = one interleaved complex number."""
        members = []
        for i in range(max(0, lib.H5Tget_nmembers(tid))):
            raw = lib.H5Tget_member_name(tid, i)
            mname = ctypes.string_at(raw).decode("latin-1") if raw else ""
            if raw:
                lib.H5free_memory(raw)
            mt = lib.H5Tget_member_type(tid, i)
            try:
                members.append((mname, int(lib.H5Tget_member_offset(tid, i)), lib.H5Tget_class(mt), int(lib.H5Tget_size(mt)),
                                lib.H5Tget_order(mt) == H5T_ORDER_LE))
            finally:
                lib.H5Tclose(mt)
        ok = (len(members) == 2 and [m[0] for m in members] == ["real", "imag"] and all(m[2] == H5T_FLOAT for m in members)
              and members[0][3] == members[1][3] and members[0][3] in (4, 8)
              and members[0][1] == 0 and members[1][1] == members[0][3] and size == 2 * members[0][3])
        if not ok:
            raise TypeError(f"{name!r}: a compound other than MATLAB's {{real, imag}} pair is not handled by hdf5_min: {members}")
        self.complex_pair = True
        self.dtype = np.dtype(f"c{size}")
        self.little_endian = members[0][4] and members[1][4]

    def _memory_type(self, lib):
        """(type id for H5Dread into ``self.dtype``, whether it must be closed)."""
        if not self.complex_pair:
            return _native_type(lib, self.dtype), False
        part = np.dtype(f"f{self.dtype.itemsize // 2}")
        tid = lib.H5Tcreate(H5T_COMPOUND, self.dtype.itemsize)
        lib.H5Tinsert(tid, b"real", 0, _native_type(lib, part))
        lib.H5Tinsert(tid, b"imag", part.itemsize, _native_type(lib, part))
        return tid, True

    def attr_string(self, name: str) -> Optional[str]:
        """A fixed-length string attribute of the dataset (``MATLAB_class``), or None if there is none of that kind."""
        lib = _lib()
        with _LOCK:
            if self._id < 0:
                raise ValueError("dataset of a closed file")
            if lib.H5Aexists(self._id, name.encode()) <= 0:
                return None
            aid = lib.H5Aopen(self._id, name.encode(), H5P_DEFAULT)
            if aid < 0:
                return None
            tid = lib.H5Aget_type(aid)
            try:
                if lib.H5Tget_class(tid) != H5T_STRING or lib.H5Tis_variable_str(tid) > 0:
                    return None
                n = int(lib.H5Tget_size(tid))
                buf = ctypes.create_string_buffer(n + 1)
                if lib.H5Aread(aid, tid, buf) < 0:
                    return None
                return buf.raw[:n].split(b"\0", 1)[0].decode("latin-1").strip()
            finally:
                lib.H5Tclose(tid)
                lib.H5Aclose(aid)

    @property
    def _parallel_chunks(self) -> bool:
        """Chunks whose filters are at most shuffle + deflate, little-endian elements: the raw chunks can be fetched from
        the library (cheap, under the lock) and inflated on several threads outside it.  (MATLAB's -v7.3 default is
        chunked + deflate with a chunk shape of its own choosing; RadioML's is chunks of whole rows.)"""
        lib = _lib()
        return (getattr(lib, "_amcx_raw_chunks", False) and self.chunks is not None and self.little_endian
                and len(self.filters) > 0
                and self.filters in ((H5Z_FILTER_DEFLATE,), (H5Z_FILTER_SHUFFLE, H5Z_FILTER_DEFLATE)))

    def _read_chunks(self, lo: int, hi: int, out: np.ndarray) -> None:
        """Rows lo .. hi-1 into ``out`` chunk by chunk: H5Dread_chunk hands over a chunk as it is stored, zlib inflates it and
        the byte shuffle is undone here, on a few threads (zlib and numpy release the GIL) -- libhdf5's own H5Dread inflates
        on the calling thread, under this module's lock.  Any chunk grid: a chunk lands in the block of ``out`` it covers."""
        import itertools
        import zlib
        from concurrent.futures import ThreadPoolExecutor
        lib = _lib()
        cshape, item = self.chunks, self.dtype.itemsize
        chunk_bytes = int(np.prod(cshape, dtype=np.int64)) * item

        def one(index) -> None:
            origin = tuple(i * c for i, c in zip(index, cshape))
            coord = _dims(origin)
            # the part of the chunk inside the dataset and inside [lo, hi)
            begin = (max(lo, origin[0]),) + origin[1:]
            end = (min(hi, origin[0] + cshape[0]),) + tuple(min(n, o + c) for n, o, c in zip(self.shape[1:], origin[1:], cshape[1:]))
            dst = (slice(begin[0] - lo, end[0] - lo),) + tuple(slice(a, b) for a, b in zip(begin[1:], end[1:]))
            with _LOCK:
                if self._id < 0:
                    raise ValueError("dataset of a closed file")
                lib.H5Eset_auto2(hid_t(0), None, None)       # the error stack's printer is per thread: quiet on this one too
                stored = hsize_t(0)
                if lib.H5Dget_chunk_storage_size(self._id, coord, ctypes.byref(stored)) < 0 or stored.value == 0:
                    # a chunk that was never written has no storage (libhdf5 1.10.6 reports that as an error): its elements
                    # are the dataset's FILL VALUE, which only H5Dread knows
                    out[dst] = self._read_hyperslab(begin, tuple(b - a for a, b in zip(begin, end)))
                    return
                raw = ctypes.create_string_buffer(int(stored.value))
                mask = ctypes.c_uint32(0)
                if lib.H5Dread_chunk(self._id, H5P_DEFAULT, coord, ctypes.byref(mask), raw) < 0:
                    raise OSError(f"{self._file.path}: reading chunk {index} of {self.name!r} failed")
                mask = mask.value
            data = raw.raw
            for pos in range(len(self.filters) - 1, -1, -1):  # undo the pipeline back to front; bit `pos` set: skipped
                if mask & (1 << pos):
                    continue
                if self.filters[pos] == H5Z_FILTER_DEFLATE:
                    # (the output size is known: without it zlib grows its buffer step by step, re-taking the GIL each
                    #  time, and eight threads are no faster than one)
                    data = zlib.decompress(data, bufsize=chunk_bytes)
                else:                                         # shuffle: byte j of every element was stored together
                    n = len(data) // item
                    data = np.frombuffer(data, dtype=np.uint8, count=n * item).reshape(item, n).T.tobytes()
            if len(data) != chunk_bytes:
                raise OSError(f"{self.name!r}: chunk {index} decodes to {len(data)} bytes, expected {chunk_bytes}")
            block = np.frombuffer(data, dtype=self.dtype).reshape(cshape)
            src = tuple(slice(a - o, b - o) for a, b, o in zip(begin, end, origin))
            out[dst] = block[src]

        grid = [range(lo // cshape[0], (hi - 1) // cshape[0] + 1)] + \
               [range((n + c - 1) // c) for n, c in zip(self.shape[1:], cshape[1:])]
        todo = list(itertools.product(*grid))
        try:
            cores = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            cores = os.cpu_count() or 2
        workers = min(len(todo), max(1, min(16, cores)))      # inflate-bound: zlib releases the GIL
        if workers <= 1:
            for index in todo:
                one(index)
        else:
            with ThreadPoolExecutor(max_workers=workers, thread_name_prefix="amcx-h5") as ex:
                list(ex.map(one, todo))

    def __len__(self) -> int:
        return self.shape[0]

    @property
    def ndim(self) -> int:
        return len(self.shape)

    def __getitem__(self, key) -> np.ndarray:
        if key is Ellipsis:
            key = slice(None)
        if isinstance(key, (int, np.integer)):
            k = int(key) + (self.shape[0] if key < 0 else 0)
            if not 0 <= k < self.shape[0]:
                raise IndexError(f"row {int(key)} of a dataset of {self.shape[0]}")
            return self[k:k + 1][0]
        if not isinstance(key, slice):
            raise TypeError("hdf5_min.Dataset reads row ranges: ds[a:b]")
        lo, hi, step = key.indices(self.shape[0])
        if step != 1:
            raise ValueError("hdf5_min.Dataset reads contiguous row ranges (step 1)")
        count = max(0, hi - lo)
        out = np.empty((count,) + self.shape[1:], dtype=self.dtype)
        if out.size == 0:
            return out
        if self._parallel_chunks:
            self._read_chunks(lo, hi, out)
            return out
        return self._read_rows(lo, hi, out)

    def _read_rows(self, lo: int, hi: int, out: Optional[np.ndarray] = None) -> np.ndarray:
        """Rows lo .. hi-1 with one H5Dread (the library decodes: any layout, any filter it has, the fill value where
        nothing was written)."""
        return self._read_hyperslab((lo,) + (0,) * (len(self.shape) - 1), (hi - lo,) + self.shape[1:], out)

    def _read_hyperslab(self, start, count, out: Optional[np.ndarray] = None) -> np.ndarray:
        if out is None:
            out = np.empty(tuple(count), dtype=self.dtype)
        lib = _lib()
        with _LOCK:
            if self._id < 0:
                raise ValueError("dataset of a closed file")
            fspace = lib.H5Dget_space(self._id)
            mspace = lib.H5Screate_simple(len(out.shape), _dims(out.shape), None)
            mtype, close_mtype = self._memory_type(lib)
            try:
                if lib.H5Sselect_hyperslab(fspace, H5S_SELECT_SET, _dims(start), None, _dims(out.shape), None) < 0:
                    raise OSError(f"{self.name!r}: selecting {tuple(count)} at {tuple(start)} failed")
                if lib.H5Dread(self._id, mtype, mspace, fspace, H5P_DEFAULT,
                               out.ctypes.data_as(ctypes.c_void_p)) < 0:
                    raise OSError(f"{self._file.path}: reading {tuple(count)} at {tuple(start)} of {self.name!r} failed "
                                  "(file cut short, or a filter this libhdf5 lacks)")
            finally:
                if close_mtype:
                    lib.H5Tclose(mtype)
                lib.H5Sclose(mspace)
                lib.H5Sclose(fspace)
        return out

    def _close(self):
        with _LOCK:
            if self._id >= 0:
                _lib().H5Dclose(self._id)
                self._id = -1


class File:
    """An HDF5 file: read-only by default (``mode="w"`` truncates; the tests write the file they read)."""

    def __init__(self, path, mode: str = "r", userblock: int = 0):
        """``userblock`` (mode "w"): bytes reserved in front of the HDF5 superblock (512, 1024, ...: MATLAB's -v7.3 header
        lives there; :func:`write_matlab_header` fills it after ``close()``)."""
        lib = _lib()
        self.path = str(path)
        self._sets = []
        with _LOCK:
            if mode == "r":
                if not os.path.exists(self.path):
                    raise FileNotFoundError(self.path)
                self._id = lib.H5Fopen(self.path.encode(), H5F_ACC_RDONLY, H5P_DEFAULT)
            elif mode == "w":
                fcpl = H5P_DEFAULT
                if userblock:
                    fcpl = lib.H5Pcreate(hid_t.in_dll(lib, "H5P_CLS_FILE_CREATE_ID_g").value)
                    lib.H5Pset_userblock(fcpl, int(userblock))
                self._id = lib.H5Fcreate(self.path.encode(), H5F_ACC_TRUNC, fcpl, H5P_DEFAULT)
                if fcpl != H5P_DEFAULT:
                    lib.H5Pclose(fcpl)
            else:
                raise ValueError("mode is 'r' or 'w'")
            if self._id < 0:
                raise OSError(f"{self.path}: not an HDF5 file this library can open")

    def __contains__(self, name: str) -> bool:
        with _LOCK:
            return _lib().H5Lexists(self._id, name.encode(), H5P_DEFAULT) > 0

    def __getitem__(self, name: str) -> Dataset:
        if name not in self:
            raise KeyError(f"{self.path}: no dataset {name!r}")
        ds = Dataset(self, name)
        self._sets.append(ds)
        return ds

    def create_dataset(self, name: str, data: np.ndarray, chunks: Optional[Tuple[int, ...]] = None,
                       deflate: Optional[int] = None, matlab_class: Optional[str] = None) -> None:
        """Write ``data`` as dataset ``name`` -- contiguous, or chunked (and deflate-compressed) when ``chunks`` is given.
        complex64 / complex128 data goes out as the compound ``{real, imag}`` MATLAB's -v7.3 files use; ``matlab_class``
        adds the ``MATLAB_class`` string attribute.  (``data`` is written in ITS shape: a MATLAB (S, K, L) variable is the
        dataset of ``np.asfortranarray(x).T``'s shape -- :func:`write_mat73_variable` does that.)"""
        lib = _lib()
        data = np.ascontiguousarray(data)
        with _LOCK:
            space = lib.H5Screate_simple(data.ndim, _dims(data.shape), None)
            dcpl = H5P_DEFAULT
            if chunks is not None:
                dcpl = lib.H5Pcreate(hid_t.in_dll(lib, "H5P_CLS_DATASET_CREATE_ID_g").value)
                lib.H5Pset_chunk(dcpl, data.ndim, _dims(chunks))
                if deflate is not None:
                    lib.H5Pset_deflate(dcpl, int(deflate))
            close_tid = False
            if data.dtype.kind == "c":
                part = np.dtype(f"f{data.dtype.itemsize // 2}")
                tid = lib.H5Tcreate(H5T_COMPOUND, data.dtype.itemsize)
                lib.H5Tinsert(tid, b"real", 0, _native_type(lib, part))
                lib.H5Tinsert(tid, b"imag", part.itemsize, _native_type(lib, part))
                close_tid = True
            else:
                tid = _native_type(lib, data.dtype)
            dset = lib.H5Dcreate2(self._id, name.encode(), tid, space, H5P_DEFAULT, dcpl, H5P_DEFAULT)
            try:
                if dset < 0 or lib.H5Dwrite(dset, tid, H5S_ALL, H5S_ALL, H5P_DEFAULT, data.ctypes.data_as(ctypes.c_void_p)) < 0:
                    raise OSError(f"{self.path}: writing {name!r} failed")
                if matlab_class is not None:
                    text = matlab_class.encode("ascii")
                    st = lib.H5Tcopy(hid_t.in_dll(lib, "H5T_C_S1_g").value)
                    lib.H5Tset_size(st, len(text))
                    sp = lib.H5Screate(0)                       # H5S_SCALAR
                    at = lib.H5Acreate2(dset, b"MATLAB_class", st, sp, H5P_DEFAULT, H5P_DEFAULT)
                    try:
                        if at < 0 or lib.H5Awrite(at, st, ctypes.c_char_p(text)) < 0:
                            raise OSError(f"{self.path}: writing the MATLAB_class attribute of {name!r} failed")
                    finally:
                        if at >= 0:
                            lib.H5Aclose(at)
                        lib.H5Sclose(sp)
                        lib.H5Tclose(st)
            finally:
                if dset >= 0:
                    lib.H5Dclose(dset)
                if close_tid:
                    lib.H5Tclose(tid)
                if dcpl != H5P_DEFAULT:
                    lib.H5Pclose(dcpl)
                lib.H5Sclose(space)

    def write_mat73_variable(self, name: str, x: np.ndarray, chunks: Optional[Tuple[int, ...]] = None,
                             deflate: Optional[int] = None) -> None:
        """A MATLAB variable the way ``save -v7.3`` lays it out: the dataset has the REVERSED shape (its bytes are the
        column-major variable), complex as ``{real, imag}``, ``MATLAB_class`` double / single."""
        cls = {"f8": "double", "c16": "double", "f4": "single", "c8": "single"}[np.dtype(x.dtype).str[1:]]
        self.create_dataset(name, np.ascontiguousarray(np.asarray(x).T), chunks=chunks, deflate=deflate, matlab_class=cls)

    def close(self):
        with _LOCK:
            for ds in self._sets:
                ds._close()
            self._sets = []
            if self._id >= 0:
                _lib().H5Fclose(self._id)
                self._id = -1

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
