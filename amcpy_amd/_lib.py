"""ctypes binding of libamcx.so (C ABI: include/amcx.h).

The HIP library is the product: if it cannot be loaded this module raises --
there is no CPU fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("AMCX_LIB", _HERE / "lib" / "libamcx.so"))

ABI_VERSION = 6          # the version this binding was written against; any library >= it will do (include/amcx.h)
NUM_FEATURES = 18
VARIANT_AUTO, VARIANT_BLOCK, VARIANT_WAVE = 0, 1, 2
VARIANTS = {"auto": VARIANT_AUTO, "block": VARIANT_BLOCK, "wave": VARIANT_WAVE}
OK, EINVAL, ENOTSUP, EHIP, ENODEV, ENOMEM, EIO = 0, -1, -2, -3, -4, -5, -6
SRC_C64, SRC_C128, SRC_F32_SPLIT, SRC_F64_SPLIT = 0, 1, 2, 3


class UploadStats(C.Structure):
    """amcx_upload_stats (include/amcx.h)."""
    _fields_ = [("frames", C.c_int64), ("source_bytes", C.c_int64), ("pcie_bytes", C.c_int64),
                ("chunks", C.c_int32), ("threads", C.c_int32), ("plane_major", C.c_int32), ("from_file", C.c_int32),
                ("seconds", C.c_double), ("seconds_staging", C.c_double), ("seconds_waiting", C.c_double),
                ("seconds_prepare", C.c_double), ("seconds_tail", C.c_double)]


class Placement(C.Structure):
    """amcx_placement (include/amcx.h)."""
    _fields_ = [("device", C.c_int32), ("numa_node", C.c_int32), ("n_cpus", C.c_int32), ("n_cpus_allowed", C.c_int32),
                ("first_cpu", C.c_int32), ("last_cpu", C.c_int32), ("pci_bus_id", C.c_char * 32)]


# every symbol include/amcx.h declares: (restype, argtypes)
_i64, _i32, _vp, _fp = C.c_int64, C.c_int32, C.c_void_p, C.POINTER(C.c_float)
SIGNATURES = {
    "amcx_abi_version": (C.c_int, []),
    "amcx_strerror": (C.c_char_p, [C.c_int]),
    "amcx_last_hip_error": (C.c_char_p, []),
    "amcx_device_count": (C.c_int, []),
    "amcx_features18_c64": (C.c_int, [_vp, _i64, _i32, _i64, _vp, _i64, _vp]),
    "amcx_features18_c64_ex": (C.c_int, [_vp, _i64, _i32, _i64, _vp, _i64, _vp, _i32]),
    "amcx_features18_workspace_bytes": (_i64, [_i32, _i64, _i32]),
    "amcx_features18_c64_ws": (C.c_int, [_vp, _i64, _i32, _i64, _vp, _i64, _vp, _i32, _vp, _i64]),
    "amcx_features18_c64_host": (C.c_int, [_vp, _i64, _i32, _i64, _vp, _i64, _i32, _i32]),
    "amcx_features18_c128_host": (C.c_int, [_vp, _i64, _i32, _i64, _vp, _i64, _i32, _i32]),
    "amcx_ctx_create": (C.c_int, [_i32, C.POINTER(_vp)]),
    "amcx_ctx_destroy": (C.c_int, [_vp]),
    "amcx_ctx_features18_c64_host": (C.c_int, [_vp, _vp, _i64, _i32, _i64, _vp, _i64, _i32]),
    "amcx_ctx_features18_c128_host": (C.c_int, [_vp, _vp, _i64, _i32, _i64, _vp, _i64, _i32]),
    "amcx_ctx_features18_strided_host": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _i64, _i32, _i64, _i64, _i64,
                                                   _vp, _i64, _i32]),
    "amcx_stage_host": (C.c_int, [_vp, _vp, _i32, _i64, _i64, _i32, _i64, _i64, _i64, _i64, _i64, _vp, _i64, _i32,
                                  C.POINTER(_i32), C.POINTER(_i32)]),
    "amcx_ctx_features18_strided_file": (C.c_int, [_vp, _i32, _i64, _i64, _i32, _i64, _i64, _i32, _i64, _i64, _i64,
                                                   _vp, _i64, _i32]),
    "amcx_stage_file": (C.c_int, [_i32, _i64, _i64, _i32, _i64, _i64, _i32, _i64, _i64, _i64, _i64, _i64, _vp, _i64, _i32,
                                  C.POINTER(_i32), C.POINTER(_i32)]),
    "amcx_ctx_configure": (C.c_int, [_vp, _i32, _i64, _i32]),
    "amcx_ctx_upload_stats": (C.c_int, [_vp, C.POINTER(UploadStats)]),
    "amcx_pack_planes_c64": (C.c_int, [_vp, _i32, _i32, _i64, _i64, _i64, _i32, _vp, _i64, _i32, _vp]),
    "amcx_kernel_name": (C.c_int, [_i32, _i32, C.c_char_p, _i32]),
    "amcx_probe_read_bw": (C.c_int, [_vp, _i64, _vp, _vp]),
    "amcx_probe_fma_rate": (C.c_int, [C.c_double, _vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "amcx_device_pci_bus_id": (C.c_int, [_i32, C.c_char_p, _i32]),
    "amcx_numa_place": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(_i32), C.POINTER(_i32), _i32, C.POINTER(_i32)]),
    "amcx_ctx_bind_cpus": (C.c_int, [_vp, C.POINTER(_i32), _i32]),
    "amcx_ctx_placement": (C.c_int, [_vp, C.POINTER(Placement)]),
    "amcx_group_stats_f32": (C.c_int, [_vp, _i64, _i64, _i64, _i32, _vp, _vp, _vp]),
    "amcx_select_scale_f32": (C.c_int, [_vp, _i64, _i64, _vp, _i32, _vp, _vp, _vp, _i64, _vp]),
    "amcx_group_stats_workspace_bytes": (_i64, [_i64, _i64, _i32]),
    "amcx_group_stats_ws_f32": (C.c_int, [_vp, _i64, _i64, _i64, _i32, _vp, _vp, _vp, _i64, _vp]),
    "amcx_standardize_workspace_bytes": (_i64, [_i64, _i32]),
    "amcx_standardize_fit_transform_f32": (C.c_int, [_vp, _i64, _i64, _i32, C.POINTER(_i32), _i32, _vp, _i64,
                                                     _vp, _vp, _vp, _i64, _vp]),
}

_lib = None
_loaded_without_torch = False


class AmcxError(RuntimeError):
    def __init__(self, code: int, detail: str = ""):
        self.code = code
        super().__init__(f"amcx error {code}: {detail}")


def torch_wanted() -> bool:
    """False when the library was loaded on the system HIP runtime (``load(skip_torch=True)``, or AMCX_SKIP_TORCH=1 in
    the environment) and torch has not been imported: the host-buffer path (containers, files, calculate_features)
    needs neither torch nor its HIP runtime, and a process that never touches torch tensors -- the `extract`
    command line -- starts a second faster without the import."""
    import sys
    if "torch" in sys.modules:
        return True
    return not _loaded_without_torch and os.environ.get("AMCX_SKIP_TORCH", "0") != "1"


def load(skip_torch: bool = False) -> C.CDLL:
    """Load libamcx.so once.  torch (if installed) is imported first so that the
    library binds to the HIP runtime torch ships and device pointers are shared
    (unless the caller opts out -- ``skip_torch``, the command line's choice, or AMCX_SKIP_TORCH=1 -- see
    :func:`torch_wanted`: importing torch AFTER the library has loaded the system runtime would put two HIP
    runtimes into one process).  The opt-out is a property of this first call, not of the environment."""
    global _lib, _loaded_without_torch
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python amcpy_amd/csrc/build.py` "
            "(hipcc, gfx950). amcpy_amd has no CPU fallback.")
    import sys
    if "torch" in sys.modules or (not skip_torch and os.environ.get("AMCX_SKIP_TORCH", "0") != "1"):
        try:
            import torch  # noqa: F401  (side effect: loads torch's libamdhip64.so.7)
        except Exception:
            pass
    else:
        _loaded_without_torch = True
    lib = C.CDLL(str(LIB_PATH), mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype, fn.argtypes = res, args
    got = lib.amcx_abi_version()
    if got < ABI_VERSION:
        raise ImportError(f"libamcx ABI {got} is older than the {ABI_VERSION} this binding needs")
    _lib = lib
    return lib


def require_torch_runtime() -> None:
    """The tensor entry points hand torch's device pointers to the library: both must sit on ONE HIP runtime."""
    if _loaded_without_torch:
        raise RuntimeError("libamcx.so was loaded without torch (load(skip_torch=True) / AMCX_SKIP_TORCH=1: the system HIP "
                           "runtime); torch tensors belong to the runtime torch ships. Import torch before amcpy_amd loads "
                           "the library, or do not opt out.")


def check(code: int) -> None:
    if code == OK:
        return
    lib = load()
    msg = lib.amcx_strerror(code).decode()
    if code in (EHIP, EIO):
        msg += ": " + lib.amcx_last_hip_error().decode()
    if code == EIO:
        raise OSError(f"amcx: {msg}")
    if code == EINVAL:
        raise ValueError(f"amcx: {msg}")
    raise AmcxError(code, msg)


def kernel_name(frame_size: int, variant: int = VARIANT_AUTO) -> str:
    buf = C.create_string_buffer(128)
    check(load().amcx_kernel_name(frame_size, variant, buf, len(buf)))
    return buf.value.decode()


def numa_place(pci_bus_id: str, sysfs_root: str = "") -> tuple:
    """(node, [cpus]) local to the PCI device ``dddd:bb:dd.f`` according to ``<sysfs_root>/bus/pci/devices`` (default
    /sys): amcx_numa_place.  (-1, []) when the platform does not say.  Host-only: needs no GPU."""
    lib = load()
    node, n = C.c_int32(-1), C.c_int32(0)
    check(lib.amcx_numa_place(sysfs_root.encode(), pci_bus_id.encode(), C.byref(node), None, 0, C.byref(n)))
    cpus = (C.c_int32 * max(1, n.value))()
    check(lib.amcx_numa_place(sysfs_root.encode(), pci_bus_id.encode(), C.byref(node), cpus, n.value, C.byref(n)))
    return node.value, [int(c) for c in cpus[:n.value]]


def device_pci_bus_id(device: int) -> str:
    buf = C.create_string_buffer(32)
    check(load().amcx_device_pci_bus_id(int(device), buf, len(buf)))
    return buf.value.decode()


def probe_fma_rate(seconds: float = 1.0, stream: int = 0) -> dict:
    """amcx_probe_fma_rate on the current device: {"wave_instr_per_s", "clock_GHz"} (synchronises the stream)."""
    rate, clock = C.c_double(0.0), C.c_double(0.0)
    check(load().amcx_probe_fma_rate(float(seconds), stream or None, C.byref(rate), C.byref(clock)))
    return {"wave_instr_per_s": rate.value, "clock_GHz": clock.value}


class HostContext:
    """Reusable host-buffer context (amcx_ctx_*): a stream and growing device scratch kept
    across calls.  One per thread and device; freed with the object."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        self.device = int(device)
        check(load().amcx_ctx_create(self.device, C.byref(self._h)))

    def run(self, x2, frame_size: int, out, variant: int) -> None:
        """x2: C-contiguous (F, L) complex64 / complex128 ndarray; out: (F, >=18) float32."""
        import numpy as np
        lib = load()
        entry = lib.amcx_ctx_features18_c128_host if x2.dtype == np.complex128 else lib.amcx_ctx_features18_c64_host
        check(entry(self._h, x2.ctypes.data, x2.shape[0], int(frame_size), x2.shape[1],
                    out.ctypes.data, out.shape[1], int(variant)))

    def configure(self, threads: int = 0, slot_bytes: int = 0, round_on_device: int = -1) -> None:
        check(load().amcx_ctx_configure(self._h, int(threads), int(slot_bytes), int(round_on_device)))

    def run_strided(self, re_ptr: int, im_ptr, kind: int, n_snr: int, n_frames: int, frame_size: int,
                    strides, out, variant: int = VARIANT_AUTO) -> None:
        """amcx_ctx_features18_strided_host: `strides` = (snr, frame, sample) in source elements;
        out: C-contiguous (n_snr * n_frames, >= 18) float32.  The caller keeps the source alive."""
        check(load().amcx_ctx_features18_strided_host(
            self._h, re_ptr, im_ptr, int(kind), int(n_snr), int(n_frames), int(frame_size),
            int(strides[0]), int(strides[1]), int(strides[2]), out.ctypes.data, out.shape[1], int(variant)))

    def run_file(self, fd: int, re_offset: int, im_offset, kind: int, n_snr: int, n_frames: int, frame_size: int,
                 strides, out, variant: int = VARIANT_AUTO) -> None:
        """amcx_ctx_features18_strided_file: the container read from its file by the staging threads (byte
        offsets of the real / imaginary arrays; strides in elements).  OSError if a read fails."""
        check(load().amcx_ctx_features18_strided_file(
            self._h, int(fd), int(re_offset), -1 if im_offset is None else int(im_offset), int(kind), int(n_snr),
            int(n_frames), int(frame_size), int(strides[0]), int(strides[1]), int(strides[2]), out.ctypes.data,
            out.shape[1], int(variant)))

    def bind_cpus(self, cpus) -> None:
        """amcx_ctx_bind_cpus: the staging threads (and the caller during a large upload) on these CPUs; [] unbinds."""
        cpus = [int(c) for c in cpus]
        arr = (C.c_int32 * max(1, len(cpus)))(*cpus)
        check(load().amcx_ctx_bind_cpus(self._h, arr, len(cpus)))

    def placement(self) -> dict:
        pl = Placement()
        check(load().amcx_ctx_placement(self._h, C.byref(pl)))
        return {"device": pl.device, "numa_node": pl.numa_node, "n_cpus": pl.n_cpus, "n_cpus_allowed": pl.n_cpus_allowed,
                "first_cpu": pl.first_cpu, "last_cpu": pl.last_cpu, "pci_bus_id": pl.pci_bus_id.decode()}

    def upload_stats(self) -> dict:
        st = UploadStats()
        check(load().amcx_ctx_upload_stats(self._h, C.byref(st)))
        return {name: getattr(st, name) for name, _ in UploadStats._fields_}

    def close(self) -> None:
        if self._h:
            load().amcx_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
