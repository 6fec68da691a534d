"""The 18 per-frame IQ features on MI355X.

Mirror of the reference's per-frame seam (src/amcpy/features.py:214-232):

    calculate_features(feature_ids, signal) -> list[float]

with the same argument meaning, result order and ``KeyError`` for an unknown
id -- computed by the HIP kernels behind the C ABI (include/amcx.h), never on
the CPU.  The batch entry points the extraction driver uses are
:func:`features18` (torch tensor in HBM -> torch tensor in HBM, asynchronous on
the current stream) and :func:`features18_host` (numpy in, numpy out).

Feature ids (reference config.py:118-137): 1 gamma_max, 2 sigma_ap,
3 sigma_dp, 4 sigma_aa, 5 sigma_af, 6 X, 7 X2, 8 mu42^a, 9 mu42^f,
10..18 |C20| |C21| |C40| |C41| |C42| |C60| |C61| |C62| |C63|.
"""
from __future__ import annotations

import threading
from typing import Iterable, List

import numpy as np

from . import _lib

FEATURE_IDS = tuple(range(1, _lib.NUM_FEATURES + 1))


def _variant(v) -> int:
    return _lib.VARIANTS[v] if isinstance(v, str) else int(v)


def features18(iq, out=None, *, frame_size: int | None = None, variant="auto"):
    """All 18 features of every frame of a complex64 CUDA(HIP) tensor.

    iq   : torch.complex64 tensor on a GPU, shape (..., L) with unit stride in the
           last dimension and uniformly strided frames (any leading shape that
           flattens to [n_frames][row_stride], e.g. the reference's
           (n_snr, n_frames, L) container).  Only the first ``frame_size``
           samples of each row are used (feature_extraction.py:68).
    out  : optional float32 tensor (..., >=18) on the same device.
    Returns the float32 tensor (..., 18); the launch is asynchronous on the
    current torch stream.
    """
    import torch

    _lib.require_torch_runtime()
    if not isinstance(iq, torch.Tensor) or iq.dtype != torch.complex64:
        raise TypeError("iq must be a torch.complex64 tensor")
    if not iq.is_cuda:
        raise ValueError("iq must live in GPU memory (use features18_host for numpy input)")
    if iq.dim() < 1:
        raise ValueError("iq needs at least one dimension")
    L = iq.shape[-1]
    N = L if frame_size is None else int(frame_size)
    if N > L:
        raise ValueError(f"frame_size {N} exceeds row length {L}")
    lead = iq.shape[:-1]
    n_frames = int(np.prod(lead)) if lead else 1
    if iq.stride(-1) != 1 and L > 1:
        raise ValueError("last dimension must have unit stride")
    flat = iq.reshape(n_frames, L) if iq.dim() != 2 else iq
    if flat.data_ptr() != iq.data_ptr() or (n_frames > 1 and flat.stride(1) != 1):
        raise ValueError("frames must be uniformly strided (no copy is made)")
    row_stride = flat.stride(0) if n_frames > 1 else max(L, N)
    if out is None:
        out = torch.empty(lead + (_lib.NUM_FEATURES,), dtype=torch.float32, device=iq.device)
    else:
        if out.dtype != torch.float32 or out.device != iq.device:
            raise TypeError("out must be float32 on the same device")
        if tuple(out.shape[:-1]) != tuple(lead) or out.shape[-1] < _lib.NUM_FEATURES:
            raise ValueError("out must have shape (..., >=18) matching iq")
    oflat = out.reshape(n_frames, out.shape[-1]) if out.dim() != 2 else out
    if oflat.data_ptr() != out.data_ptr() or oflat.stride(-1) != 1:
        raise ValueError("out must be uniformly strided with unit stride in the last dimension")
    out_stride = oflat.stride(0) if n_frames > 1 else out.shape[-1]
    lib = _lib.load()
    v = _variant(variant)
    with torch.cuda.device(iq.device):
        stream = torch.cuda.current_stream(iq.device).cuda_stream
        # The any-size path above 8192 samples wants a workspace for its FFT form (amcx_features18_workspace_bytes: 0 for
        # every other size).  It comes from TORCH's allocator here -- the allocator that owns this process's device
        # memory -- through amcx_features18_c64_ws: left to amcx_features18_c64_ex it would come from HIP's own
        # stream-ordered pool, which knows nothing of what torch has cached, and a failed allocation there silently
        # selects the O(N^2) form (~100x slower at N = 32767).  torch raises if it cannot provide the bytes.
        need = int(lib.amcx_features18_workspace_bytes(N, n_frames, v)) if n_frames > 0 else 0
        if need > 0:
            ws = torch.empty(need, dtype=torch.uint8, device=iq.device)
            _lib.check(lib.amcx_features18_c64_ws(iq.data_ptr(), n_frames, N, row_stride, oflat.data_ptr(), out_stride,
                                                  stream, v, ws.data_ptr(), need))
            ws.record_stream(torch.cuda.current_stream(iq.device))       # freed for reuse behind the launch, not before
        else:
            _lib.check(lib.amcx_features18_c64_ex(
                iq.data_ptr(), n_frames, N, row_stride, oflat.data_ptr(), out_stride, stream, v))
    return out[..., :_lib.NUM_FEATURES]


def features18_iq_pairs(iq_pairs, **kw):
    """Zero-copy entry for data stored as float32 (I, Q) pairs: RadioML-style
    ``(..., N, 2)`` float32 arrays and raw GNU-Radio complex64 streams reshaped to
    ``(F, N, 2)`` (reference old/dataset.py:50-56, old/read_binary_stream.py:28,48) are
    bit-identical to the kernel's complex64 layout, so the tensor is only re-viewed."""
    import torch
    if not isinstance(iq_pairs, torch.Tensor) or iq_pairs.dtype != torch.float32 or iq_pairs.shape[-1] != 2:
        raise TypeError("expected a float32 tensor whose last dimension is (I, Q)")
    if iq_pairs.stride(-1) != 1 or (iq_pairs.shape[-2] > 1 and iq_pairs.stride(-2) != 2):
        raise ValueError("(I, Q) pairs must be interleaved in memory")
    return features18(torch.view_as_complex(iq_pairs), **kw)


_tls = threading.local()


def _host_context(device: int) -> "_lib.HostContext":
    """This thread's reusable context for `device` (amcx_ctx_*: stream + device scratch kept
    across calls, so calculate_features in a loop is not allocation-bound)."""
    cache = getattr(_tls, "ctx", None)
    if cache is None:
        cache = _tls.ctx = {}
    ctx = cache.get(device)
    if ctx is None:
        ctx = cache[device] = _lib.HostContext(device)
    return ctx


def features18_host(frames: np.ndarray, *, frame_size: int | None = None, device: int = 0,
                    variant="auto") -> np.ndarray:
    """numpy (..., L) complex -> numpy (..., 18) float32 via the GPU.

    complex128 input (MATLAB doubles) is uploaded as it is and rounded to complex64,
    the engine's input type, on the GPU.  Raises if no MI355X is present (AMCX_ENODEV)."""
    x = np.asarray(frames)
    if not np.iscomplexobj(x):
        x = x.astype(np.complex64)
    L = x.shape[-1]
    N = L if frame_size is None else int(frame_size)
    if N > L:
        raise ValueError(f"frame_size {N} exceeds row length {L}")
    lead = x.shape[:-1]
    if x.dtype == np.complex128:
        # MATLAB doubles: uploaded as they are, rounded to complex64 on the GPU
        x2 = np.ascontiguousarray(x.reshape(-1, L))
    else:
        x2 = np.ascontiguousarray(x.reshape(-1, L), dtype=np.complex64)
    out = np.empty((x2.shape[0], _lib.NUM_FEATURES), dtype=np.float32)
    _host_context(int(device)).run(x2, N, out, _variant(variant))
    return out.reshape(lead + (_lib.NUM_FEATURES,))


def calculate_features(feature_ids: Iterable[int], signal, *, device: int = 0,
                       variant="auto") -> List[float]:
    """Drop-in for the reference's ``calculate_features`` (features.py:214-232):
    values in the order of ``feature_ids`` (repeats and subsets allowed); an id
    outside 1..18 raises ``KeyError`` before anything is launched."""
    ids = list(feature_ids)
    for fid in ids:
        if fid not in FEATURE_IDS:
            raise KeyError(fid)
    sig = np.asarray(signal)
    if sig.ndim != 1:
        raise ValueError("signal must be one frame (1-D complex array)")
    row = features18_host(sig[None, :], device=device, variant=variant)[0]
    return [float(row[fid - 1]) for fid in ids]
