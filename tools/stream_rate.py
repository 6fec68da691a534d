#!/usr/bin/env python3
"""Rate of the any-size fallback above 8192 samples (amcx_features18_stream_kernel: AMCX_VARIANT_BLOCK, and AUTO at the
sizes that are not powers of two): frames/s at a few sizes, next to the group kernels where they exist.
    python tools/stream_rate.py [frames]          (GPU box)"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch                                                   # noqa: E402
from amcpy_amd import _lib                                     # noqa: E402
from amcpy_amd.features import features18                      # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for N in (8193, 10000, 12289, 16384, 20000, 24576, 32767, 32768):
    g = torch.Generator(device="cuda").manual_seed(N)
    x = torch.view_as_complex(torch.randn((F, N, 2), device="cuda", generator=g))
    for variant in (["block", "wave"] if N in (16384, 32768) else ["auto"]):
        out = features18(x, variant=variant)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            features18(x, out=out, variant=variant)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"N = {N:6d}  {variant:5s} {_lib.kernel_name(N, _lib.VARIANTS[variant]):40s} {F / dt:12.0f} frames/s  "
              f"({dt * 1e3:8.2f} ms for {F} frames)", flush=True)
