#!/usr/bin/env python3
"""Feature-kernel step time through the library (features18 on a resident arena) on two kinds of data: the bench's
synthetic modulations and plain Gaussian noise (what tools/wave_clock feeds the kernel).  For same-box A/B of library
builds: run once per build (tools/ab_lib.sh; AMCX_LIB selects the library file)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from amcpy_amd import synth
from amcpy_amd.features import features18

dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048          # frame size: python tools/ab_lib_timing.py [N]
S, K, M = 26, (4096 if N <= 8192 else 512 if N == 16384 else 256), 6       # (the two arenas stay below ~85 GB)
arena = torch.empty((M, S, K, N), dtype=torch.complex64, device=dev)
for mi in range(M):
    synth.device_frames(synth.MODS6[mi], S, K, N, device=dev, rank=0, mod_idx=mi, out=arena[mi])
noise = torch.view_as_complex(torch.randn((M, S, K, N, 2), device=dev) * 0.7071)
out = torch.empty((M, S, K, 18), dtype=torch.float32, device=dev)
for name, data in (("bench modulations", arena), ("gaussian noise", noise), ("bench modulations", arena), ("gaussian noise", noise)):
    for _ in range(400):
        features18(data, out=out)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
    for a, b in ev:
        a.record(); features18(data, out=out); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    print(f"  {name:18s} step min/median/max {ms[0]:.4f}/{ms[len(ms)//2]:.4f}/{ms[-1]:.4f} ms  -> {M*S*K/ms[len(ms)//2]/1e3:.1f} M frames/s")
