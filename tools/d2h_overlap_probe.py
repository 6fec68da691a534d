#!/usr/bin/env python3
"""Per-step durations of the feature launch while the previous result is copied to the host on a second
stream (bench.py's `wall_incl_d2h_ms` leg): where does the overlap fail?  python tools/d2h_overlap_probe.py [steps]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from amcpy_amd import synth
from amcpy_amd.features import features18

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda", 0)
arena = torch.empty((6, 26, 4096, 2048), dtype=torch.complex64, device=dev)
for mi in range(6):
    synth.device_frames(synth.MODS6[mi], 26, 4096, 2048, device=dev, rank=0, mod_idx=mi, out=arena[mi])
RING = 3
outs = [torch.empty((6, 26, 4096, 18), dtype=torch.float32, device=dev) for _ in range(RING)]
hosts = [torch.empty(outs[0].shape, dtype=torch.float32, pin_memory=True) for _ in range(RING)]
main, cp = torch.cuda.current_stream(), torch.cuda.Stream(device=dev)
for _ in range(300):
    features18(arena, out=outs[0])
torch.cuda.synchronize()
k0 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
k1 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
c0 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
c1 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
t0 = time.perf_counter()
for k in range(steps):
    b = k % RING
    if k >= RING:
        main.wait_event(c1[k - RING])
    k0[k].record(main)
    features18(arena, out=outs[b])
    k1[k].record(main)
    with torch.cuda.stream(cp):
        cp.wait_event(k1[k])
        c0[k].record(cp)
        hosts[b].copy_(outs[b], non_blocking=True)
        c1[k].record(cp)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / steps * 1e3
kd = [a.elapsed_time(b) for a, b in zip(k0, k1)]
cd = [a.elapsed_time(b) for a, b in zip(c0, c1)]
gap = [k1[i].elapsed_time(k0[i + 1]) for i in range(steps - 1)]
print(f"wall per step {wall:.3f} ms")
print("kernel ms  :", " ".join(f"{v:.2f}" for v in kd[:40]))
print("copy ms    :", " ".join(f"{v:.2f}" for v in cd[:40]))
print("gap ms     :", " ".join(f"{v:.2f}" for v in gap[:40]))
print(f"kernel mean {sum(kd)/len(kd):.3f} max {max(kd):.3f}; copy mean {sum(cd)/len(cd):.3f} max {max(cd):.3f}; gap mean {sum(gap)/len(gap):.3f} max {max(gap):.3f}")
print("kernel ms, means of 10:", " ".join(f"{sum(kd[i:i + 10]) / len(kd[i:i + 10]):.2f}" for i in range(0, steps, 10)))
print("copy   ms, means of 10:", " ".join(f"{sum(cd[i:i + 10]) / len(cd[i:i + 10]):.2f}" for i in range(0, steps, 10)))
# when does each copy start relative to the end of the launch it copies, and how long after it does the next launch end?
lag = [k1[i].elapsed_time(c0[i]) for i in range(steps)]
print("copy start lag behind its launch, means of 10:", " ".join(f"{sum(lag[i:i + 10]) / len(lag[i:i + 10]):.2f}" for i in range(0, steps, 10)))
