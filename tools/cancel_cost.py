#!/usr/bin/env python3
"""What a frame flagged by the cancellation predicate costs (round 6).  Builds, from the bench's own synthetic arena, three
arenas of the same size -- as generated, flagged frames replaced by unflagged ones, every frame a flagged one -- and times the
feature kernel on each (same box, same library).  Which frames are flagged is read off the results: the library given as
REFERENCE (a build without the exact path, e.g. round 5's) differs from the tree's in columns 9-17 of exactly those rows.

    AMCX_LIB=amcpy_amd/lib/libamcx.so python tools/cancel_cost.py [frame_size=2048] [reference_out.pt]
    AMCX_LIB=amcpy_amd/lib/libamcx_r5.so python tools/cancel_cost.py 2048 --write-reference gpurun_out/ref.pt
    ... --save-mask m.pt / --load-mask m.pt: the same three arenas under another library (one without the exact path)
"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from amcpy_amd import synth
from amcpy_amd.features import features18

dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
S, K, M = 26, 4096 * 2048 // N if N >= 2048 else 4096, 6
arena = torch.empty((M, S, K, N), dtype=torch.complex64, device=dev)
for mi in range(M):
    synth.device_frames(synth.MODS6[mi], S, K, N, device=dev, rank=0, mod_idx=mi, out=arena[mi])
flat = arena.view(-1, N)
F = flat.shape[0]
out = features18(flat)
torch.cuda.synchronize()
if "--write-reference" in sys.argv:
    torch.save(out.cpu(), sys.argv[sys.argv.index("--write-reference") + 1])
    print("reference written", F, "frames")
    sys.exit(0)
if "--load-mask" in sys.argv:                       # a library without the exact path on the SAME arenas: the mask another run saved
    flagged = torch.load(sys.argv[sys.argv.index("--load-mask") + 1], weights_only=True).to(dev)
else:
    ref = torch.load(sys.argv[2], weights_only=True).to(dev)
    assert torch.equal(out[:, :9].view(torch.int32), ref[:, :9].view(torch.int32)), "features 1-9 differ between the libraries"
    flagged = (out[:, 9:].view(torch.int32) != ref[:, 9:].view(torch.int32)).any(dim=1)
if "--save-mask" in sys.argv:
    torch.save(flagged.cpu(), sys.argv[sys.argv.index("--save-mask") + 1])
nf = int(flagged.sum())
per_mod = flagged.view(M, -1).float().mean(dim=1).tolist()
print(f"N={N}: {nf} of {F} frames differ in ids 10-18 ({nf / F:.3%}); per modulation " + " ".join(f"{m}:{p:.3%}" for m, p in zip(synth.MODS6, per_mod)))


def timed(x, label):
    o = torch.empty((x.shape[0], 18), dtype=torch.float32, device=dev)
    for _ in range(300):
        features18(x, out=o)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(100)]
    for a, b in ev:
        a.record(); features18(x, out=o); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    print(f"  {label:42s} median {ms[len(ms) // 2]:.4f} ms -> {x.shape[0] / ms[len(ms) // 2] / 1e3:.1f} M frames/s")
    return ms[len(ms) // 2]


t_as_is = timed(flat, "as generated")
idx_f = flagged.nonzero().flatten()
idx_u = (~flagged).nonzero().flatten()
torch.manual_seed(1)
clean = flat.clone()
clean[idx_f] = flat[idx_u[torch.randint(0, idx_u.numel(), (idx_f.numel(),), device=dev)]]
t_clean = timed(clean, "flagged frames replaced by unflagged ones")
del clean
for share in (0.05, 0.25, 1.0):
    mixed = flat.clone()
    n_put = int(F * share)
    where = torch.randperm(F, device=dev)[:n_put]
    mixed[where] = flat[idx_f[torch.randint(0, idx_f.numel(), (n_put,), device=dev)]]
    t = timed(mixed, f"{share:.0%} of the frames flagged ones")
    del mixed
    print(f"    -> a flagged frame costs {(t - t_clean) / (share - 0) / t_clean:.2f} frame times more than an ordinary one")
print(f"  as generated vs clean: {100 * (t_as_is / t_clean - 1):+.2f} %  at a flag rate of {nf / F:.3%}")
