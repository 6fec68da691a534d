#!/usr/bin/env python3
"""Wide parity sweep of the HIP path against the oracle (GPU box): every modulation,
26 SNRs, several seeds, every wave-kernel size plus two Bluestein sizes of the block kernel.  Prints the worst scaled and
plain relative error per feature; not part of the test-suite (takes ~1 min)."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from amcpy_amd import synth
from amcpy_amd.features import features18
from oracle import iq_features_oracle as orc

frames_per = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for N in (128, 256, 512, 1000, 1024, 1536, 2048, 4096, 8192):
    worst_s = np.zeros(18); worst_p = np.zeros(18); n = 0
    t0 = time.time()
    for mi, mod in enumerate(synth.MODS6):
        for si, snr in enumerate(synth.snr_grid(26)):
            x = synth.host_block(mod, float(snr), frames_per, N, seed=90000 + 100 * mi + si)
            gold = orc.features18_batch(x).astype(np.float32)
            S = orc.conditioning_scales(x)
            for variant in ("auto",):
                got = features18(torch.from_numpy(x).cuda(), variant=variant).cpu().numpy()
                p, s = orc.parity_errors(got, gold, S)
                worst_s = np.maximum(worst_s, s.max(axis=0)); worst_p = np.maximum(worst_p, p.max(axis=0))
            n += frames_per
    print(f"N={N} frames={n} ({time.time()-t0:.0f}s)")
    print("  worst scaled:", " ".join(f"{v:.1e}" for v in worst_s), " max", f"{worst_s.max():.2e}")
    print("  worst plain :", " ".join(f"{v:.1e}" for v in worst_p))
