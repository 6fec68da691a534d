#!/usr/bin/env python3
"""Wide parity sweep of the HIP path against the oracle (GPU box): every modulation, 26 SNRs,
every wave-kernel size plus two Bluestein sizes of the block kernel.  Prints the worst scaled and
plain relative error per feature and writes a JSON summary (bench.py replays it as `parity`):

    python tests/manual/parity_sweep.py [frames_per_cell=12] [out.json]

Criterion (tests/test_gpu_parity.py): features 1-9, 11 plain relative; cumulants 10, 12-18 relative
to max(|value|, S), S = sum |terms| of the cumulant's formula.  `beyond_unfloored` counts frames over
1e-5 on that scale; `worst_floored` uses max(S, 2e-3 * S with every moment replaced by the mean of
its summands' magnitudes) -- chance cancellation of a whole moment (|mean x^6| 1000x below mean |x|^6)
otherwise collapses S below what any fp32 accumulation can resolve.  Not part of the test-suite."""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch  # noqa: E402
from amcpy_amd import synth  # noqa: E402
from amcpy_amd.features import features18  # noqa: E402
from oracle import iq_features_oracle as orc  # noqa: E402

frames_per = int(sys.argv[1]) if len(sys.argv) > 1 else 12
out_json = sys.argv[2] if len(sys.argv) > 2 else None
SUM_FLOOR = 2e-3
summary = {"criterion": "ids 1-9, 11: |got-ref|/|ref|; ids 10, 12-18: |got-ref|/max(|ref|, S), S = sum|terms| "
                        "(SURVEY.md 8c); ref = oracle on the complex128 cast, stored float32",
           "floor": f"worst_floored: S floored at {SUM_FLOOR} x S(|summand| means)", "sizes": {}}
for N in (128, 256, 512, 1000, 1024, 1536, 2048, 4096, 8192):
    per = frames_per if N in (1024, 2048, 4096) else max(4, frames_per // 4)
    worst_s = np.zeros(18)
    worst_p = np.zeros(18)
    worst_f = np.zeros(18)
    n = beyond = 0
    t0 = time.time()
    for mi, mod in enumerate(synth.MODS6):
        for si, snr in enumerate(synth.snr_grid(26)):
            x = synth.host_block(mod, float(snr), per, N, seed=90000 + 100 * mi + si)
            gold = orc.features18_batch(x).astype(np.float32)
            S = orc.conditioning_scales(x)
            Sf = np.maximum(S, SUM_FLOOR * orc.conditioning_scales(x, absolute=True))
            got = features18(torch.from_numpy(x).cuda(), variant="auto").cpu().numpy()
            p, s = orc.parity_errors(got, gold, S)
            _, f = orc.parity_errors(got, gold, Sf)
            worst_s = np.maximum(worst_s, s.max(axis=0))
            worst_p = np.maximum(worst_p, p.max(axis=0))
            worst_f = np.maximum(worst_f, f.max(axis=0))
            beyond += int((s > 1e-5).any(axis=1).sum())
            n += per
    print(f"N={N} frames={n} ({time.time()-t0:.0f}s)  beyond unfloored 1e-5: {beyond}")
    print("  worst scaled :", " ".join(f"{v:.1e}" for v in worst_s), " max", f"{worst_s.max():.2e}")
    print("  worst floored:", " ".join(f"{v:.1e}" for v in worst_f), " max", f"{worst_f.max():.2e}")
    print("  worst plain  :", " ".join(f"{v:.1e}" for v in worst_p))
    summary["sizes"][str(N)] = {"frames": n, "worst_scaled": float(worst_s.max()),
                                "worst_scaled_feature": int(worst_s.argmax()) + 1,
                                "beyond_unfloored": beyond, "worst_floored": float(worst_f.max()),
                                "worst_plain_ids_1_9_11": float(worst_p[[0, 1, 2, 3, 4, 5, 6, 7, 8, 10]].max()),
                                "worst_plain_per_feature": [float(v) for v in worst_p],
                                "worst_scaled_per_feature": [float(v) for v in worst_s]}
if out_json:
    Path(out_json).write_text(json.dumps(summary, indent=1))
