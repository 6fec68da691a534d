#!/usr/bin/env python3
"""Wide parity sweep of the HIP path against the oracle (GPU box): every modulation, 26 SNRs,
every wave-kernel size plus two Bluestein sizes of the block kernel.  Prints the worst scaled and
plain relative error per feature and writes a JSON summary (bench.py replays it as `parity`):

    python tests/manual/parity_sweep.py [frames_per_cell=12] [out.json]

Criterion (tests/test_gpu_parity.py): features 1-9, 11 plain relative; cumulants 10, 12-18 relative
to max(|value|, S), S = sum |terms| of the cumulant's formula.  `beyond_unfloored` counts frames over
1e-5 on that scale; `worst_floored` uses max(S, 2e-3 * S with every moment replaced by the mean of
its summands' magnitudes) -- chance cancellation of a whole moment (|mean x^6| 1000x below mean |x|^6)
otherwise collapses S below what any fp32 accumulation can resolve.  At the BASELINE frame sizes every
fourth frame is also run through the oracle IN COMPLEX64 (features18_frame(dtype=complex64): the
reference's own arithmetic on the same samples): `ref_c64_plain_per_feature` is that path's worst plain
relative distance from the complex128 result, `kernel_plain_same_frames` the kernel's on the same frames,
`kernel_beyond_twice_ref_gap` the frames where the kernel is outside both the scaled 1e-5 and twice the
reference's gap.  Not part of the test-suite.  (Round 6: the test-suite holds EVERY frame to the unfloored criterion --
`beyond_unfloored` must be 0 at every size; `worst_floored` is the rule rounds 2-5 judged large samples by, kept here for
comparison with their records.)"""
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
    ref32_gap = np.zeros(18)
    kern_same = np.zeros(18)
    n = beyond = n32 = beyond_gap = 0
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
            if N in (1024, 2048, 4096):
                pick = np.arange(0, per, 4)
                with np.errstate(all="ignore"):
                    g32 = np.stack([orc.features18_frame(x[i], dtype=np.complex64) for i in pick]).astype(np.float64)
                    g64 = gold[pick].astype(np.float64)
                    gap = np.abs(g32 - g64)
                    err = np.abs(got[pick].astype(np.float64) - g64)
                    ref32_gap = np.maximum(ref32_gap, np.nan_to_num(gap / np.abs(g64)).max(axis=0))
                    kern_same = np.maximum(kern_same, np.nan_to_num(err / np.abs(g64)).max(axis=0))
                    allowed = np.maximum(1e-5 * np.maximum(np.abs(g64), S[pick]), 2.0 * gap)
                beyond_gap += int((err > allowed).any(axis=1).sum())
                n32 += len(pick)
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
    if n32:
        print("  ref complex64 path, plain :", " ".join(f"{v:.1e}" for v in ref32_gap), f"({n32} frames)")
        print("  kernel, same frames, plain:", " ".join(f"{v:.1e}" for v in kern_same),
              f" beyond scaled 1e-5 AND twice the reference's gap: {beyond_gap}")
        summary["sizes"][str(N)].update({"frames_also_run_in_complex64": n32,
                                         "ref_c64_plain_per_feature": [float(v) for v in ref32_gap],
                                         "kernel_plain_same_frames": [float(v) for v in kern_same],
                                         "kernel_beyond_twice_ref_gap": beyond_gap})
if out_json:
    Path(out_json).write_text(json.dumps(summary, indent=1))
