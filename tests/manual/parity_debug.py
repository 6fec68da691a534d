import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from amcpy_amd import synth
from amcpy_amd.features import features18
from oracle import iq_features_oracle as orc
N = 4096
worst = []
for mi, mod in enumerate(synth.MODS6):
    for si, snr in enumerate(synth.snr_grid(26)):
        x = synth.host_block(mod, float(snr), 12, N, seed=90000 + 100 * mi + si)
        gold = orc.features18_batch(x)
        S = orc.conditioning_scales(x)
        gw = features18(torch.from_numpy(x).cuda(), variant="wave").cpu().numpy()
        gb = features18(torch.from_numpy(x).cuda(), variant="block").cpu().numpy()
        pw, sw = orc.parity_errors(gw, gold.astype(np.float32), S)
        pb, sb = orc.parity_errors(gb, gold.astype(np.float32), S)
        for f in range(12):
            for j in (4, 8):
                worst.append((sw[f, j], mod, snr, f, j + 1, gw[f, j], gb[f, j], gold[f, j], sb[f, j]))
worst.sort(reverse=True)
for w in worst[:8]:
    print("err %.2e %s snr %g frame %d feat %d wave %.9g block %.9g gold %.9g (block err %.1e)" % w)
# detail on the worst frame
_, mod, snr, f, j, *_ = worst[0]
mi = synth.MODS6.index(mod); si = list(synth.snr_grid(26)).index(snr)
x = synth.host_block(mod, float(snr), 12, N, seed=90000 + 100 * mi + si)[f]
th = np.angle(x.astype(np.complex128)); w = orc.wrapped_first_difference(th)
d = np.diff(th)
print("steps near +-pi:", np.sort(np.abs(np.abs(d) - np.pi))[:5], " std(w)", w.std(), "kurt", orc.pearson_kurtosis(w))
