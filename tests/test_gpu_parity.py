"""Parity of the HIP path (through the C ABI) with the oracle and the golden
vectors captured from the reference.  Needs an MI355X: -m gpu.

Tolerance (BASELINE.json north_star "within 1e-5
relative fp32"): features 1-9 and 11 plain relative <= 1e-5 against golden64
(the reference evaluated on the complex128 cast of the same complex64 frame,
stored float32); the cancellation-dominated cumulants 10, 12-18 within
1e-5 * max(|golden64|, S), S = sum of |terms| of the cumulant's formula
(S = m21 for id 10).  The reference's own complex64 path misses plain 1e-5 on
those ids by up to 8.5e-3, so plain relative error is
reported, not asserted, for them.
"""
import json
import os
import textwrap
from pathlib import Path
import numpy as np
import pytest

from oracle import iq_features_oracle as orc
from tests.conftest import load_npz

pytestmark = pytest.mark.gpu

REPO = Path(__file__).resolve().parents[1]

TOL = 1e-5
VARIANTS_POW2 = ["block", "wave"]


def _torch():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    return torch


def _run(frames, variant, frame_size=None):
    """numpy (F, L) complex64 -> numpy (F, 18) float32 via device tensors."""
    torch = _torch()
    from amcpy_amd.features import features18
    x = torch.from_numpy(np.ascontiguousarray(frames)).cuda()
    y = features18(x, frame_size=frame_size, variant=variant)
    torch.cuda.synchronize()
    return y.cpu().numpy()


def _variants_for(N):
    from amcpy_amd import _lib
    out = ["block"]                                  # every size up to AMCX_MAX_BLOCK_FRAME_SIZE = 32768 (above 8192: amcx_stream_kernel.h)
    try:
        _lib.kernel_name(N, _lib.VARIANT_WAVE)
        out.append("wave")
    except Exception:
        pass
    return out


EDGE_RTOL = 1e-5      # test_golden_edges


def _assert_parity(got, golden_f64, frames, what):
    """THE tolerance of the module docstring, at every call site and for every frame -- a dozen golden frames or the
    6 000 rows of configs[0]: ids 1-9, 11 within 1e-5 plain relative; ids 10, 12-18 within 1e-5 of max(|golden64|, S).
    (Until round 6 samples of thousands of frames were judged by a looser rule -- S floored at 2e-3 of the summands'
    scale, 0.1 % of the frames allowed up to 10x the bound, one frame of configs[0] whitelisted -- because among thousands
    of noise-like frames a few have a cumulant whose terms all cancel to ~1/1000 of the summands they are averaged from,
    below what fp32 sums resolve.  The kernels' finalisers now find those frames themselves and give them fp64 sums in the
    same launch: amcx_math.h cancellation_suspect, amcx_wave_kernel.h wave_exact_cumulants.)"""
    S = orc.conditioning_scales(frames)
    plain, scaled = orc.parity_errors(got, golden_f64.astype(np.float32), S)
    worst = scaled.max(axis=0)
    print(f"\n[{what}] worst scaled rel per feature:", " ".join(f"{v:.1e}" for v in worst))
    print(f"[{what}] worst plain  rel per feature:", " ".join(f"{v:.1e}" for v in plain.max(axis=0)))
    strict = [i for i in range(18) if i < 9 or i == 10]
    assert plain[:, strict].max() <= TOL, f"{what}: feature {strict[int(plain[:, strict].max(axis=0).argmax())] + 1} off by {plain[:, strict].max():.3e} (plain relative)"
    over = np.flatnonzero((scaled > TOL).any(axis=1))
    assert worst.max() <= TOL, (f"{what}: feature {int(worst.argmax()) + 1} off by {worst.max():.3e}; {over.size} of {len(scaled)} "
                                f"frames beyond 1e-5 of max(|value|, S): {over[:20].tolist()}")


def test_oracle_pin_holds_on_this_box(golden_frames):
    """The checker checked where it is used: this box's numpy / scipy are not the build container's, and every test
    below trusts oracle.features18_batch.  The same comparisons tests/test_oracle.py makes on the CPU -- the oracle's
    reference-shaped and fused evaluators against what the imported reference produced (tests/golden, captured by
    oracle/capture_golden.py) -- run here too."""
    from tests import test_oracle as pin
    pin.test_reference_shaped_matches_reference_outputs(golden_frames)
    pin.test_fused_matches_reference_outputs(golden_frames)
    pin.test_moments_match_reference(golden_frames)


def test_oracle_pin_edges_and_known_answers_on_this_box(golden_edges, kat):
    """... and the degenerate frames and the reference's own known-answer table (features.py:286-305)."""
    from tests import test_oracle as pin
    pin.test_kat_reference_shaped(kat)
    pin.test_kat_fused(kat)
    pin.test_edges_fused_vs_reference(golden_edges)


def test_library_loads_and_sees_gpu():
    from amcpy_amd import _lib
    lib = _lib.load()
    assert lib.amcx_device_count() >= 1


def test_kat_through_calculate_features(kat):
    """The reference's own known-answer table (features.py:286-305), N = 10,
    through the drop-in per-frame entry point."""
    from amcpy_amd.features import calculate_features
    x = np.array(kat["signal_re"]) + 1j * np.array(kat["signal_im"])
    got = calculate_features(list(range(1, 19)), x)
    for fid, (g, e) in enumerate(zip(got, kat["expected"]), start=1):
        assert np.isclose(g, e, rtol=kat["rtol"], atol=0), f"feature {fid}: {g} vs {e}"
    sub = calculate_features([14, 2, 2, 7], x)
    assert np.allclose(sub, [got[13], got[1], got[1], got[6]], rtol=0, atol=0)
    with pytest.raises(KeyError):
        calculate_features([0], x)
    with pytest.raises(KeyError):
        calculate_features([1, 19], x)


def test_golden_frames(golden_frames):
    N, g = golden_frames
    for variant in _variants_for(N):
        got = _run(g["iq"], variant)
        _assert_parity(got, g["golden64_f64"], g["iq"], f"golden N={N} {variant}")


def test_golden_edges(golden_edges):
    N, g = golden_edges
    names = [str(s) for s in g["names"]]
    ref = g["golden64_f64"]
    S = orc.conditioning_scales(np.nan_to_num(g["iq"]), absolute=True)
    for variant in _variants_for(N):
        got = _run(g["iq"], variant).astype(np.float64)
        for i, name in enumerate(names):
            r, o = ref[i], got[i]
            if name == "one_nan":
                assert np.isnan(o).all(), (variant, name, o)
                continue
            if name in ("zeros", "const_pos", "impulse"):
                assert (np.isnan(r) == np.isnan(o)).all(), (variant, name, r, o)
            # kurtosis of a numerically constant series is rounding noise in the
            # reference itself: not a parity target
            skip = np.zeros(18, bool)
            if name not in ("real_only", "imag_only"):
                skip[[7, 8]] = True
            if name in ("real_only", "imag_only"):
                skip[8] = True
            if name in ("ramp_phase_pi", "imag_only"):
                # every phase step is pi - O(1e-16): an exact +-pi tie in fp32, just
                # off the tie in fp64, so the sign of each wrapped step (hence f5)
                # is decided by rounding residue in the reference itself; the
                # deterministic tie case is the 'alternating' frame.  ('imag_only':
                # re == 0 exactly, so sign flips of im are +-pi steps as well.)
                skip[4] = True
            sel = ~np.isnan(r) & ~skip
            assert not np.isnan(o[sel]).any(), (variant, name, o, r)
            # the kurtoses that are not held to the reference here (rounding noise on a numerically constant series, in the
            # reference as well) are still held to what a Pearson kurtosis of n values can be at all: NaN (scipy's
            # degenerate rule) or a number in [1, n]
            for j, n_vals in ((7, N), (8, N - 1)):
                if skip[j]:
                    assert np.isnan(o[j]) or (1.0 - 1e-3 <= o[j] <= n_vals * (1.0 + 1e-3)), (variant, name, j + 1, o[j])
            # cumulants (ids 10-18): 1e-5 of max(|ref|, S) like everywhere else (rounds 1-4 allowed 2e-5 here without a
            # reason; tightened in round 5); ids 1-9: the same plus 2e-6 absolute, because series that are exactly
            # constant in fp64 (zero std) carry fp32 rounding dust of ~1e-7 here
            atol = np.where(np.arange(18) < 9, 2e-6, 0.0)
            lim = EDGE_RTOL * np.maximum(np.abs(r), S[i]) + atol
            bad = (np.abs(o - r) > lim) & sel
            assert not bad.any(), (variant, name, np.nonzero(bad)[0] + 1, o, r)


def test_random_frames_against_oracle():
    """Seeded synthetic frames the oracle finishes in seconds, every modulation
    and a wide SNR range, each power-of-two size the fast kernel serves."""
    from amcpy_amd import synth
    for N in (128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768):
        blocks = [synth.host_block(m, snr, 8 if N <= 8192 else 3, N, seed=77 + 13 * i + j)
                  for i, m in enumerate(synth.MODS6) for j, snr in enumerate((-20.0, -6.0, 8.0, 30.0))]
        x = np.concatenate(blocks).astype(np.complex64)
        gold = orc.features18_batch(x)
        for variant in _variants_for(N):
            _assert_parity(_run(x, variant), gold, x, f"synthetic N={N} {variant}")


@pytest.mark.parametrize("N", [128, 256, 512, 1024, 2048, 4096])
def test_full_snr_grid_against_oracle(N):
    """The whole SNR grid of the BASELINE configs -- 6 modulations x 26 SNRs (-20 ... +30 dB, step 2) x 8 frames = 1 248
    frames per frame size, the host generator with SURVEY 8d's seeds (1000 + 10 mod + snr index) -- against the oracle,
    run by the DRIVER's suite (until round 5 this breadth existed only as a builder-run sweep, tests/manual/
    parity_sweep.py, replayed into the bench line).  NOT ONE frame beyond |got - golden64| <= 1e-5 max(|golden64|, S)."""
    from amcpy_amd import synth
    snrs = np.linspace(-20.0, 30.0, 26)
    x = np.concatenate([synth.host_block(m, float(snr), 8, N, seed=1000 + 10 * mi + si)
                        for mi, m in enumerate(synth.MODS6) for si, snr in enumerate(snrs)]).astype(np.complex64)
    assert x.shape == (1248, N)
    gold = orc.features18_batch(x)
    for variant in _variants_for(N):
        _assert_parity(_run(x, variant), gold, x, f"full SNR grid N={N} {variant}")


@pytest.mark.parametrize("N", [128, 512, 1024, 2048, 4096, 8192, 16384])
def test_cancelling_cumulants_take_the_fp64_path(N):
    """Frames built so that whole cumulants cancel: noiseless QPSK at 8 samples per symbol with EXACTLY balanced squares
    (as many symbols with x^2 = +u as with x^2 = -u: mean x^2 = 0, so C41 = m41 - 3 m20 m21 and C60 vanish up to the
    little noise added) -- 5 % of a noiseless QPSK cell's frames are like that by chance.  S = sum |terms| of those
    cumulants is then ~1e-4 of the summands' scale, far below what fp32 sums resolve (~1e-8 of that scale: the bound
    1e-5 S would be missed by 10x); every throughput kernel's finaliser must flag such a frame (amcx_math.h:
    cancellation_suspect) and take its moment sums from the fp64 sweep (wave_exact_cumulants).  Checked: the contract
    1e-5 max(|value|, S) on every frame; that the path was TAKEN (the result is within 2e-7 of S -- ten times tighter
    than fp32 sums could be -- on the cancelling ids); rows longer than the frame, an output wider than 18 columns; and
    the same frames at 2^40 and 2^-40 times the amplitude, where the flagged frame is ALSO outside the fp32 sums'
    range: the re-run on a scaled copy flags it again and the features follow the scaling laws."""
    torch = _torch()
    from amcpy_amd.features import features18
    rng = np.random.default_rng(N)
    F, sps = 24, 8
    n_sym = N // sps
    pts = np.exp(1j * (np.pi / 4 + np.pi / 2 * np.arange(4)))
    x = np.empty((F, N + 24), np.complex128)
    for f in range(F):
        half = n_sym // 2
        # x^2 = +i e^{2 i phi} for points 0, 2 and -i e^{2 i phi} for points 1, 3: half of the symbols from each pair
        sym = np.concatenate([rng.choice([0, 2], half), rng.choice([1, 3], n_sym - half)])
        rng.shuffle(sym)
        base = np.repeat(pts[sym], sps) * np.exp(1j * rng.uniform(0, 2 * np.pi))
        noise = (rng.standard_normal(N) + 1j * rng.standard_normal(N)) * 1e-4
        x[f, :N] = base + noise
        x[f, N:] = 7.0 + 3.0j                                     # beyond the frame: must not be read
    x = x.astype(np.complex64)
    frames = x[:, :N]
    gold = orc.features18_batch(frames)
    S = orc.conditioning_scales(frames)
    assert (S[:, [12, 14]] < 3e-3).all(), S[:, [12, 14]].max(axis=0)      # ids 13 and 15 do cancel: S << the summands' scale (~1 ... 16)
    xd = torch.from_numpy(x).cuda()
    out = torch.full((F, 24), -7.0, dtype=torch.float32, device="cuda")
    features18(xd, out=out, frame_size=N, variant="wave")
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert (got[:, 18:] == -7.0).all()                                    # columns beyond 18 are the caller's
    plain, scaled = orc.parity_errors(got[:, :18], gold.astype(np.float32), S)
    print(f"\n[cancelling QPSK N={N}] worst scaled rel per feature:", " ".join(f"{v:.1e}" for v in scaled.max(axis=0)))
    # ids 4 and 8 are statistics of |x| about its mean, and |x| is constant here up to the 1e-4 of noise: rounding noise in
    # the reference as well (the constant-modulus edge frames); every other id is held to the contract
    held = [j for j in range(18) if j not in (3, 7)]
    assert plain[:, [j for j in held if j < 9 or j == 10]].max() <= TOL and scaled[:, held].max() <= TOL, scaled.max(axis=0)
    # taken, not merely passed: fp32 sums are good to ~1e-8 of the SUMMANDS' scale = 1e-5 ... 1e-4 of such an S
    assert scaled[:, [12, 14]].max() <= 2e-6, scaled[:, [12, 14]].max(axis=0)
    # out of the fp32 sums' range AND cancelling: re-run on a scaled copy, flagged again, un-scaled through the laws
    order = np.array([2, 0, 0, 0, 0, 1, 0.5, 0, 0, 2, 2, 4, 4, 4, 6, 6, 6, 6])
    for p2 in (40, -40):
        sc = np.float32(2.0 ** p2)
        ys = features18(torch.from_numpy(frames * sc).cuda(), variant="wave").cpu().numpy().astype(np.float64)
        with np.errstate(over="ignore", under="ignore", invalid="ignore"):
            want = (got[:, :18].astype(np.float64) * float(sc) ** order[None, :]).astype(np.float32)
        fin = np.isfinite(want) & (np.abs(want) > 1e-30)
        rel = np.abs(ys.astype(np.float32)[fin] / want[fin] - 1.0)
        assert rel.max() <= 3e-6, (N, p2, rel.max())
        assert (np.isinf(want) == np.isinf(ys)).all()


def test_variants_agree():
    from amcpy_amd import synth
    x = np.concatenate([synth.host_block(m, 6.0, 5, 2048, seed=5 + i)
                        for i, m in enumerate(synth.MODS6)]).astype(np.complex64)
    vs = _variants_for(2048)
    if len(vs) < 2:
        pytest.skip("only one kernel variant built")
    a, b = (_run(x, v) for v in vs)
    S = orc.conditioning_scales(x)
    _, scaled = orc.parity_errors(a, b, S)
    assert scaled.max() <= TOL


def test_wave_kernel_short_power_of_two_frames():
    """N = 128, 256, 512: four frames per wave, sixteen lanes per frame (amcx_short_kernel.h: row reductions, a 4 x 8 x 4 /
    8 x 8 x 4 / 16 x 8 x 4 register FFT with two LDS transposes -- at 512 in two batches of eight 32-point transforms --, every
    frame scaled by a power of two).  Frame counts chosen to leave ragged passes, batches and
    tail chunks; pure tones on bins of every residue of the short kernel's index split X[kj + R (kc + 8 ka)] (each residue takes
    another lane and register through the transposes).  Checked against the oracle and the block kernel."""
    from amcpy_amd import synth, _lib
    for N, F in ((128, 1531), (256, 777), (512, 403)):
        assert "short_kernel" in _lib.kernel_name(N, _lib.VARIANT_AUTO)
        x = np.concatenate([synth.host_block(m, snr, F // 3 + 1, N, seed=N + i)
                            for i, (m, snr) in enumerate((("BPSK", 0.0), ("16QAM", 12.0), ("WGN", -10.0)))])[:F]
        n = np.arange(N)
        for i, k in enumerate((1, 2, 3, 5, 6, 7, 37, 41, N // 2, N // 2 + 3, N - 10, N - 1)):
            x[5 + 7 * i] = (0.7 * np.exp(2j * np.pi * (k * n / N + 0.1 * i))).astype(np.complex64) + x[5 + 7 * i] * np.float32(0.1)
        gold = orc.features18_batch(x)
        got = _run(x, "wave")
        _assert_parity(got, gold, x, f"wave N={N}")
        blk = _run(x, "block")
        S = orc.conditioning_scales(x)
        _, scaled = orc.parity_errors(got, blk, S)
        assert scaled.max() <= 2 * TOL, (N, scaled.max(axis=0))       # two results, each within TOL of the oracle
        assert np.array_equal(_run(x, "auto"), got)
        # a frame's 18 floats depend on its samples only, not on which copy of the frame body
        # (ping-pong register set, exchange slot) or which batch it lands in
        perm = np.random.default_rng(N).permutation(F)
        assert np.array_equal(_run(x[perm], "wave"), got[perm]), N


def test_wave_kernel_8192():
    """N = 8192: four waves per frame (amcx_quad_kernel.h) -- quarters in registers, radix-4 exchange through
    LDS, four 2048-point register FFTs; against the oracle and the block kernel (LDS radix-2 FFT, fp64 sums).
    Pure tones on bins of every residue mod 4 (each residue is one wave's FFT), frame counts that leave ragged
    batches and idle workgroups, and every frame bit-identical whatever its position in the batch."""
    from amcpy_amd import synth, _lib
    N = 8192
    assert _lib.kernel_name(N, _lib.VARIANT_AUTO) == "amcx_features18_quad_kernel"
    x = np.concatenate([synth.host_block(m, snr, 23, N, seed=N + i)
                        for i, (m, snr) in enumerate((("BPSK", -5.0), ("QPSK", 3.0), ("64QAM", 20.0), ("WGN", 0.0)))])
    # pure tones: residues 3, 0, 1, 2 of the bin index mod 4 -- the peak comes from a different wave each time
    n = np.arange(N)
    x[0] = np.exp(2j * np.pi * 1235 * n / N).astype(np.complex64)
    x[1] = np.exp(2j * np.pi * 2468 * n / N).astype(np.complex64)
    x[2] = np.exp(2j * np.pi * 4097 * n / N).astype(np.complex64)
    x[3] = np.exp(2j * np.pi * 8190 * n / N).astype(np.complex64)
    gold = orc.features18_batch(x)
    got = _run(x, "wave")
    assert np.all(np.abs(got[:4, 0] / N - 1.0) < 1e-5)                               # |X|^2 / N = N
    _assert_parity(got[4:], gold[4:], x[4:], "quad N=8192")
    blk = _run(x, "block")
    _, scaled = orc.parity_errors(got, blk, orc.conditioning_scales(x, absolute=True))
    assert scaled[:, 0].max() <= 2 * TOL and scaled[4:].max() <= 2 * TOL, scaled.max(axis=0)   # two results, each within TOL of the oracle
    # ragged batches (4 frames each), fewer batches than workgroups, one frame alone: bit-identical rows
    for count in (1, 2, 3, 5, 7, 41, 92):
        sub = _run(x[:count], "wave")
        assert np.array_equal(sub, got[:count], equal_nan=True), count
    perm = np.random.default_rng(8).permutation(x.shape[0])
    assert np.array_equal(_run(x[perm], "wave"), got[perm], equal_nan=True)


@pytest.mark.parametrize("N,W", [(16384, 8), (32768, 16)])
def test_group_kernels_16384_and_32768(N, W):
    """N = 16384 / 32768: eight / sixteen waves per frame (amcx_group_kernel.h) -- 2048-sample blocks in registers, two
    radix stages across the group through LDS (2 then 4; 4 then 4), W register FFTs; frame_size is a free integer in the
    reference (config.py:96, np.fft.fft: features.py:68).  Against the oracle; pure tones on bins of every residue mod W
    (each residue is one wave's FFT, and the stage twiddles decide whether its energy arrives); ragged batches, fewer
    batches than workgroups, one frame alone, a permutation: rows bit-identical whatever the position; a NaN, an
    infinite and an all-zero frame in the middle; frames outside the fp32 sums' range re-run in the same launch."""
    from amcpy_amd import synth, _lib
    assert _lib.kernel_name(N, _lib.VARIANT_AUTO) == f"amcx_features18_group_kernel<{W}>"
    assert _lib.kernel_name(N, _lib.VARIANT_WAVE) == _lib.kernel_name(N)
    x = np.concatenate([synth.host_block(m, snr, 9, N, seed=N + i)
                        for i, (m, snr) in enumerate((("BPSK", -5.0), ("QPSK", 3.0), ("64QAM", 20.0), ("WGN", 0.0)))])
    n = np.arange(N)
    bins = [(W * 37 + r) % N for r in range(W)] + [N - 1, N // 2, N // 2 + 1, 1]
    for k, b in enumerate(bins):
        x[k] = np.exp(2j * np.pi * b * n / N).astype(np.complex64)
    T = len(bins)
    gold = orc.features18_batch(x)
    got = _run(x, "wave")
    assert np.all(np.abs(got[:T, 0] / N - 1.0) < 1e-5), got[:T, 0] / N                  # |X|^2 / N = N on every residue
    _assert_parity(got[T:], gold[T:], x[T:], f"group N={N}")
    for count in (1, 2, 3, 5, 7, 17, 33):
        sub = _run(x[:count], "wave")
        assert np.array_equal(sub, got[:count], equal_nan=True), count
    perm = np.random.default_rng(8).permutation(x.shape[0])
    assert np.array_equal(_run(x[perm], "wave"), got[perm], equal_nan=True)
    # the reference's per-frame seam (features.py:214-232) at this frame size: one frame through the host context
    from amcpy_amd.features import calculate_features
    row = calculate_features(list(range(1, 19)), x[T + 2])
    assert np.array_equal(np.asarray(row, dtype=np.float32), got[T + 2])
    assert calculate_features([18, 1], x[T + 2].astype(np.complex128)) == [row[17], row[0]]
    # bad frames stay to themselves
    y = x.copy()
    y[T + 1, N // 2] = np.nan
    y[T + 4, 5] = complex(np.inf, 0.0)
    y[T + 6] = 0
    bad = _run(y, "wave")
    assert np.isnan(bad[T + 1]).all() and np.isnan(bad[T + 4]).all()
    keep = np.ones(len(x), bool)
    keep[[T + 1, T + 4, T + 6]] = False
    assert np.array_equal(bad[keep], got[keep], equal_nan=True)                       # (the tones' f8 / f9 are NaN)
    zero = orc.features18_batch(y[T + 6:T + 7])[0]
    assert (np.isnan(bad[T + 6]) == np.isnan(zero)).all() and np.allclose(bad[T + 6][~np.isnan(zero)], zero[~np.isnan(zero)])
    # out of the fp32 sums' range, scattered through batches and across an epoch's mask words: the scaling laws hold exactly
    z = np.tile(x[T:T + 8], (9, 1))                                                    # 72 frames
    sc = np.ones(len(z), np.float32)
    sc[[0, 5, 31, 32, 33, 64, 71]] = [2.0 ** 30, 2.0 ** -40, 2.0 ** 24, 2.0 ** 26, 2.0 ** -34, 2.0 ** 28, 2.0 ** -30]
    zs = (z * sc[:, None]).astype(np.complex64)
    a, b = _run(z, "wave").astype(np.float64), _run(zs, "wave").astype(np.float64)
    order = np.array([2, 0, 0, 0, 0, 1, 0.5, 0, 0, 2, 2, 4, 4, 4, 6, 6, 6, 6])
    with np.errstate(over="ignore", under="ignore", invalid="ignore"):
        want = (a * sc[:, None].astype(np.float64) ** order[None, :]).astype(np.float32)
    fin = np.isfinite(want) & (want != 0)
    assert np.allclose(b.astype(np.float32)[fin], want[fin], rtol=3e-6, atol=0), np.abs(b.astype(np.float32)[fin] / want[fin] - 1).max()
    assert (np.isinf(want) == np.isinf(b)).all()


@pytest.mark.parametrize("N,W,extra", [(16384, 8, 1024), (32768, 16, 512)])
def test_group_kernels_across_epoch_boundaries(N, W, extra):
    """The group kernels keep their re-run list as a bit mask in LDS that covers one EPOCH of 2048 frames of a workgroup;
    a launch in which every workgroup owns more than 2048 frames (256 workgroups x 2048 frames and some: 69 GB at N = 16384,
    137 GB at N = 32768 -- a fraction of the card's 288 GB) crosses epoch boundaries everywhere.  A block of ordinary frames is
    tiled over the arena; frames out of the fp32 sums' range are planted in the last frames of a first epoch, the first of a
    second one, mid-epoch, and at the very end.  Every row must equal the row of the same frame computed in a small launch
    (bit for bit: results do not depend on the position), so a mark set in one epoch and re-run in another, a mark lost at a
    boundary, or a re-run of the wrong frame shows."""
    torch = _torch()
    from amcpy_amd import synth
    from amcpy_amd.features import features18
    torch.cuda.empty_cache()                                          # (what earlier tests left in torch's allocator counts as free)
    free, _ = torch.cuda.mem_get_info()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    F = cus * 2048 + extra * (cus // 256 or 1)
    need = F * N * 8 + (4 << 30)
    if free < need:
        pytest.skip(f"needs {need >> 30} GiB of free device memory")
    T = 512                                                              # the tile: 512 ordinary frames
    tile = torch.from_numpy(np.concatenate([synth.host_block(m, 8.0, T // 4, N, seed=N + i)
                                            for i, m in enumerate(("BPSK", "QPSK", "64QAM", "WGN"))])).cuda()
    arena = torch.empty((F, N), dtype=torch.complex64, device="cuda")
    for a in range(0, F, T):
        b = min(F, a + T)
        arena[a:b] = tile[: b - a]
    batch = 8 if W == 8 else 2
    per_wg = -(-(-(-F // batch)) // cus) * batch                         # frames a workgroup owns (ceil over batches)
    assert per_wg > 2048                                                 # every workgroup crosses an epoch boundary
    planted = sorted({2046, 2047, 2048, 2049, 3, per_wg + 2047, per_wg + 2048, 5 * per_wg + 1000, F - 1, F - 2,
                      (cus - 1) * per_wg + 2048})
    planted = [f for f in planted if f < F]
    scales = [2.0 ** (26 + 2 * (k % 4)) if k % 2 == 0 else 2.0 ** (-(30 + 2 * (k % 3))) for k in range(len(planted))]
    for f, sc in zip(planted, scales):
        arena[f] *= sc
    out = features18(arena)
    torch.cuda.synchronize()
    base = features18(tile)                                              # the tile alone: one small launch
    small = features18(arena[torch.tensor(planted, device="cuda")].contiguous())
    torch.cuda.synchronize()
    got = out.view(torch.int32)
    idx = torch.arange(F, device="cuda") % T
    want = base.view(torch.int32)[idx]
    pl = torch.tensor(planted, device="cuda")
    want[pl] = small.view(torch.int32)
    bad = (got != want).any(dim=1).nonzero().flatten()
    assert bad.numel() == 0, (N, bad[:10].tolist(), per_wg)
    # and the planted frames really took the re-run path: their scale-free features equal the ordinary frame's closely,
    # their scaled ones follow the scaling laws (exactly, as powers of two)
    order = torch.tensor([2, 0, 0, 0, 0, 1, 0.5, 0, 0, 2, 2, 4, 4, 4, 6, 6, 6, 6], dtype=torch.float64, device="cuda")
    ref = base.double()[pl % T] * torch.tensor(scales, dtype=torch.float64, device="cuda")[:, None] ** order[None, :]
    fin = torch.isfinite(ref.float()) & (ref.float() != 0)
    rel = ((small.double() - ref).abs() / ref.abs())
    rel[~fin] = 0.0
    strict = [0, 1, 2, 3, 4, 5, 6, 7, 8, 10]                          # ids 1-9 and 11: not cancellation-dominated
    assert rel[:, strict].max() < 1e-5 and rel.max() < 1e-3, (rel[:, strict].max(), rel.max())
    del arena, out, want, got
    torch.cuda.empty_cache()


def test_run_extraction_at_frame_size_16384(tmp_path):
    """The whole drop-in at a frame size the reference accepts (config.py:96 is a free integer) and rounds 1-4 refused: a
    column-major container of doubles with rows longer than the frame, `run_extraction(cfg)` with frame_size = 16384 -- the
    staging threads, the device transposition and the eight-waves-per-frame kernel -- six files, against the oracle."""
    import scipy.io
    from amcpy_amd import synth
    from amcpy_amd.config import Config, Paths, SignalConfig
    from amcpy_amd.feature_extraction import release_engines, run_extraction
    N, n_frames = 16384, 3
    cfg = Config(paths=Paths(root=tmp_path), signals=SignalConfig(snr_values={0: "0", 1: "10"}, num_frames=n_frames, frame_size=N))
    cfg.paths.ensure_dirs()
    blocks = {}
    for mi, m in enumerate(cfg.signals.modulations_with_noise):
        x = np.stack([synth.host_block(m, snr, n_frames, N + 100, seed=160 + 10 * mi + si) for si, snr in enumerate((0.0, 10.0))])
        blocks[m] = x.astype(np.complex128) * (1.0 + 1e-9)
    scipy.io.savemat(str(cfg.paths.mat_data / cfg.paths.mat_filename), {cfg.signals.mat_info[m]: v for m, v in blocks.items()})
    run_extraction(cfg, verbose=False)
    release_engines()
    for m, v in blocks.items():
        got = scipy.io.loadmat(str(cfg.paths.calculated_features / f"{m}_features.mat"))[cfg.signals.mat_info[m]]
        assert got.shape == (2, n_frames, 18) and got.dtype == np.float32
        x = v[:, :, :N].reshape(-1, N)
        _assert_parity(got.reshape(-1, 18), orc.features18_batch(x), x, f"run_extraction N=16384 {m}")


def test_bad_frames_do_not_leak_into_neighbours():
    """Grouped short frames share FFT passes 2-3 and a finaliser batch; the ping-pong variants
    share registers across frames: a NaN / Inf / all-zero frame in the middle of a batch must
    give NaNs (resp. the all-zero result) for itself only, and leave its neighbours' floats
    exactly as they are without it."""
    from amcpy_amd import synth
    for N in (128, 256, 512, 1024, 2048, 4096, 8192, 1000):
        F = 40
        x = synth.host_block("16QAM", 10.0, F, N, seed=7 * N)
        clean = _run(x, "auto")
        y = x.copy()
        y[3, N // 2] = np.nan
        y[12, 5] = complex(np.inf, 0.0)
        y[21] = 0
        y[22] = 1.0 + 0.0j
        got = _run(y, "auto")
        assert np.isnan(got[3]).all() and np.isnan(got[12]).all(), N
        keep = np.ones(F, bool)
        keep[[3, 12, 21, 22]] = False
        assert np.array_equal(got[keep], clean[keep]), N
        zero = orc.features18_batch(y[21:23])
        for row, ref in zip(got[21:23], zero):
            assert (np.isnan(row) == np.isnan(ref)).all(), (N, row, ref)
            ok = ~np.isnan(ref)
            ok[[7, 8]] = False                     # kurtosis of a constant series: rounding noise in the reference too
            assert np.allclose(row[ok], ref[ok], rtol=EDGE_RTOL, atol=2e-6), (N, row, ref)    # test_golden_edges' bound


def test_results_do_not_depend_on_batch_position():
    """Same property for the long-frame variants and the block kernel."""
    from amcpy_amd import synth
    for N, F in ((128, 517), (256, 401), (512, 333), (1024, 301), (2048, 203), (4096, 101), (8192, 67), (16384, 37), (32768, 21), (100, 57)):
        x = synth.host_block("64QAM", 8.0, F, N, seed=N)
        perm = np.random.default_rng(N).permutation(F)
        for variant in _variants_for(N):
            a = _run(x, variant)
            assert np.array_equal(_run(x[perm], variant), a[perm]), (N, variant)
            assert np.array_equal(_run(x[: F // 3], variant), a[: F // 3]), (N, variant)
            for tiny in (1, 2, 3, 9):              # fewer frames than waves in a workgroup
                assert np.array_equal(_run(x[5:5 + tiny], variant), a[5:5 + tiny]), (N, variant, tiny)


def test_generic_sizes_block_kernel():
    """Frame sizes outside the fast set -- the reference accepts any N (np.fft.fft): tiny and
    odd lengths (direct DFT, N <= 64), non powers of two through Bluestein's chirp-z FFT
    (65 <= N <= 4096 with both spectra in LDS; 4097 <= N <= 8191 with a 16384-point convolution whose chirp
    spectrum lives in registers; primes and both ends of the ranges included), and the radix-2 LDS FFT for
    the powers of two the wave kernel leaves."""
    from amcpy_amd import _lib
    assert _lib.kernel_name(64) == "amcx_features18_block_kernel<1>"
    assert _lib.kernel_name(65) == _lib.kernel_name(4095) == "amcx_features18_block_kernel<2>"
    assert _lib.kernel_name(4097) == _lib.kernel_name(8191) == "amcx_features18_block_kernel<3>"
    assert _lib.kernel_name(10) == _lib.kernel_name(63) == "amcx_features18_block_kernel<0>"
    rng = np.random.default_rng(3)
    for N in (3, 7, 10, 64, 65, 100, 127, 1000, 1536, 2047, 3000, 4093, 4095, 4097, 5000, 6007, 8191, 256, 8192):
        F = 3
        x = (rng.standard_normal((F, N)) + 1j * rng.standard_normal((F, N))).astype(np.complex64)
        x += np.exp(2j * np.pi * 0.05 * np.arange(N))[None, :].astype(np.complex64)
        gold = orc.features18_batch(x)
        got = _run(x, "auto")
        S = orc.conditioning_scales(x)
        plain, scaled = orc.parity_errors(got, gold.astype(np.float32), S)
        ok = np.isfinite(gold)
        # a handful of samples: every per-sample rounding shows in the statistic
        assert scaled[ok].max() <= TOL, (N, scaled.max(axis=0))


def test_row_stride_and_slicing():
    """Rows longer than frame_size: only the first frame_size samples count
    (feature_extraction.py:68); works on a strided view without a copy."""
    torch = _torch()
    from amcpy_amd.features import features18
    from amcpy_amd import synth
    L, N = 2304, 2048
    x = synth.host_block("QPSK", 10.0, 6, L, seed=9)
    xd = torch.from_numpy(x).cuda()
    y = features18(xd, frame_size=N).cpu().numpy()
    y2 = features18(xd[:, :N].contiguous()).cpu().numpy()
    assert np.array_equal(y, y2)
    gold = orc.features18_batch(x[:, :N])
    _assert_parity(y, gold, x[:, :N], "strided rows")
    # 3-D container layout (n_snr, n_frames, L) and a padded output
    x3 = xd.reshape(2, 3, L)
    out = torch.full((2, 3, 24), -1.0, device="cuda")
    r = features18(x3, out=out, frame_size=N)
    torch.cuda.synchronize()
    assert r.shape == (2, 3, 18)
    assert np.array_equal(r.cpu().numpy().reshape(6, 18), y)
    assert (out[..., 18:] == -1).all()


def test_host_buffer_entry_matches_device_entry():
    from amcpy_amd.features import features18_host
    from amcpy_amd import synth
    x = synth.host_block("16QAM", 12.0, 4, 1024, seed=21)
    a = features18_host(x)
    b = _run(x, "auto")
    assert np.array_equal(a, b)
    # complex128 input is rounded to complex64 first
    c = features18_host(x.astype(np.complex128))
    assert np.array_equal(a, c)


def test_argument_errors():
    torch = _torch()
    from amcpy_amd import _lib
    from amcpy_amd.features import features18
    lib = _lib.load()
    x = torch.zeros((2, 64), dtype=torch.complex64, device="cuda")
    o = torch.zeros((2, 18), dtype=torch.float32, device="cuda")
    f = lib.amcx_features18_c64
    assert f(x.data_ptr(), 2, 64, 32, o.data_ptr(), 18, None) == _lib.EINVAL      # stride < N
    assert f(x.data_ptr(), 2, 64, 64, o.data_ptr(), 17, None) == _lib.EINVAL      # out stride < 18
    assert f(x.data_ptr(), -1, 64, 64, o.data_ptr(), 18, None) == _lib.EINVAL
    assert f(None, 2, 64, 64, o.data_ptr(), 18, None) == _lib.EINVAL
    assert f(x.data_ptr(), 2, 1, 64, o.data_ptr(), 18, None) == _lib.EINVAL       # N < 2
    assert f(x.data_ptr(), 2, 1 << 20, 1 << 20, o.data_ptr(), 18, None) == _lib.EINVAL
    assert f(x.data_ptr(), 2, 65536, 65536, o.data_ptr(), 18, None) == _lib.EINVAL          # AMCX_MAX_FRAME_SIZE is 32768
    assert f(x.data_ptr(), 2, 32769, 32769, o.data_ptr(), 18, None) == _lib.EINVAL
    # 8193 ... 32767: every size has a kernel since ABI 6 (amcx_stream_kernel.h); the one-wave / multi-wave kernels only the powers of two
    big = torch.zeros((2, 16384), dtype=torch.complex64, device="cuda")
    assert f(big.data_ptr(), 2, 10000, 16384, o.data_ptr(), 18, None) == _lib.OK
    assert f(big.data_ptr(), 2, 8193, 8193, o.data_ptr(), 18, None) == _lib.OK
    assert lib.amcx_features18_c64_ex(big.data_ptr(), 2, 16384, 16384, o.data_ptr(), 18, None, _lib.VARIANT_BLOCK) == _lib.OK
    assert lib.amcx_features18_c64_ex(big.data_ptr(), 2, 10000, 10000, o.data_ptr(), 18, None, _lib.VARIANT_WAVE) == _lib.ENOTSUP
    assert _lib.kernel_name(10000, _lib.VARIANT_AUTO) == "amcx_features18_stream_kernel"
    assert _lib.kernel_name(16384, _lib.VARIANT_BLOCK) == "amcx_features18_stream_kernel"
    torch.cuda.synchronize()
    assert f(None, 0, 64, 64, None, 18, None) == _lib.OK                          # empty batch
    assert lib.amcx_features18_c64_ex(x.data_ptr(), 2, 100, 100, o.data_ptr(), 18, None,
                                      _lib.VARIANT_WAVE) == _lib.ENOTSUP
    with pytest.raises(ValueError):
        features18(x, frame_size=128)
    with pytest.raises(TypeError):
        features18(x.real)
    empty = features18(torch.zeros((0, 64), dtype=torch.complex64, device="cuda"))
    assert empty.shape == (0, 18)


def test_linearity_properties_full_size():
    """Size-independent properties at a benchmark-shaped batch (too large for
    the oracle): scaling x by c scales feature j by c**p_j exactly in the
    homogeneous features; a global phase rotation leaves |x|-based features
    and cumulant magnitudes unchanged; frame order does not matter."""
    torch = _torch()
    from amcpy_amd.features import features18
    from amcpy_amd import synth
    x = synth.device_frames("64QAM", 4, 4096, 2048, device="cuda", rank=0, mod_idx=4)
    y = features18(x).double()
    c = 2.0                                             # exact in fp32: bitwise-scaled sums
    yc = features18(x * c).double()
    power = torch.tensor([2, 0, 0, 0, 0, 1, 0.5, 0, 0, 2, 2, 4, 4, 4, 6, 6, 6, 6],
                         dtype=torch.float64, device="cuda")
    ratio = yc / (y * c ** power)
    assert torch.allclose(ratio, torch.ones_like(ratio), rtol=2e-6, atol=0)
    # rotation by exactly -1 (phase pi): x -> -x is exact in fp32
    yr = features18(-x).double()
    inv = [0, 3, 4, 5, 6, 7, 8] + list(range(9, 18))    # all but the two raw-phase stds
    assert torch.allclose(yr[..., inv], y[..., inv], rtol=1e-6, atol=1e-7)
    # permutation of frames permutes rows
    perm = torch.randperm(4096, device="cuda")
    yp = features18(x[:, perm].contiguous()).double()
    assert torch.equal(yp, y[:, perm])


def test_run_extraction_roundtrip_on_gpu(tmp_path):
    """The drop-in batch driver against what the reference's own
    run_extraction wrote for the same container (fixture captured from the
    imported reference): same files, keys, dtype, shape; values within
    tolerance.  Rows are longer than frame_size and complex128, as MATLAB gives."""
    import scipy.io
    from amcpy_amd.config import Config, Paths, SignalConfig
    from amcpy_amd.feature_extraction import run_extraction
    g = load_npz("extract_roundtrip.npz")
    fs, n_frames = int(g["frame_size"]), int(g["n_frames"])
    mods = [str(m) for m in g["mods"]]
    cfg = Config(paths=Paths(root=tmp_path),
                 signals=SignalConfig(snr_values={0: "0", 1: "10"}, num_frames=n_frames, frame_size=fs))
    cfg.paths.ensure_dirs()
    scipy.io.savemat(str(cfg.paths.mat_data / cfg.paths.mat_filename),
                     {cfg.signals.mat_info[m]: g[f"in_{m}"].astype(np.complex128) for m in mods})
    run_extraction(cfg, verbose=False)
    for m in mods:
        d = scipy.io.loadmat(str(cfg.paths.calculated_features / f"{m}_features.mat"))
        assert sorted(k for k in d if not k.startswith("__")) == sorted(["Modulation", cfg.signals.mat_info[m]])
        arr = d[cfg.signals.mat_info[m]]
        assert arr.dtype == np.float32 and arr.shape == (2, n_frames, 18)
        assert str(np.ravel(d["Modulation"])[0]) == m
        x = g[f"in_{m}"][:, :, :fs].reshape(-1, fs)
        _assert_parity(arr.reshape(-1, 18), g[f"out_{m}"].reshape(-1, 18).astype(np.float64), x,
                       f"run_extraction {m}")


_RANK_WORKER = """
import os, sys
import numpy as np
sys.path.insert(0, os.environ["AMCX_REPO"])
import torch, torch.distributed as dist
from pathlib import Path
from amcpy_amd.config import Config, Paths, SignalConfig
from amcpy_amd.feature_extraction import run_extraction
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)                      # one GPU on the box: both ranks compute on it
dist.init_process_group("gloo", rank=rank, world_size=world)
cfg = Config(paths=Paths(root=Path(os.environ["AMCX_ROOT"])),
             signals=SignalConfig(snr_values={0: "0", 1: "10"}, num_frames=int(os.environ["AMCX_NFRAMES"]),
                                  frame_size=int(os.environ["AMCX_FS"])))
run_extraction(cfg, verbose=False)
dist.barrier()
# extract_modulation: every rank holds the array; the ranks cut it along the frame axis (or the flattening when
# there are fewer frames per row than ranks) and rank 0 gets what one process computes
from amcpy_amd.feature_extraction import FrameRows, HipEngine, extract_modulation
rng = np.random.default_rng(5)
for shape, nf in (((2, 9, cfg.signals.frame_size + 3), 9), ((3, 1, cfg.signals.frame_size), 1)):
    arr = np.asfortranarray(rng.standard_normal(shape) + 1j * rng.standard_normal(shape))
    c2 = Config(paths=cfg.paths, signals=SignalConfig(snr_values={i: str(i) for i in range(shape[0])}, num_frames=nf,
                                                      frame_size=cfg.signals.frame_size))
    got = extract_modulation(arr, c2)
    if rank == 0:
        want = HipEngine(cfg.signals.frame_size)(FrameRows(arr, shape[0], nf)).reshape(shape[0], nf, 18)
        assert got.shape == want.shape and np.array_equal(got, want, equal_nan=True), shape
    else:
        assert got is None
dist.barrier()
dist.destroy_process_group()
print("RANK_DONE", rank)
"""


def test_two_ranks_run_extraction_sharded(tmp_path):
    """The N>1 path end to end with the real engine: two processes under torch.distributed
    (gloo rendezvous -- the box has one GPU, which both ranks compute on), each taking its
    contiguous half of every modulation's frames; rank 0 gathers and writes the .mat files,
    which must equal the single-process result bit for bit."""
    import socket
    import subprocess
    import sys
    import scipy.io
    from amcpy_amd.config import Config, Paths, SignalConfig
    from amcpy_amd.feature_extraction import run_extraction
    g = load_npz("extract_roundtrip.npz")
    fs, n_frames = int(g["frame_size"]), int(g["n_frames"])
    mods = [str(m) for m in g["mods"]]
    roots = {k: tmp_path / k for k in ("single", "sharded")}
    for root in roots.values():
        cfg = Config(paths=Paths(root=root),
                     signals=SignalConfig(snr_values={0: "0", 1: "10"}, num_frames=n_frames, frame_size=fs))
        cfg.paths.ensure_dirs()
        scipy.io.savemat(str(cfg.paths.mat_data / cfg.paths.mat_filename),
                         {cfg.signals.mat_info[m]: g[f"in_{m}"].astype(np.complex128) for m in mods})
    run_extraction(Config(paths=Paths(root=roots["single"]),
                          signals=SignalConfig(snr_values={0: "0", 1: "10"}, num_frames=n_frames, frame_size=fs)),
                   verbose=False)
    script = tmp_path / "rank_worker.py"
    script.write_text(_RANK_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    repo = str(Path(__file__).resolve().parents[1])
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), AMCX_REPO=repo, AMCX_ROOT=str(roots["sharded"]),
                   AMCX_NFRAMES=str(n_frames), AMCX_FS=str(fs), PYTHONDONTWRITEBYTECODE="1",
                   HSA_ENABLE_IPC_MODE_LEGACY="0", AMCX_SHARE_GPU="1")      # both ranks on the box's one GPU, on purpose
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    cfg = Config(paths=Paths(root=roots["single"]))
    for m in mods:
        a = scipy.io.loadmat(str(roots["single"] / "calculated-features" / f"{m}_features.mat"))
        b = scipy.io.loadmat(str(roots["sharded"] / "calculated-features" / f"{m}_features.mat"))
        key = cfg.signals.mat_info[m]
        assert a[key].shape == b[key].shape == (2, n_frames, 18)
        diff = np.argwhere(a[key] != b[key])
        assert diff.size == 0, (m, diff[:5], a[key][tuple(diff[0])], b[key][tuple(diff[0])])


_NCCL_ONE_RANK_WORKER = """
import os, sys
import numpy as np
sys.path.insert(0, os.environ["AMCX_REPO"])
import torch, torch.distributed as dist
from pathlib import Path
from amcpy_amd.config import Config, Paths, SignalConfig
from amcpy_amd.feature_extraction import run_extraction
from amcpy_amd import sharding
assert sharding.collectives_forced()
rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
assert (rank, world) == (0, 1)
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
assert "nccl" in str(dist.get_backend()).lower()
one = torch.ones(1, dtype=torch.int32, device=dev)
dist.all_reduce(one)                                                  # the first RCCL collective of the process
assert int(one.item()) == 1
# 1. gather_blocks down its DEVICE-TENSOR branch (padded cuda tensors, dist.gather over RCCL) against the host shortcut
rng = np.random.default_rng(3)
block = rng.standard_normal((1237, 18)).astype(np.float32)
block[5, 3] = np.nan
got = sharding.gather_blocks(block, [1237], 0, 1)
assert isinstance(got, list) and len(got) == 1 and np.array_equal(got[0].view(np.int32), block.view(np.int32))
rows = sharding.gather_rows(block, 1237, 0, 1)
assert rows is not block and np.array_equal(rows.view(np.int32), block.view(np.int32))
cols = sharding.gather_frame_columns(block[:1236], 2, 618, 0, 1)
assert np.array_equal(cols.view(np.int32), block[:1236].reshape(2, 618, 18).view(np.int32))
t = torch.from_numpy(block).to(dev)
every = sharding.all_gather_rows(t, 1237, 0, 1)                      # all_gather_into_tensor over RCCL
assert every.is_cuda and every.data_ptr() != t.data_ptr() and torch.equal(every.view(torch.int32), t.view(torch.int32))
# 2. run_extraction through the multi-rank branch: status words, broadcast of the container's place, the gather
cfg = Config(paths=Paths(root=Path(os.environ["AMCX_ROOT"])),
             signals=SignalConfig(snr_values={0: "0", 1: "10"}, num_frames=int(os.environ["AMCX_NFRAMES"]),
                                  frame_size=int(os.environ["AMCX_FS"])))
run_extraction(cfg, verbose=False)
dist.barrier()
dist.destroy_process_group()
print("NCCL_ONE_RANK_DONE")
"""


def test_one_rank_nccl_takes_the_collective_path(tmp_path):
    """What one GPU can exercise of the code the first 8-GPU run will execute.  Every multi-rank run of rounds 1-5 shared
    the box's one device over gloo: the `nccl` branch of sharding.gather_blocks (cuda tensors, dist.gather over RCCL),
    all_gather_rows on device tensors and run_extraction's multi-rank branch under an RCCL process group had never run.
    Here ONE rank under torch.distributed.run with the nccl backend, with the test switch AMCX_TEST_FORCE_COLLECTIVES=1
    (amcpy_amd/sharding.py: the world == 1 shortcuts are not taken): the collectives run over RCCL on device tensors and
    the files equal the plain single-process run's bit for bit (reference: feature_extraction.py:89-97 -- its fork/join).
    Then bench.py as one rank of the launcher: its nccl init, all-reduce, barriers and timing MAX."""
    import socket
    import subprocess
    import sys
    import scipy.io
    from amcpy_amd.config import Config, Paths, SignalConfig
    from amcpy_amd.feature_extraction import run_extraction
    g = load_npz("extract_roundtrip.npz")
    fs, n_frames = int(g["frame_size"]), int(g["n_frames"])
    mods = [str(m) for m in g["mods"]]
    roots = {k: tmp_path / k for k in ("single", "nccl")}
    for root in roots.values():
        cfg = Config(paths=Paths(root=root), signals=SignalConfig(snr_values={0: "0", 1: "10"}, num_frames=n_frames, frame_size=fs))
        cfg.paths.ensure_dirs()
        scipy.io.savemat(str(cfg.paths.mat_data / cfg.paths.mat_filename),
                         {cfg.signals.mat_info[m]: g[f"in_{m}"].astype(np.complex128) for m in mods})
    run_extraction(Config(paths=Paths(root=roots["single"]),
                          signals=SignalConfig(snr_values={0: "0", 1: "10"}, num_frames=n_frames, frame_size=fs)), verbose=False)
    script = tmp_path / "nccl_one_rank.py"
    script.write_text(_NCCL_ONE_RANK_WORKER)

    def free_port():
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            return s.getsockname()[1]

    repo = str(Path(__file__).resolve().parents[1])
    env = dict(_launcher_free_env(), AMCX_REPO=repo, AMCX_ROOT=str(roots["nccl"]), AMCX_NFRAMES=str(n_frames), AMCX_FS=str(fs),
               AMCX_TEST_FORCE_COLLECTIVES="1", PYTHONDONTWRITEBYTECODE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1"]
    r = subprocess.run(launch + ["--master-port", str(free_port()), str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "NCCL_ONE_RANK_DONE" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    cfg = Config(paths=Paths(root=roots["single"]))
    for m in mods:
        a = scipy.io.loadmat(str(roots["single"] / "calculated-features" / f"{m}_features.mat"))
        b = scipy.io.loadmat(str(roots["nccl"] / "calculated-features" / f"{m}_features.mat"))
        key = cfg.signals.mat_info[m]
        assert a[key].shape == b[key].shape == (2, n_frames, 18)
        assert np.array_equal(a[key].view(np.int32), b[key].view(np.int32)), m
    # bench.py as the one rank of an external launcher, RCCL backend (the driver's own command line at N = 1 is launcher-free)
    env_b = dict(_launcher_free_env(), PYTHONDONTWRITEBYTECODE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(launch + ["--master-port", str(free_port()), str(Path(repo) / "bench.py"), "--gpus", "1", "--steps", "3",
                                 "--warmup", "2", "--frames", "128", "--no-cpu-baseline", "--no-h2d", "--no-d2h", "--no-other-configs"],
                       env=env_b, capture_output=True, text=True, timeout=600, cwd=repo)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1 and line["launcher"] == "external/nccl" and line["value"] > 0
    assert line["per_rank"][0]["rank"] == 0 and line["per_rank"][0]["frames"] == 6 * 26 * 128


def test_raw_complex64_stream_file(tmp_path):
    """GNU-Radio style raw complex64 file through the real engine: same floats as the frames
    handed over directly, leading samples skipped, partial last frame dropped."""
    from amcpy_amd import synth
    from amcpy_amd.feature_extraction import extract_raw_stream
    N, F, skip = 1024, 37, 300 * 8
    x = synth.host_block("8PSK", 6.0, F, N, seed=41)
    path = tmp_path / "binary_awgn_8PSK(6)"
    np.concatenate([np.zeros(skip, np.complex64), x.reshape(-1), np.ones(N // 2, np.complex64)]).tofile(path)
    got = extract_raw_stream(path, N, skip_samples=skip)
    assert got.shape == (F, 18)
    assert np.array_equal(got, _run(x, "auto"))
    _assert_parity(got, orc.features18_batch(x), x, "raw stream")


def test_odd_row_stride_and_ragged_counts():
    """Rows that start on 8-byte (not 16-byte) boundaries and frame counts that
    are not a multiple of any chunk size: every frame must still be computed once,
    by both kernels, with identical results to the packed layout."""
    torch = _torch()
    from amcpy_amd.features import features18
    rng = np.random.default_rng(11)
    for N, F in ((128, 2111), (256, 1033), (512, 779), (1024, 1237), (2048, 611), (4096, 205), (8192, 77), (16384, 37), (32768, 19)):
        L = N + 3                                             # odd stride: frames 8-byte aligned only
        x = (rng.standard_normal((F, L)) + 1j * rng.standard_normal((F, L))).astype(np.complex64)
        xd = torch.from_numpy(x).cuda()
        y_strided = features18(xd, frame_size=N, variant="wave").cpu().numpy()
        y_packed = features18(xd[:, :N].contiguous(), variant="wave").cpu().numpy()
        assert np.array_equal(y_strided, y_packed), N
        if N > 8192:                                          # no block kernel up there: the oracle instead
            _assert_parity(y_strided, orc.features18_batch(x[:, :N]), x[:, :N], f"odd stride N={N}")
            continue
        y_block = features18(xd, frame_size=N, variant="block").cpu().numpy()
        S = orc.conditioning_scales(x[:, :N])
        _, scaled = orc.parity_errors(y_strided, y_block, S)
        assert scaled.max() <= TOL, (N, scaled.max(axis=0))
        assert np.isfinite(y_strided).all()


def test_concurrent_streams_are_independent():
    """The library keeps no global mutable state: launches on two streams with
    different inputs give the same answers as serial launches."""
    torch = _torch()
    from amcpy_amd.features import features18
    from amcpy_amd import synth
    a = torch.from_numpy(synth.host_block("8PSK", 5.0, 3000, 2048, seed=31)).cuda()
    b = torch.from_numpy(synth.host_block("64QAM", 15.0, 3000, 2048, seed=32)).cuda()
    ya, yb = features18(a), features18(b)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for _ in range(3):
        with torch.cuda.stream(s1):
            oa = features18(a)
        with torch.cuda.stream(s2):
            ob = features18(b)
        outs.append((oa, ob))
    torch.cuda.synchronize()
    for oa, ob in outs:
        assert torch.equal(oa, ya) and torch.equal(ob, yb)


def _near_pi_tie_frames(N, n_frames, seed):
    """Noisy frames with ~40 planted phase steps that are antiparallel to within +-1 or
    +-2 fp32 ulps of one component: the wrapped step is +-pi to ~1e-8..1e-7 rad, which
    fp32 angles cannot resolve but the fp64 reference does (from the same fp32 samples).
    (Exactly antiparallel pairs are left out: there the reference's own sign hangs on the
    1e-16 rounding of np.angle -- unless both angles are exact, as in the axis-aligned case
    checked separately.)"""
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((n_frames, N)) + 1j * rng.standard_normal((n_frames, N))).astype(np.complex64)
    for f in range(n_frames):
        pos = rng.choice(np.arange(2, N - 2, 3), size=40, replace=False)
        for n in pos:
            re, im = np.float32(x[f, n].real), np.float32(x[f, n].imag)
            k = int(rng.choice([-2, -1, 1, 2]))
            scale = np.float32(2.0 ** int(rng.integers(-1, 2)))
            re2 = np.float32(-re * scale)
            im2 = np.float32(-im * scale)
            for _ in range(abs(k)):                       # nudge the imaginary part by |k| ulps
                im2 = np.nextafter(im2, np.float32(np.inf if k > 0 else -np.inf), dtype=np.float32)
            x[f, n + 1] = re2 + 1j * im2
    return x


def test_phase_steps_within_an_ulp_of_pi():
    """Frames the fast kernel flags (a step within kTieBand of +-pi) get f5/f9 from the
    exact-sign fix-up launch; the block kernel decides exactly in place.  Both must follow
    the fp64 reference, including exact ties (numpy's unwrap rule)."""
    for N in (128, 256, 512, 1024, 2048, 4096, 8192):
        x = _near_pi_tie_frames(N, 24, seed=1234 + N)
        th = np.angle(x.astype(np.complex128))
        d = np.abs(np.abs(np.diff(th, axis=-1)) - np.pi)
        assert (d < 2.5e-7).sum() >= 24 * min(30, N // 8), "fixture lost its near-ties"
        gold = orc.features18_batch(x)
        for variant in _variants_for(N):
            got = _run(x, variant)
            assert (got[:, 4] > 0).all(), "tie flag (negative f5) leaked to the caller"
            _assert_parity(got, gold, x, f"near-pi ties N={N} {variant}")
    # exact antiparallel neighbours everywhere: every step is a true tie
    alt = np.tile(np.array([1 + 1j, -1 - 1j], dtype=np.complex64), 1024)[None, :]
    gold = orc.features18_batch(alt)
    for variant in _variants_for(2048):
        got = _run(alt, variant)
        assert np.isclose(got[0, 4], gold[0, 4], rtol=1e-5), (variant, got[0, 4], gold[0, 4])


def test_iq_pair_layout_is_zero_copy():
    """RadioML-style (F, N, 2) float32 input: same bits as
    complex64, so the result equals the complex path exactly and no copy is made."""
    torch = _torch()
    from amcpy_amd.features import features18, features18_iq_pairs
    from amcpy_amd import synth
    x = synth.host_block("16QAM", 6.0, 64, 1024, seed=77)
    pairs = torch.from_numpy(np.stack([x.real, x.imag], axis=-1).astype(np.float32)).cuda()
    a = features18_iq_pairs(pairs)
    b = features18(torch.from_numpy(x).cuda())
    assert torch.equal(a, b)
    with pytest.raises(TypeError):
        features18_iq_pairs(pairs.double())
    with pytest.raises(ValueError):
        features18_iq_pairs(pairs.transpose(1, 2).contiguous().transpose(1, 2))


@pytest.mark.parametrize("n_mods,N,n_frames,label",
                         [(6, 2048, 4096, "configs[1]"), (6, 4096, 4096, "configs[2]"),
                          (6, 2048, 8192, "configs[3] per-GPU shard"), (3, 1024, 4096, "configs[4] per-GPU shard")])
def test_full_benchmark_shard_properties(n_mods, N, n_frames, label):
    """BASELINE configs at full size -- configs[1] 6 x 26 x 4096 frames x 2048 samples (10.5 GB
    in HBM), configs[2] the same at 4096 samples (20.9 GB), and one GPU's eighth of configs[4]
    (24 x 26 x 4096 x 1024 over 8 GPUs = 3 modulations each, 2.6 GB) and of configs[3]
    (6 x 26 x 65536 x 2048 frame-sharded over 8 GPUs = 8192 frames per (mod, SNR), 20.9 GB):
    too large for the oracle,
    so size-independent properties --
    every frame is computed exactly once and independently of its position (one launch
    over the whole shard == per-modulation launches == a gathered sample recomputed
    alone, bit for bit), per-block checksums agree, the output is finite, and a sample of
    1 024 frames (64 until round 5) matches the oracle: every frame within 1e-5 of max(|value|, S)."""
    torch = _torch()
    from amcpy_amd import synth
    from amcpy_amd.features import features18
    n_snr = 26
    arena = torch.empty((n_mods, n_snr, n_frames, N), dtype=torch.complex64, device="cuda")
    for mi, mod in enumerate(synth.MODS6[:n_mods]):
        synth.device_frames(mod, n_snr, n_frames, N, device="cuda", rank=0, mod_idx=mi, out=arena[mi])
    whole = features18(arena)
    assert whole.shape == (n_mods, n_snr, n_frames, 18)
    assert torch.isfinite(whole).all()
    # different batching, same bits; block checksums (a checksum of checksums) agree
    for mi in range(n_mods):
        part = features18(arena[mi])
        assert torch.equal(part, whole[mi]), f"modulation {mi}: result depends on the batching"
    csum_whole = whole.double().sum(dim=2)                        # (n_mods, 26, 18)
    rev = arena.flip(2).contiguous()
    csum_rev = features18(rev).double().sum(dim=2)
    del rev
    assert torch.allclose(csum_whole, csum_rev, rtol=1e-12, atol=0)
    # a gathered random sample, recomputed alone and checked against the oracle
    g = torch.Generator(device="cpu").manual_seed(5)
    idx = torch.randint(0, n_mods * n_snr * n_frames, (1024,), generator=g)
    flat = arena.reshape(-1, N)
    sample = flat[idx.cuda()].contiguous()
    alone = features18(sample)
    assert torch.equal(alone, whole.reshape(-1, 18)[idx.cuda()])
    x = sample.cpu().numpy()
    _assert_parity(alone.cpu().numpy(), orc.features18_batch(x), x, f"sample of the full shard, {label}")
    # SNR trend sanity on the signal classes: mean |x| falls towards 1 as noise vanishes
    assert (whole[0, 0, :, 5].mean() > whole[0, -1, :, 5].mean())


def test_complex128_host_entry_chunks_and_rounds_on_device():
    """MATLAB-double containers are rounded to complex64 exactly as numpy's astype would (by the staging
    threads on their way to pinned memory); hundreds of megabytes in many chunks, rows longer than the frame."""
    from amcpy_amd.features import features18_host
    rng = np.random.default_rng(8)
    F, L, N = 9000, 4100, 4096                       # 590 MB of complex128 -> two chunks
    x = rng.standard_normal((F, L)) + 1j * rng.standard_normal((F, L))
    got = features18_host(x, frame_size=N)
    want = features18_host(x.astype(np.complex64), frame_size=N)
    assert np.array_equal(got, want)
    assert np.isfinite(got).all()


def test_launch_is_graph_capturable():
    """The launch path makes no allocation or synchronisation, so after one warm call
    (which sets the kernel's LDS attribute) it can be captured in a HIP graph and
    replayed (N = 2048: the one launch of the wave kernel)."""
    torch = _torch()
    from amcpy_amd.features import features18
    from amcpy_amd import synth
    x = torch.from_numpy(synth.host_block("QPSK", 8.0, 500, 2048, seed=3)).cuda()
    out = torch.empty((500, 18), dtype=torch.float32, device="cuda")
    want = features18(x).clone()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        features18(x, out=out)                       # warm on the capture stream
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            features18(x, out=out)
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)


@pytest.mark.parametrize("N", [2048, 4096, 8192, 10000, 16384, 32768])
def test_dynamic_range_matches_the_float32_stored_reference(N):
    """Frames of ordinary shape at scales 1e-12 ... 1e12, one whose halves differ by ten orders
    of magnitude and one with a single 5e7 sample (tests/golden/range_n{N}.npz, captured from
    the reference; N = 4096 and 8192 since round 4: the one-wave kernel's and the quad kernel's own re-runs).  The reference evaluates in complex128 and stores float32
    (features.py:46-58, feature_extraction.py:35,56): its sixth-order cumulants are inf above
    |x| ~ 2.6e6 and 0 below ~ 3e-8 while everything else stays finite.  Both variants must
    reproduce that pattern exactly and every finite value within the usual tolerance -- the wave
    kernel by flagging what its fp32 sums cannot hold for the fp64-sum fix-up."""
    g = load_npz(f"range_n{N}.npz")
    x, names = g["iq"], [str(n) for n in g["names"]]
    gold32, gold64 = g["golden64"], g["golden64_f64"]
    S = orc.conditioning_scales(x.astype(np.complex128))
    with np.errstate(all="ignore"):
        stored = gold64.astype(np.float32)
    assert np.array_equal(stored, gold32, equal_nan=True)
    strict = [i for i in range(18) if i < 9 or i == 10]
    for variant in _variants_for(N):
        got = _run(x, variant)
        special = ~np.isfinite(gold32) | (gold32 == 0)
        bad = np.argwhere(special & (got != gold32))
        assert bad.size == 0, (variant, [(names[i], j + 1, got[i, j], gold32[i, j]) for i, j in bad[:6]])
        assert np.isfinite(got[~special]).all(), variant
        with np.errstate(all="ignore"):
            diff = np.abs(got.astype(np.float64) - gold32.astype(np.float64))
            # float32 denormals of the reference's store carry fewer bits: allow their quantum
            quantum = np.where(np.abs(gold32) < 1.2e-38, 1.5e-45, 0.0)
            scaled = (diff - quantum).clip(min=0) / np.maximum(np.abs(gold32.astype(np.float64)), S)
            plain = (diff - quantum).clip(min=0) / np.abs(gold32.astype(np.float64))
        scaled[special] = 0.0
        plain[special] = 0.0
        worst = scaled.max(axis=0)
        print(f"\n[range {variant}] worst scaled rel per feature:", " ".join(f"{v:.1e}" for v in worst))
        i, j = np.unravel_index(scaled.argmax(), scaled.shape)
        assert worst.max() <= TOL, (variant, names[i], j + 1, got[i, j], gold32[i, j])
        assert plain[:, strict].max() <= TOL, variant


def test_run_extraction_on_a_container_of_genuine_doubles(tmp_path):
    """The reference evaluates MATLAB doubles in complex128 (feature_extraction.py:46-48,68);
    the engine rounds them to complex64 on the GPU first.  Fixture: a container whose samples
    are NOT float32-representable and the six files the reference's own run_extraction wrote
    for it (tests/golden/extract_roundtrip_f64.npz).  The rounding must stay inside the
    tolerance of SURVEY.md section 8c."""
    import scipy.io
    from amcpy_amd.config import Config, Paths, SignalConfig
    from amcpy_amd.feature_extraction import run_extraction
    g = load_npz("extract_roundtrip_f64.npz")
    fs, n_frames = int(g["frame_size"]), int(g["n_frames"])
    mods = [str(m) for m in g["mods"]]
    cfg = Config(paths=Paths(root=tmp_path),
                 signals=SignalConfig(snr_values={0: "0", 1: "10"}, num_frames=n_frames, frame_size=fs))
    cfg.paths.ensure_dirs()
    container = {cfg.signals.mat_info[m]: g[f"in_{m}"] for m in mods}
    assert all(v.dtype == np.complex128 and not np.array_equal(v, v.astype(np.complex64)) for v in container.values())
    scipy.io.savemat(str(cfg.paths.mat_data / cfg.paths.mat_filename), container)
    run_extraction(cfg, verbose=False)
    for m in mods:
        d = scipy.io.loadmat(str(cfg.paths.calculated_features / f"{m}_features.mat"))
        arr = d[cfg.signals.mat_info[m]]
        assert arr.dtype == np.float32 and arr.shape == (2, n_frames, 18)
        x = g[f"in_{m}"][:, :, :fs].reshape(-1, fs)                 # complex128: the scales of the true input
        _assert_parity(arr.reshape(-1, 18), g[f"out_{m}"].reshape(-1, 18).astype(np.float64), x,
                       f"genuine doubles {m}")


def test_run_extraction_on_a_mat73_container_equals_the_level5_one(tmp_path):
    """A MATLAB -v7.3 (HDF5) container through the engine: contiguous variables are read from the file by the staging
    threads as interleaved complex128 (H5Dget_offset + pread), chunked / compressed ones decoded by libhdf5 first.
    The six feature files must be BIT-IDENTICAL to those of a level-5 container of the same arrays -- same doubles,
    same rounding to complex64 on the way up, same kernel -- in one process and over the fan-out (--devices 0,0)."""
    import shutil
    import scipy.io
    from amcpy_amd import hdf5_min
    from amcpy_amd.config import Config, Paths, SignalConfig
    from amcpy_amd.feature_extraction import release_engines, run_extraction
    from tests.test_host_cpu import _mat73_variable
    if not hdf5_min.available():
        pytest.skip("no HDF5 C library on this machine")
    outs = {}
    for kind in ("v73", "v5", "v73-fanout"):
        cfg = Config(paths=Paths(root=tmp_path / kind),
                     signals=SignalConfig(snr_values={0: "0", 1: "10"}, num_frames=5, frame_size=256))
        cfg.paths.ensure_dirs()
        target = cfg.paths.mat_data / cfg.paths.mat_filename
        if kind != "v5":
            shutil.copy(REPO / "tests" / "golden" / "mat73_like.mat", target)
        else:
            scipy.io.savemat(str(target), {cfg.signals.mat_info[m]: _mat73_variable(i, m)
                                           for i, m in enumerate(cfg.signals.modulations_with_noise)})
        run_extraction(cfg, verbose=False, **({"devices": [0, 0]} if kind == "v73-fanout" else {}))
        outs[kind] = {m: scipy.io.loadmat(str(cfg.paths.calculated_features / f"{m}_features.mat"))[cfg.signals.mat_info[m]]
                      for m in cfg.signals.modulations_with_noise}
    release_engines()
    for m, a in outs["v5"].items():
        assert a.shape == (2, 5, 18) and a.dtype == np.float32 and np.isfinite(a).all()
        for kind in ("v73", "v73-fanout"):
            assert np.array_equal(a.view(np.int32), outs[kind][m].view(np.int32)), (kind, m)
    # and against the oracle on the frames themselves
    for i, m in enumerate(("BPSK", "WGN")):
        mi = list(Config().signals.modulations_with_noise).index(m)
        x = _mat73_variable(mi, m)[:, :, :256].reshape(-1, 256)
        _assert_parity(outs["v73"][m].reshape(-1, 18), orc.features18_batch(x), x, f"mat73 {m}")


@pytest.mark.parametrize("fixture", ["configs0_reference_run.npz", "configs2_reference_run.npz", "configs4_reference_run.npz"])
def test_run_extraction_against_the_references_own_run_of_configs0(tmp_path, fixture):
    """BASELINE configs[0] end to end against the REFERENCE ITSELF: tests/golden/configs0_reference_run.npz holds what the
    reference's run_extraction (feature_extraction.py:85-99) wrote for 6 modulations x 2 SNR x 500 frames x 2048 samples
    of MATLAB doubles (oracle/capture_golden.py configs0; 16.7 s there).  The same container -- regenerated from its
    seeds, SHA-256 checked -- through this package's run_extraction: same files, same keys, EVERY one of the 6 000 rows
    within the parity bounds of the reference's stored float32 (no frame excepted: WGN frame 106, whose sixth-order
    moment cancels to 0.003 of mean |x|^6 and which rounds 1-5 whitelisted at 1.43e-5, gets fp64 sums in the kernel).  configs2_reference_run.npz: the same at BASELINE
    configs[2]'s frame size, 6 x 2 x 50 x 4096."""
    import hashlib
    import scipy.io
    from amcpy_amd import synth
    from amcpy_amd.config import Config, Paths, SignalConfig
    from amcpy_amd.feature_extraction import run_extraction
    g = load_npz(fixture)
    n_snr, n_frames, fs = int(g["n_snr"]), int(g["n_frames"]), int(g["frame_size"])
    blocks = synth.host_frames(synth.MODS6, n_snr, n_frames, fs)
    for m in synth.MODS6:
        assert hashlib.sha256(np.ascontiguousarray(blocks[m]).tobytes()).hexdigest() == str(g[f"sha256_in_{m}"]), m
    cfg = Config(paths=Paths(root=tmp_path),
                 signals=SignalConfig(snr_values={0: "0", 1: "10"}, num_frames=n_frames, frame_size=fs))
    cfg.paths.ensure_dirs()
    scipy.io.savemat(str(cfg.paths.mat_data / cfg.paths.mat_filename),
                     {cfg.signals.mat_info[m]: blocks[m].astype(np.complex128) for m in synth.MODS6})
    run_extraction(cfg, verbose=False)
    for m in synth.MODS6:
        d = scipy.io.loadmat(str(cfg.paths.calculated_features / f"{m}_features.mat"))
        assert sorted(k for k in d if not k.startswith("__")) == sorted(["Modulation", cfg.signals.mat_info[m]])
        arr = d[cfg.signals.mat_info[m]]
        want = g[f"out_{m}"]
        assert arr.dtype == np.float32 and arr.shape == want.shape == (n_snr, n_frames, 18)
        x = blocks[m].reshape(-1, fs)
        _assert_parity(arr.reshape(-1, 18), want.reshape(-1, 18).astype(np.float64), x,
                       f"run_extraction vs the reference's run, {m}")


def test_extract_cli_on_the_configs0_shape(tmp_path):
    """`python -m amcpy_amd extract` as a subprocess on BASELINE configs[0]: 6 modulations x 2 SNR
    x 500 frames x 2048 samples in mat-data/all_modulations.mat -> six {mod}_features.mat, checked
    against the oracle (reference CLI: main.py:32,85-87,160-175).  Then the SAME command over several devices --
    `--devices 0,0`: two engines inside the one process, frames cut across them -- and as two ranks of an external
    launcher (torch.distributed.run; every rank takes its LOCAL_RANK's GPU and joins the process group itself):
    both must write files bit-identical to the single-device run."""
    import socket
    import subprocess
    import sys
    import scipy.io
    from amcpy_amd import synth
    from amcpy_amd.config import Config, Paths, SignalConfig
    n_snr, n_frames, fs = 2, 500, 2048
    cfg = Config(paths=Paths(root=tmp_path),
                 signals=SignalConfig(snr_values={0: "0", 1: "10"}, num_frames=n_frames, frame_size=fs))
    cfg.paths.ensure_dirs()
    blocks = synth.host_frames(synth.MODS6, n_snr, n_frames, fs)             # complex64, seeds of SURVEY 8d
    scipy.io.savemat(str(cfg.paths.mat_data / cfg.paths.mat_filename),
                     {cfg.signals.mat_info[m]: blocks[m] for m in synth.MODS6})
    repo = str(Path(__file__).resolve().parents[1])
    env = _launcher_free_env()
    env["PYTHONPATH"] = repo + os.pathsep + os.environ.get("PYTHONPATH", "")
    base = [sys.executable, "-m", "amcpy_amd", "extract", "--root", str(tmp_path),
            "--num-frames", str(n_frames), "--frame-size", str(fs), "--snr-values", "0", "10"]
    r = subprocess.run(base, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "All feature calculations complete!" in r.stdout

    def read_all():
        out = {}
        for m in synth.MODS6:
            d = scipy.io.loadmat(str(cfg.paths.calculated_features / f"{m}_features.mat"))
            assert sorted(k for k in d if not k.startswith("__")) == sorted(["Modulation", cfg.signals.mat_info[m]])
            out[m] = d[cfg.signals.mat_info[m]]
            (cfg.paths.calculated_features / f"{m}_features.mat").unlink()
        return out

    single = read_all()
    for m in synth.MODS6:
        arr = single[m]
        assert arr.dtype == np.float32 and arr.shape == (n_snr, n_frames, 18)
        x = blocks[m].reshape(-1, fs)
        _assert_parity(arr.reshape(-1, 18), orc.features18_batch(x), x, f"CLI extract {m}")
    # several devices from the one process
    r = subprocess.run(base + ["--devices", "0,0"], env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    fanned = read_all()
    for m in synth.MODS6:
        assert np.array_equal(single[m].view(np.int32), fanned[m].view(np.int32)), m
    # two ranks of an external launcher (gloo + one shared GPU on this box; RCCL and one GPU each on a node)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "amcpy_amd", *base[3:]]
    r = subprocess.run(cmd, env=dict(env, AMCX_SHARE_GPU="1", AMCX_DIST_BACKEND="gloo"), cwd=str(tmp_path),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    ranked = read_all()
    for m in synth.MODS6:
        assert np.array_equal(single[m].view(np.int32), ranked[m].view(np.int32)), m
    # the same launch WITHOUT saying that the GPU is shared is refused on every rank (one GPU on this box)
    if _torch().cuda.device_count() == 1:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            cmd[cmd.index("--master-port") + 1] = str(sk.getsockname()[1])
        r = subprocess.run(cmd, env=dict(env, AMCX_DIST_BACKEND="gloo"), cwd=str(tmp_path),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode != 0 and "no GPU 1" in (r.stdout + r.stderr), r.stdout[-2000:] + r.stderr[-2000:]
    bad = subprocess.run([sys.executable, "-m", "amcpy_amd", "extract", "--root", str(tmp_path / "nowhere")],
                         env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0                                                # a missing container is an error


def test_calculate_features_in_a_loop_reuses_its_context():
    """The reference's usage pattern -- calculate_features once per frame (feature_extraction.py:30-39)
    -- must not pay allocation per call: the per-thread context (amcx_ctx_*) keeps stream and
    scratch.  Same values as the batch entry, and a loop of 200 calls stays under 50 ms each."""
    import time
    from amcpy_amd import synth
    from amcpy_amd.features import calculate_features, features18_host
    x = synth.host_block("16QAM", 8.0, 8, 2048, seed=77)
    batch = features18_host(x)
    calculate_features(range(1, 19), x[0])                                    # warm: context + kernels
    t0 = time.perf_counter()
    for k in range(200):
        row = calculate_features(range(1, 19), x[k % 8])
        assert np.array_equal(np.float32(row), batch[k % 8])
    per_call = (time.perf_counter() - t0) / 200
    print(f"\ncalculate_features per call: {per_call * 1e6:.0f} us")
    assert per_call < 50e-3
    # complex128 frames take the double entry of the same context
    row = calculate_features([6, 11], x[0].astype(np.complex128) * (1 + 1e-9))
    assert np.allclose(row, [batch[0][5], batch[0][10]], rtol=1e-6)


def test_scale_sweep_wave_equals_block_across_the_range_threshold():
    """The same frames at scales 1e-9 ... 1e9 (powers of ten, so not exact in binary): every
    feature follows its scaling law, and the wave variant -- fp32 sums inside 1e-10 <= mean|x|^2
    <= 1e10, the fp64-sum fix-up outside -- agrees with the block variant on both sides of the
    two thresholds (1e-5 and 1e5 in amplitude) and right at them."""
    from amcpy_amd import synth
    base = np.concatenate([synth.host_block(m, 6.0, 2, 2048, seed=321 + i) for i, m in enumerate(synth.MODS6)])
    order = np.array([2, 0, 0, 0, 0, 1, 0.5, 0, 0, 2, 2, 4, 4, 4, 6, 6, 6, 6])      # feature j scales as s**order[j]
    ref = _run(base.astype(np.complex64), "block").astype(np.float64)
    S = orc.conditioning_scales(base.astype(np.complex128))
    for k in range(-9, 10):
        s = 10.0 ** k
        x = (base.astype(np.complex128) * s).astype(np.complex64)
        wave = _run(x, "wave").astype(np.float64)
        block = _run(x, "block").astype(np.float64)
        with np.errstate(all="ignore"):
            law = s ** order
            want = (ref * law).astype(np.float32).astype(np.float64)          # incl. float32 overflow / underflow
            tol = 3e-5 * np.maximum(np.abs(ref), S) * law + 1.5e-45
        for name, got in (("wave", wave), ("block", block)):
            fin = np.isfinite(want) & (np.abs(want) > 1e-37)
            assert np.array_equal(np.isinf(got), np.isinf(want)), (name, k)
            with np.errstate(invalid="ignore"):
                bad = np.argwhere(fin & (np.abs(got - want) > tol))
            assert bad.size == 0, (name, k, bad[:4], got[tuple(bad[0])], want[tuple(bad[0])])
        both = np.isfinite(wave) & np.isfinite(block)
        assert np.all(np.abs(wave[both] - block[both]) <= tol[both]), k


def test_plain_c_client_computes_the_same_features(tmp_path):
    """tests/c_abi/abi_check.c through the host-buffer entry and through a reusable context
    (bit-identical to each other, checked in C) equals the Python binding's result for the same
    frame -- the boundary needs neither Python nor torch."""
    import subprocess
    from amcpy_amd.features import features18_host
    from tests.test_host_cpu import _build_abi_check
    exe = _build_abi_check(tmp_path)
    for n in (2048, 1000):
        k = np.arange(n, dtype=np.float64)
        ph, a = 0.37 * k + 0.0009 * k * k, 1.0 + 0.25 * np.sin(0.05 * k)       # a chirp with amplitude ripple
        x = ((a * np.cos(ph)).astype(np.float32) + 1j * (a * np.sin(ph)).astype(np.float32)).astype(np.complex64)
        path = tmp_path / f"frame_{n}.c64"
        x.tofile(path)
        r = subprocess.run([str(exe), "compute", str(n), str(path)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        got = np.array([float(v) for v in r.stdout.split()], dtype=np.float32)     # %.9g round-trips a float32
        want = features18_host(x[None, :])[0]
        assert got.shape == (18,) and np.array_equal(got, want), (n, got, want)


def test_upload_pipeline_many_chunks_equals_one_launch():
    """HipEngine (amcx_ctx_features18_strided_host) with pinned slots far smaller than the data (dozens of
    trips round its three pinned / two device slots, ragged last chunk): from a Fortran-ordered complex128
    container with more frames and longer rows than the configuration uses (sample planes + device
    transposition), from the same container in C order (rows), as two real arrays (a memory-mapped .mat),
    rounded on the host or on the device, from a complex64 array, and twice in a row (slots reused across
    calls): every row equals the one-launch result on the same frames, bit for bit."""
    torch = _torch()
    from amcpy_amd.feature_extraction import FrameRows, HipEngine, SplitComplex
    from amcpy_amd.features import features18
    rng = np.random.default_rng(17)
    n_snr, n_frames, L, N = 3, 211, 300, 256
    full = rng.standard_normal((n_snr, n_frames + 5, L)) + 1j * rng.standard_normal((n_snr, n_frames + 5, L))
    parsed = np.asfortranarray(full)                                  # as scipy.io.loadmat returns it
    flat = full[:, :n_frames, :N].reshape(-1, N)
    want128 = features18(torch.from_numpy(flat).cuda().to(torch.complex64)).cpu().numpy()
    F = n_snr * n_frames
    eng = HipEngine(N, chunk_bytes=7 * F * 8, threads=4)              # 7 planes per slot: ramp 1, 1, 1, 3, then 7s
    rows = FrameRows(parsed, n_snr, n_frames)
    for _ in range(2):
        got = eng(rows)
        assert eng.stats["chunks"] >= 30 and eng.stats["plane_major"] == 1 and got.shape == (F, 18)
        assert eng.stats["pcie_bytes"] == F * N * 8 and eng.stats["source_bytes"] == F * N * 16
        assert np.array_equal(got, want128, equal_nan=True)
    part = eng(rows.slice(100, 433))                                  # a rank's contiguous range: three rectangles
    assert np.array_equal(part, want128[100:433], equal_nan=True)
    dev_round = HipEngine(N, chunk_bytes=5 * F * 16, threads=2, round_on_device=True)
    assert np.array_equal(dev_round(rows), want128, equal_nan=True) and dev_round.stats["pcie_bytes"] == F * N * 16
    split = SplitComplex(np.asfortranarray(full.real), np.asfortranarray(full.imag))        # as matfile.py maps it
    assert np.array_equal(eng(FrameRows(split, n_snr, n_frames)), want128, equal_nan=True)
    # C order: rows go up in chunks of whole frames, the kernel runs per chunk
    row_eng = HipEngine(N, chunk_bytes=37 * N * 8, threads=3)
    got = row_eng(FrameRows(full, n_snr, n_frames))
    assert row_eng.stats["plane_major"] == 0 and row_eng.stats["chunks"] >= 16
    assert np.array_equal(got, want128, equal_nan=True)
    assert np.array_equal(HipEngine(N, chunk_bytes=29 * N * 24, round_on_device=True)(FrameRows(full, n_snr, n_frames)),
                          want128, equal_nan=True)
    x64 = flat.astype(np.complex64)
    want64 = features18(torch.from_numpy(x64).cuda()).cpu().numpy()
    got64 = HipEngine(N, chunk_bytes=50 * N * 8)(x64)                 # complex64 goes up as it is
    assert np.array_equal(got64, want64, equal_nan=True)
    assert np.array_equal(want64, want128, equal_nan=True)            # host / GPU rounding of doubles == numpy's astype
    assert np.array_equal(eng(x64[::-1]), want64[::-1], equal_nan=True)       # negative stride: copied first
    # a layout with no contiguous axis, and an integer container: one host copy, then the row path
    strided = np.asfortranarray(np.repeat(full, 2, axis=0))[::2]
    assert np.array_equal(eng(FrameRows(strided[:, :, :], n_snr, n_frames)), want128, equal_nan=True)
    ints = np.round(full.real * 100).astype(np.int16)
    want_int = features18(torch.from_numpy(ints[:, :n_frames, :N].reshape(-1, N).astype(np.complex64)).cuda()).cpu().numpy()
    assert np.array_equal(eng(FrameRows(np.asfortranarray(ints), n_snr, n_frames)), want_int, equal_nan=True)
    assert HipEngine(N)(x64[:0]).shape == (0, 18)


def test_iq_pair_dataset_through_the_engine():
    """extract_iq_pairs on a RadioML-shaped (F, 1024, 2) float32 array (the configs[4] frame length): equal,
    bit for bit, to handing the same frames over as complex64 -- and to the zero-copy device view."""
    torch = _torch()
    from amcpy_amd import synth
    from amcpy_amd.feature_extraction import extract_iq_pairs
    from amcpy_amd.features import features18, features18_iq_pairs
    x = np.concatenate([synth.host_block(m, 2.0, 40, 1024, seed=900 + i) for i, m in enumerate(synth.MODS6)])
    pairs = np.ascontiguousarray(np.stack([x.real, x.imag], axis=-1).astype(np.float32))     # (240, 1024, 2)
    want = features18(torch.from_numpy(x).cuda()).cpu().numpy()
    assert np.array_equal(extract_iq_pairs(pairs), want)
    assert np.array_equal(extract_iq_pairs(pairs, first_frame=17, max_frames=100), want[17:117])
    assert np.array_equal(features18_iq_pairs(torch.from_numpy(pairs).cuda()).cpu().numpy(), want)
    _assert_parity(want, orc.features18_batch(x), x, "RadioML-shaped (I, Q) pairs, N = 1024")


def test_radioml_hdf5_container_through_the_engine():
    """A genuine HDF5 container in RadioML 2018.01A's layout, written by the real h5py (tests/golden/radioml_like.h5: X
    float32 (24, 1024, 2) chunked + shuffled + gzip; reference old/dataset.py:43-56), through extract_radioml_hdf5 -- h5py
    where it is importable, else libhdf5 itself through amcpy_amd.hdf5_min: bit for bit what the same (I, Q) pairs give as a
    device tensor, at the full frame length and a shorter one, and within the parity bounds of the oracle."""
    torch = _torch()
    import json
    from amcpy_amd import hdf5_min
    from amcpy_amd.feature_extraction import extract_radioml_hdf5
    from amcpy_amd.features import features18_iq_pairs
    try:
        import h5py  # noqa: F401
    except ImportError:
        if not hdf5_min.available():
            pytest.skip("neither h5py nor an HDF5 C library on this machine")
    path = REPO / "tests" / "golden" / "radioml_like.h5"
    meta = json.loads((REPO / "tests" / "golden" / "radioml_like.json").read_text())
    with hdf5_min.File(path) as fh:
        pairs = fh["X"][:]
    import hashlib
    assert hashlib.sha256(pairs.tobytes()).hexdigest() == meta["sha256"]["X"]
    x = np.ascontiguousarray(pairs[..., 0] + 1j * pairs[..., 1]).astype(np.complex64)
    got = extract_radioml_hdf5(path, chunk_frames=7)                      # several chunks, the last one ragged
    assert got.shape == (24, 18) and got.dtype == np.float32
    assert np.array_equal(got, features18_iq_pairs(torch.from_numpy(pairs).cuda()).cpu().numpy())
    _assert_parity(got, orc.features18_batch(x), x, "RadioML-shaped HDF5 container, N = 1024")
    half = extract_radioml_hdf5(path, frame_size=512, first_frame=5, max_frames=11)
    assert np.array_equal(half, _run(np.ascontiguousarray(x[5:16, :512]), "auto"))


def test_contiguous_hdf5_dataset_is_read_by_the_staging_threads(tmp_path):
    """A CONTIGUOUS X (h5py's default layout, no chunks or filters) is a raw complex64 stream at the file offset libhdf5
    reports: extract_radioml_hdf5 hands that to the engine's staging threads (pread into the pinned slots; the library is
    not on the data path).  Same 18 floats as the device-tensor path, for the whole set and for a frame range; a shorter
    frame length goes through the library's own reads and agrees too."""
    torch = _torch()
    from amcpy_amd import hdf5_min, synth
    from amcpy_amd.feature_extraction import extract_radioml_hdf5
    from amcpy_amd.features import features18
    if not hdf5_min.available():
        pytest.skip("no HDF5 C library on this machine")
    x = np.concatenate([synth.host_block(m, 6.0, 50, 1024, seed=700 + i) for i, m in enumerate(synth.MODS6)])     # 300 frames
    pairs = np.ascontiguousarray(np.stack([x.real, x.imag], axis=-1).astype(np.float32))
    path = tmp_path / "contiguous.h5"
    with hdf5_min.File(path, "w") as fh:
        fh.create_dataset("Y", np.arange(7, dtype=np.int64))           # something in front of X in the file
        fh.create_dataset("X", pairs)
    with hdf5_min.File(path) as fh:
        assert fh["X"].file_offset is not None
    want = features18(torch.from_numpy(x).cuda()).cpu().numpy()
    assert np.array_equal(extract_radioml_hdf5(path), want)
    assert np.array_equal(extract_radioml_hdf5(path, first_frame=37, max_frames=201), want[37:238])
    assert extract_radioml_hdf5(path, first_frame=300).shape == (0, 18)
    assert np.array_equal(extract_radioml_hdf5(path, frame_size=256, max_frames=40), _run(np.ascontiguousarray(x[:40, :256]), "auto"))


@pytest.mark.parametrize("N", [128, 256, 512, 1024, 2048, 4096, 8192])
def test_range_pass_mixed_batches_and_scaling_laws(N):
    """The wave kernel's re-run of out-of-range frames (inside the kernel, on a power-of-two pre-scaled copy):
    in a batch where some frames are in range and others are scaled by 2^30 or 2^-40 (out of range both
    ways), every frame's 18 floats equal what the frame yields alone -- flagged and unflagged frames
    share 8..64-frame scan blocks, with ragged counts -- and a scaled frame equals its in-range
    original through the scaling laws (an exact power of two changes no rounding: at most the final
    float32 store differs, by its overflow / underflow)."""
    torch = _torch()
    from amcpy_amd import synth
    from amcpy_amd.features import features18
    base = np.concatenate([synth.host_block(m, 4.0, 7, N, seed=555 + i) for i, m in enumerate(synth.MODS6)])   # 42 frames
    rng = np.random.default_rng(N)
    kinds = rng.integers(0, 3, size=201)                       # 0: as is, 1: x 2^30, 2: x 2^-40
    pick = rng.integers(0, base.shape[0], size=201)
    scale = np.array([1.0, 2.0 ** 30, 2.0 ** -40])[kinds]
    batch = (base[pick].astype(np.complex128) * scale[:, None]).astype(np.complex64)       # exact: powers of two
    got = features18(torch.from_numpy(batch).cuda()).cpu().numpy()
    ref = features18(torch.from_numpy(base).cuda()).cpu().numpy().astype(np.float64)
    order = np.array([2, 0, 0, 0, 0, 1, 0.5, 0, 0, 2, 2, 4, 4, 4, 6, 6, 6, 6])
    for i in range(0, 201, 5):                                 # alone == in the batch, bit for bit
        alone = features18(torch.from_numpy(batch[i:i + 1]).cuda()).cpu().numpy()[0]
        assert np.array_equal(alone, got[i], equal_nan=True), (N, i, kinds[i])
    with np.errstate(all="ignore"):
        want = (ref[pick] * scale[:, None] ** order[None, :]).astype(np.float32)
    same = np.isfinite(want) & (np.abs(want) > 1.2e-38)        # normal floats: the laws are exact
    assert np.array_equal(np.isinf(got), np.isinf(want))
    assert np.allclose(got[same], want[same], rtol=2.5e-7, atol=0), N
    assert np.all(got[kinds == 0] == ref[pick][kinds == 0].astype(np.float32))


@pytest.mark.parametrize("N", [512, 1024, 2048, 8192])
def test_out_of_range_frames_scattered_over_a_full_grid(N):
    """The in-kernel re-run under a FULL grid: 40 000 frames (every workgroup's slice has a body, a tail and batches
    with none, one or several out-of-range frames), 1 % of them scaled by 2^28 or 2^-36 -- half of those noiseless
    axis-aligned BPSK, whose every phase step is an exact +-pi tie, so that the same frame is out of range AND
    tie-flagged.  Every row equals, through the exact scaling laws, the row of its in-range original computed in a
    small launch of its own: nothing is lost, doubled or left marked whichever wave and chunk a frame lands in."""
    torch = _torch()
    from amcpy_amd import synth
    from amcpy_amd.features import features18
    rng = np.random.default_rng(77 + N)
    base = np.concatenate([synth.host_block(m, 6.0, 4, N, seed=900 + i) for i, m in enumerate(synth.MODS6)])     # 24 frames
    ties = np.tile(np.array([1.0, 1.0, -1.0, 1.0, -1.0, -1.0, 1.0, -1.0], dtype=np.complex64), (4, N // 8))        # +-pi steps
    ties[1] *= 1j
    ties[2, ::3] *= -1
    ties[3] *= (1 + 0j)
    base = np.concatenate([base, ties]).astype(np.complex64)                                                    # 28 originals
    F = 40000 if N < 8192 else 12000          # N = 8192: 512 quads take batches of four; 786 MB of frames
    pick = rng.integers(0, 24, size=F)
    kinds = np.zeros(F, dtype=np.int64)
    out_of_range = rng.choice(F, size=F // 100, replace=False)
    kinds[out_of_range] = rng.integers(1, 3, size=out_of_range.size)
    pick[out_of_range[::2]] = 24 + rng.integers(0, 4, size=out_of_range[::2].size)      # half of them also tie-flagged
    scale = np.array([1.0, 2.0 ** 28, 2.0 ** -36])[kinds]
    batch = torch.from_numpy(base)[torch.from_numpy(pick)].cuda() * torch.from_numpy(scale.astype(np.float32)).cuda()[:, None]
    got = features18(batch).cpu().numpy()
    ref = features18(torch.from_numpy(base).cuda()).cpu().numpy().astype(np.float64)
    assert not np.isneginf(got[:, 4]).any(), "an in-band range mark reached the caller"
    assert ((got[:, 4] >= 0) | np.isnan(got[:, 4])).all(), "a tie flag (negative f5) reached the caller"
    order = np.array([2, 0, 0, 0, 0, 1, 0.5, 0, 0, 2, 2, 4, 4, 4, 6, 6, 6, 6])
    with np.errstate(all="ignore"):
        want = (ref[pick] * scale[:, None] ** order[None, :]).astype(np.float32)
    same = np.isfinite(want) & (np.abs(want) > 1.2e-38)
    assert np.array_equal(np.isinf(got), np.isinf(want))
    assert np.array_equal(np.isnan(got), np.isnan(want))
    bad = np.argwhere(same & ~np.isclose(got, want, rtol=2.5e-7, atol=0))
    assert bad.size == 0, (N, bad[:5], [(got[i, j], want[i, j], kinds[i], pick[i]) for i, j in bad[:5]])
    assert np.array_equal(got[kinds == 0], ref[pick][kinds == 0].astype(np.float32), equal_nan=True)


@pytest.mark.parametrize("N", [2048, 4096, 8192, 10000, 16384, 32768])
def test_ends_of_float32_through_the_range_pass(N):
    """range_extreme_n{N}.npz (captured from the reference): the range fixture's frames at 1e-30, 1e-20,
    1e20 and 1e30 -- |x|^2 itself leaves float32.  The reference, evaluating in complex128, still returns
    finite scale-free features, a finite mean magnitude, and inf / 0 / float32 denormals for the rest.
    The wave kernel flags such frames (also the ones whose power underflows to that of an all-zero
    frame: their angles give them away) and its range pass, working on the frame times an exact power
    of two, reproduces all of it."""
    g = load_npz(f"range_extreme_n{N}.npz")
    x, names, gold32 = g["iq"], [str(n) for n in g["names"]], g["golden64"]
    S = orc.conditioning_scales(x.astype(np.complex128))
    variants = _variants_for(N)
    for variant in variants:                 # the block kernels take every frame times a power of two as well
        _check_ends_of_float32(_run(x, variant), gold32, S, names, variant)
    # non-power-of-two frame sizes (block kernel, Bluestein): the first 1000 samples of the same frames vs the oracle
    x1000 = np.ascontiguousarray(x[:, :1000])
    with np.errstate(all="ignore"):
        gold1000 = orc.features18_batch(x1000.astype(np.complex128)).astype(np.float32)
    _check_ends_of_float32(_run(x1000, "auto"), gold1000, orc.conditioning_scales(x1000.astype(np.complex128)), names, "N=1000")
    # an all-zero frame is still a zero frame (not flagged, not scaled): NaN pattern of the reference
    for variant in variants:
        z = _run(np.zeros((1, N), np.complex64), variant)[0]
        assert np.isnan(z[[3, 7, 8]]).all() and np.all(z[[0, 1, 2, 4, 5, 6] + list(range(9, 18))] == 0)


def _check_ends_of_float32(got, gold32, S, names, what):
    inf_or_zero = ~np.isfinite(gold32) | (gold32 == 0)
    tiny = np.abs(gold32) < 1.2e-38                                   # float32 denormals (and zeros): fewer bits
    bad = np.argwhere(~np.isfinite(gold32) & (got != gold32))
    assert bad.size == 0, (what, [(names[i], j + 1, got[i, j], gold32[i, j]) for i, j in bad[:6]])
    assert np.all(np.abs(got[tiny].astype(np.float64) - gold32[tiny].astype(np.float64)) <= 1e-5 * np.abs(gold32[tiny]) + 1.5e-45)
    rest = ~inf_or_zero & ~tiny
    with np.errstate(all="ignore"):
        diff = np.abs(got.astype(np.float64) - gold32.astype(np.float64))
        scaled = diff / np.maximum(np.abs(gold32.astype(np.float64)), S)
    scaled[~rest] = 0.0
    i, j = np.unravel_index(scaled.argmax(), scaled.shape)
    print(f"\n[ends of float32, {what}] worst scaled rel per feature:", " ".join(f"{v:.1e}" for v in scaled.max(axis=0)))
    assert scaled.max() <= TOL, (what, names[i], j + 1, got[i, j], gold32[i, j])


# ----------------------------------------------------------------------------
# round 3
# ----------------------------------------------------------------------------
@pytest.mark.parametrize("N", [1024, 2048, 4096])
def test_no_worse_than_the_references_own_complex64_path(N):
    """BASELINE's "within 1e-5 relative fp32" cannot hold in plain relative terms for the cumulants that
    cancel (ids 10, 12-18): the REFERENCE ITSELF, run on the same complex64 samples, misses its own
    complex128 result by up to 8.5e-3 there.  The fixtures hold both reference
    outputs for every frame -- golden64 (complex128 evaluation, the baseline of record) and golden32
    (features.py on the complex64 array as it is) -- so the claim "no worse than the reference's complex64
    path" is checked frame by frame: the kernel's distance from golden64 is within the scaled 1e-5, or
    within twice the reference's own complex64-to-complex128 gap on that frame."""
    g = load_npz(f"frames_n{N}.npz")
    x, g64, g32 = g["iq"], g["golden64"].astype(np.float64), g["golden32"].astype(np.float64)
    S = orc.conditioning_scales(x.astype(np.complex128))
    ids = [9, 11, 12, 13, 14, 15, 16, 17]                       # feature ids 10, 12 .. 18, zero-based
    for variant in VARIANTS_POW2:
        got = _run(x, variant).astype(np.float64)
        err = np.abs(got - g64)[:, ids]
        ref_gap = np.abs(g32 - g64)[:, ids]
        allowed = np.maximum(TOL * np.maximum(np.abs(g64[:, ids]), S[:, ids]), 2.0 * ref_gap)
        with np.errstate(divide="ignore", invalid="ignore"):
            plain_hip = (err / np.abs(g64[:, ids])).max(axis=0)
            plain_ref = (ref_gap / np.abs(g64[:, ids])).max(axis=0)
        print(f"\n[N={N} {variant}] ids 10,12..18 worst plain rel, kernel   :", " ".join(f"{v:.1e}" for v in plain_hip))
        print(f"[N={N} {variant}] ids 10,12..18 worst plain rel, ref c64  :", " ".join(f"{v:.1e}" for v in plain_ref))
        bad = np.argwhere(err > allowed)
        assert bad.size == 0, (variant, [(int(i), ids[j] + 1, got[i, ids[j]], g64[i, ids[j]], g32[i, ids[j]]) for i, j in bad[:5]])


def _check_two_rank_line(stdout):
    import json
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, stdout
    assert len(lines[0]) <= 4096, len(lines[0])                 # the driver keeps only the tail of a long line
    rec = json.loads(lines[0])
    per_rank = 6 * 26 * 64
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 2 and rec["scaling"] == "weak"
    assert rec["rccl_ranks"] == 2
    assert rec["config"]["frames_per_gpu_per_step"] == per_rank and rec["cpu_baseline"] is None and rec["h2d"] is None
    total = rec["value"] * rec["ms_per_step"] * 1e-3 * rec["steps"]
    assert abs(total - 2 * per_rank * 3) < 1e-3 * total, (total, rec["value"], rec["ms_per_step"])
    assert rec["roofline"]["launch_ms_min"] <= rec["roofline"]["launch_ms_median"] <= rec["roofline"]["launch_ms_max"]
    assert rec["gather"]["rows_on_rank0"] == 2 * per_rank and rec["gather"]["bytes_per_rank"] == per_rank * 72, rec["gather"]
    # round 5: the line says which rank / device was slow, and carries the one-process fan-out over the same devices
    pr = rec["per_rank"]
    assert [r["rank"] for r in pr] == [0, 1] and all(r["dev"] == 0 and r["frames"] == per_rank for r in pr), pr
    assert all(r["bus"] and 0 < r["ms"][1] <= r["ms"][0] <= r["ms"][2] and r["wall_s"] > 0 for r in pr), pr
    assert all(r["fma_G"] and 100 < r["fma_G"] < 3000 for r in pr), pr
    assert 0 < rec["scaling_efficiency"] <= 1.05 and 0 < rec["rank_balance"] <= 1.0, (rec["scaling_efficiency"], rec["rank_balance"])
    fo = rec["h2d_fanout"]
    assert "error" not in fo and fo["devices"] == [0, 0] and fo["GBps"] > 0 and len(fo["per_device_seconds"]) == 2, fo
    assert sum(fo["per_device_frames"]) == 26 * 64 and fo["bus"] == [pr[0]["bus"]] * 2 and len(fo["numa"]) == len(fo["cpus"]) == 2, fo
    # the same line at the driver's largest N: eight entries wherever this one has two -- still inside the 4 KB the driver keeps
    big = json.loads(lines[0])
    big["per_rank"] = [dict(pr[0], rank=r, dev=r) for r in range(8)]
    for k in ("devices", "staging_threads", "per_device_seconds", "per_device_frames", "bus", "numa", "cpus"):
        big["h2d_fanout"][k] = (fo[k] * 4)[:8]
    big["n_gpus"] = big["rccl_ranks"] = 8
    assert len(json.dumps(big)) <= 4096, len(json.dumps(big))
    return rec


def _launcher_free_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                        "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_RUN_ID", "AMCX_BENCH_SELF_LAUNCHED")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONDONTWRITEBYTECODE="1")
    return env


def test_bench_two_ranks_on_one_gpu(tmp_path):
    """bench.py's N > 1 branch before the driver runs it on a real node, rehearsed on the box's one GPU (gloo for the
    barrier / MAX since RCCL refuses two ranks on one device; --share-gpu maps both ranks to it), started BOTH ways:
    `python bench.py --gpus 2` from a launcher-free environment -- the script starts its own two ranks, as the
    reference's run_extraction forks its own workers (feature_extraction.py:89-97) -- and under torch.distributed.run
    as the driver's multi-GPU command line does.  Either way exactly one JSON line of at most 4 KB reaches stdout, it
    says n_gpus 2 and rccl_ranks 2, and value x wall equals the frames both ranks processed."""
    import socket
    import subprocess
    import sys
    repo = Path(__file__).resolve().parents[1]
    env = _launcher_free_env()
    flags = ["--gpus", "2", "--steps", "3", "--warmup", "2", "--frames", "64", "--dist-backend", "gloo", "--share-gpu"]
    # 1. by itself
    r = subprocess.run([sys.executable, str(repo / "bench.py"), *flags], env=env, cwd=str(tmp_path),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert _check_two_rank_line(r.stdout)["launcher"] == "self/gloo"
    # 2. under the external launcher
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(repo / "bench.py"), *flags]
    r = subprocess.run(cmd, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert _check_two_rank_line(r.stdout)["launcher"] == "external/gloo"
    # 2b. RCCL itself, as far as one GPU allows: ONE rank under the launcher with the nccl backend -- communicator with
    #     device_id, the all-reduce behind rccl_ranks, the barriers around the timed region, the MAX of the wall times
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(repo / "bench.py"),
           "--gpus", "1", "--steps", "3", "--warmup", "2", "--frames", "64", "--no-cpu-baseline", "--no-h2d"]
    r = subprocess.run(cmd, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                            # RCCL's banner must not reach stdout
    one = __import__("json").loads(lines[0])
    assert one["n_gpus"] == 1 and one["rccl_ranks"] == 1 and one["launcher"] == "external/nccl" and one["gather"] is None
    # 3. a launcher that started a different number of ranks than --gpus says is refused, not silently run
    bad = subprocess.run([sys.executable, str(repo / "bench.py"), "--gpus", "2", "--no-cpu-baseline"],
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                         cwd=str(tmp_path), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in (bad.stdout + bad.stderr)
    # 4. a rank that fails takes the job down with its exit code and leaves no rank behind (two ranks, no
    #    --share-gpu: rank 1 has no GPU of its own on this one-GPU box and says so)
    if _torch().cuda.device_count() == 1:
        bad = subprocess.run([sys.executable, str(repo / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                              "--frames", "8", "--dist-backend", "gloo", "--launch-timeout", "120"],
                             env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
        assert bad.returncode != 0 and "no GPU of its own" in bad.stderr, bad.stderr[-2000:]
        assert not [ln for ln in bad.stdout.splitlines() if ln.strip()], bad.stdout


def test_fma_ceiling_probe_and_host_placement_on_this_box():
    """ABI 4 on the device.  amcx_probe_fma_rate: the board's instruction-issue ceiling under its power cap -- between
    the ~850 G wave-instructions/s rounds 1-4 measured under load and the 256 CUs x 4 SIMDs x 2.4 GHz / 2.0 cycles
    ~ 1 230 G an uncapped chip could issue -- at a clock inside the part's range.  Placement: a context knows its
    device's PCI bus id; where the kernel's tree names a NUMA node the staging threads are bound to CPUs this process
    may use (never more than the node has), where it says -1 nothing is bound; an explicit binding is honoured and
    results do not depend on any of it."""
    torch = _torch()
    from amcpy_amd import _lib
    from amcpy_amd import feature_extraction as fe
    got = _lib.probe_fma_rate(0.4, torch.cuda.current_stream().cuda_stream)
    assert 400e9 < got["wave_instr_per_s"] < 1400e9 and 1.0 < got["clock_GHz"] < 2.6, got
    bus = _lib.device_pci_bus_id(0)
    assert len(bus.split(":")) == 3 and bus == bus.lower()
    node, cpus = _lib.numa_place(bus)
    rng = np.random.default_rng(11)
    box = np.asfortranarray((rng.standard_normal((3, 40, 2048)) + 1j * rng.standard_normal((3, 40, 2048))))   # 3.9 MB: threaded path
    rows = fe.FrameRows(box, 3, 40)
    eng = fe.HipEngine(2048, 0, threads=4)
    ref = eng(rows)
    pl = eng._context().placement()
    assert pl["pci_bus_id"] == bus and pl["device"] == 0 and pl["numa_node"] == node and pl["n_cpus"] == len(cpus)
    assert pl["n_cpus_allowed"] <= pl["n_cpus"]
    before = os.sched_getaffinity(0)
    two = sorted(before)[:2]
    eng._context().bind_cpus(two)
    assert eng._context().placement()["n_cpus"] == len(two)
    again = eng(rows)
    assert os.sched_getaffinity(0) == before                         # the caller's own mask came back
    eng._context().bind_cpus([])
    unbound = eng(rows)
    assert eng._context().placement()["numa_node"] == -1
    assert np.array_equal(ref.view(np.int32), again.view(np.int32)) and np.array_equal(ref.view(np.int32), unbound.view(np.int32))
    eng.close()


def test_strided_engine_on_random_layouts():
    """amcx_ctx_features18_strided_host on forty random containers: every axis order (the snr, the frame or the
    sample axis contiguous), padded in every dimension (strides larger than the extents), complex64 / complex128 /
    split real+imaginary / real-only, ranges that start inside an snr row, slots small enough for many chunks --
    each result equal, bit for bit, to the one-launch result on a packed complex64 copy of the same frames."""
    torch = _torch()
    import itertools
    from amcpy_amd.feature_extraction import FrameRows, HipEngine, SplitComplex
    from amcpy_amd.features import features18
    rng = np.random.default_rng(2026)
    orders = list(itertools.permutations(range(3)))
    for case in range(40):
        S, K = int(rng.integers(1, 6)), int(rng.integers(1, 40))
        N = int(rng.choice([64, 100, 128, 256, 300]))
        pad = [int(rng.integers(0, 3)) for _ in range(3)]
        shape = (S + pad[0], K + pad[1], N + pad[2])
        order = orders[int(rng.integers(0, len(orders)))]          # memory order: order[0] is the slowest axis
        full = (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) * 10.0 ** rng.integers(-2, 3)
        mem = np.ascontiguousarray(full.transpose(order))          # laid out with `order`'s axes from slow to fast
        arr = mem.transpose(np.argsort(order))                     # ... and viewed as (snr, frame, sample) again
        assert arr.shape == shape and np.array_equal(arr, full)
        kind = int(rng.integers(0, 4))
        if kind == 0:
            src = arr.astype(np.complex64, order="K")
            packed = np.asarray(src)
        elif kind == 1:
            src, packed = arr, arr.astype(np.complex64)
        elif kind == 2:
            src = SplitComplex(np.ascontiguousarray(arr.real.transpose(order)).transpose(np.argsort(order)),
                               np.ascontiguousarray(arr.imag.transpose(order)).transpose(np.argsort(order)))
            packed = arr.astype(np.complex64)
        else:
            src = np.ascontiguousarray(arr.real.transpose(order)).transpose(np.argsort(order))     # a real signal
            packed = arr.real.astype(np.complex64)
        want_all = features18(torch.from_numpy(np.ascontiguousarray(packed[:S, :K, :N]).reshape(S * K, N)).cuda()).cpu().numpy()
        lo = int(rng.integers(0, S * K))
        hi = int(rng.integers(lo, S * K + 1))
        eng = HipEngine(N, chunk_bytes=int(rng.choice([4096, 20000, 1 << 20])), threads=int(rng.integers(1, 5)),
                        round_on_device=bool(rng.integers(0, 2)))
        got = eng(FrameRows(src, S, K, lo, hi))
        assert got.shape == (hi - lo, 18)
        assert np.array_equal(got, want_all[lo:hi], equal_nan=True), (case, shape, order, kind, lo, hi, N)
        eng.close()


def test_reference_side_ctypes_stub_from_integration_md(tmp_path):
    """The binding INTEGRATION.md section 2 tells a maintainer of the reference to add -- ctypes and numpy only (the
    file form also takes the variable's offsets from amcpy_amd.matfile) -- executed as written (the fenced python
    blocks of that section, in order) on what loadmat returns: a Fortran-ordered complex128 container with more frames
    and longer rows than the configuration uses; and on the .mat file itself."""
    import re
    import subprocess
    import sys
    repo = Path(__file__).resolve().parents[1]
    text = (repo / "INTEGRATION.md").read_text()
    sec = text[text.index("## 2. The stub"):text.index("## 3. The two edits")]
    blocks = re.findall(r"```python\n(.*?)```", sec, flags=re.S)
    assert len(blocks) >= 3
    lib = repo / "amcpy_amd" / "lib" / "libamcx.so"
    assert "amcx_ctx_features18_strided_file" in blocks[2]
    code = f"import os, sys\nsys.path.insert(0, {str(repo)!r})\n" + \
        "\n".join(blocks[:3]).replace('C.CDLL("libamcx.so")', f'C.CDLL({str(lib)!r})')
    code += textwrap.dedent(f"""
        rng = np.random.default_rng(4)
        full = np.asfortranarray(rng.standard_normal((3, 40, 300)) + 1j * rng.standard_normal((3, 40, 300)))
        got = container_features(full, 2, 33, 256)
        want = features18(full[:2, :33, :256].reshape(66, 256), 256).reshape(2, 33, 18)
        assert got.shape == (2, 33, 18) and got.dtype == np.float32
        assert np.array_equal(got, want, equal_nan=True), np.abs(got - want).max()
        import scipy.io
        scipy.io.savemat({str(tmp_path / "c.mat")!r}, {{"other": np.arange(5.0), "signal": full}})
        from_file = mat_variable_features({str(tmp_path / "c.mat")!r}, "signal", 3, 40, 256)
        assert np.array_equal(from_file, container_features(full, 3, 40, 256), equal_nan=True)
        print("STUB_OK", float(got[1, 2, 5]))
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "STUB_OK" in r.stdout, r.stdout + r.stderr


def test_engine_reads_the_container_from_its_file(tmp_path):
    """amcx_ctx_features18_strided_file: the variable of an uncompressed level-5 .mat staged from the FILE by the
    engine's threads gives bit-identical features to the array loadmat returns for it (whole container, a
    sub-rectangle with fewer snr rows / frames / samples than stored, a frame range of a shard); run_extraction
    writes the same files whichever way it reads; a file cut short is an OSError that leaves the engine usable."""
    import scipy.io
    from amcpy_amd.config import Config, Paths, SignalConfig
    from amcpy_amd.feature_extraction import FileComplex, FrameRows, HipEngine, run_extraction
    from amcpy_amd.matfile import load_variable
    rng = np.random.default_rng(31)
    S, K, L, N = 5, 400, 1100, 1024        # planes of 16 KB: long enough runs for the file path
    cfg = Config(paths=Paths(root=tmp_path), signals=SignalConfig(snr_values={i: str(i) for i in range(S)}, num_frames=K,
                                                                  frame_size=N, modulations_with_noise=("BPSK", "QPSK")))
    cfg.paths.ensure_dirs()
    mats = {cfg.signals.mat_info[m]: np.asfortranarray(rng.standard_normal((S, K, L)) + 1j * rng.standard_normal((S, K, L)))
            for m in cfg.signals.modulations_with_noise}
    path = cfg.paths.mat_data / cfg.paths.mat_filename
    scipy.io.savemat(str(path), mats)
    key = cfg.signals.mat_info["QPSK"]
    loaded = scipy.io.loadmat(str(path))[key]
    fx = load_variable(path, key, direct=True)
    assert isinstance(fx, FileComplex)
    eng = HipEngine(N, threads=4, chunk_bytes=1 << 20)              # several chunks
    # whole container: planes read from the file; fewer snr rows / a shard's frame range / one snr row leave runs of
    # a few elements, which go through the mapping instead
    for (s, k, lo, hi, from_file) in [(S, K, 0, None, 1), (S, 300, 0, None, 1), (3, 50, 0, None, 0), (S, K, 37, 1290, 0), (1, K, 0, None, 0)]:
        a = eng(FrameRows(loaded, s, k, lo, hi))
        assert eng.stats["from_file"] == 0
        b = eng(FrameRows(fx, s, k, lo, hi))
        assert eng.stats["from_file"] == from_file, (s, k, lo, hi, eng.stats)
        assert np.array_equal(a, b, equal_nan=True), (s, k, lo, hi)
    # a rank's share when the container is cut along the frame axis: frames [k_lo, k_hi) of every snr row, from the file
    from amcpy_amd.feature_extraction import FrameColumns
    whole = eng(FrameRows(loaded, S, K)).reshape(S, K, 18)
    share = eng(FrameColumns(fx, S, K, 120, 330))
    assert eng.stats["from_file"] == 1 and np.array_equal(share.reshape(S, 210, 18), whole[:, 120:330], equal_nan=True)
    want = orc.features18_batch(loaded[:S, :K, :N].reshape(-1, N).astype(np.complex64))
    _assert_parity(eng(FrameRows(fx, S, K)), want, loaded[:S, :K, :N].reshape(-1, N), "file engine")
    # run_extraction: direct (default) against the python-side readers
    run_extraction(cfg, verbose=False)
    first = {m: scipy.io.loadmat(str(cfg.paths.calculated_features / f"{m}_features.mat"))[cfg.signals.mat_info[m]]
             for m in cfg.signals.modulations_with_noise}
    os.environ["AMCX_DIRECT_FILE"] = "0"
    try:
        run_extraction(cfg, verbose=False)
    finally:
        del os.environ["AMCX_DIRECT_FILE"]
    for m, f in first.items():
        again = scipy.io.loadmat(str(cfg.paths.calculated_features / f"{m}_features.mat"))[cfg.signals.mat_info[m]]
        assert f.shape == (S, K, 18) and np.array_equal(f, again, equal_nan=True), m
    # a file that ends inside the variable
    cut = tmp_path / "cut.mat"
    cut.write_bytes(path.read_bytes()[:fx.imag_offset + 4096])
    broken = FileComplex(cut, fx.store, fx.shape, fx.real_offset, fx.imag_offset)
    with pytest.raises(OSError):
        eng(FrameRows(broken, S, K))
    assert np.array_equal(eng(FrameRows(fx, S, K)), eng(FrameRows(loaded, S, K)), equal_nan=True)
    eng.close()


def test_small_host_calls_replay_a_graph():
    """Small row-major host calls (the per-frame loop of calculate_features, features.py:214-232) run as one cached
    graph per shape -- copy in, kernels, copy out captured once, then hipGraphLaunch + one synchronisation.  Same
    bits as the general path (AMCX_NO_GRAPH=1) for complex64 and complex128, several frame sizes and frame counts
    in turn (more shapes than the cache holds), a padded output, and after a large call has made the context
    reallocate its buffers; and it is the faster of the two."""
    import time
    from amcpy_amd import _lib
    from amcpy_amd.features import features18_host
    rng = np.random.default_rng(61)
    frames = {(F, N, dt): (rng.standard_normal((F, N)) + 1j * rng.standard_normal((F, N))).astype(dt)
              for F, N in [(1, 2048), (3, 2048), (1, 1024), (2, 4096), (1, 1000), (5, 256)] for dt in (np.complex64, np.complex128)}

    def run_all():
        return {k: features18_host(x) for k, x in frames.items()}

    os.environ["AMCX_NO_GRAPH"] = "1"
    try:
        want = run_all()
    finally:
        del os.environ["AMCX_NO_GRAPH"]
    for rep in range(3):                                        # twelve shapes through a cache of four, three times over
        got = run_all()
        for k in frames:
            assert np.array_equal(got[k], want[k], equal_nan=True), (rep, k)
    big = (rng.standard_normal((3000, 2048)) + 1j * rng.standard_normal((3000, 2048))).astype(np.complex64)
    ref_big = features18_host(big)                               # 49 MB: buffers grow, the graphs' addresses are stale
    assert np.array_equal(features18_host(frames[(1, 2048, np.complex64)]), want[(1, 2048, np.complex64)], equal_nan=True)
    assert np.array_equal(ref_big[:3], features18_host(big[:3]), equal_nan=True)
    # the per-frame loop, with and without the graph
    x = frames[(1, 2048, np.complex64)]
    ctx = _lib.HostContext(0)
    out = np.empty((1, 18), dtype=np.float32)
    def loop(n=300):
        t0 = time.perf_counter()
        for _ in range(n):
            ctx.run_strided(x.ctypes.data, None, _lib.SRC_C64, 1, 1, 2048, (0, 2048, 1), out)
        return (time.perf_counter() - t0) / n
    loop(20)
    with_graph = min(loop(), loop())
    os.environ["AMCX_NO_GRAPH"] = "1"
    try:
        loop(20)
        without = min(loop(), loop())
    finally:
        del os.environ["AMCX_NO_GRAPH"]
    print(f"\nper 2048-sample frame through the context: {with_graph * 1e6:.1f} us as a graph, {without * 1e6:.1f} us as separate calls")
    assert np.array_equal(out, want[(1, 2048, np.complex64)], equal_nan=True)
    assert with_graph < without
    ctx.close()


# ----------------------------------------------------------------------------
# round 5, third session: every frame size up to 32768 (ABI 6, amcx_stream_kernel.h)
# ----------------------------------------------------------------------------
@pytest.mark.parametrize("N", [8193, 9999, 16385, 20000, 30011])
def test_any_frame_size_above_8192_against_oracle(N):
    """frame_size is a free integer in the reference (config.py:96; np.fft.fft takes any N, features.py:68).  Sizes above
    8192 that are not powers of two -- 8193 (just past the LDS-staged block kernel), an odd composite, 16385 (the first size
    whose samples are staged in two chunks), 20000, a prime -- through the device entry (AUTO = the stream kernel) and through
    the host engine on a complex128 container with a row stride above N (run_extraction's [0:frame_size] slice,
    feature_extraction.py:68), against the oracle in complex128; more frames than one workgroup round of a small grid
    would need is not required: the kernel strides a grid of one workgroup per CU over them."""
    from amcpy_amd import _lib, synth
    from amcpy_amd.feature_extraction import HipEngine
    assert _lib.kernel_name(N, _lib.VARIANT_AUTO) == "amcx_features18_stream_kernel"
    L = N + 37
    frames = np.concatenate([synth.host_block(mod, snr, 1, L, seed=7000 + N % 1000 + i)
                             for i, (mod, snr) in enumerate([("BPSK", 4.0), ("16QAM", 12.0), ("WGN", 0.0), ("8PSK", -6.0)])])
    x128 = frames.astype(np.complex128) * (1.0 + 1e-9)                  # genuine doubles: the engine rounds them on the way
    want = orc.features18_batch(np.ascontiguousarray(x128[:, :N]).astype(np.complex64).astype(np.complex128))
    got = _run(frames.astype(np.complex64), "auto", frame_size=N)
    _assert_parity(got, want, frames[:, :N].astype(np.complex64), f"any-size N={N} device")
    host = HipEngine(N)(x128)                                            # its context owns a workspace: the FFT form here too
    assert np.array_equal(host, got), "the host engine and the device entry disagree"


def test_stream_kernel_strides_its_grid_and_keeps_frames_apart():
    """More frames than CUs (the grid is one workgroup per CU, striding), a NaN frame, an all-zero frame and an
    out-of-fp32-range frame in between: every row equals the row the same frame gets alone."""
    torch = _torch()
    from amcpy_amd import synth
    N, F = 8200, 300
    x = np.concatenate([synth.host_block("QPSK", 6.0, F // 2, N, seed=91), synth.host_block("64QAM", 15.0, F - F // 2, N, seed=92)])
    x = x.astype(np.complex64)
    x[17] = 0
    x[130, 5] = np.nan
    x[277] *= np.float32(1e15)
    got = _run(x, "auto")
    pick = [0, 17, 130, 131, 256, 277, 299]
    alone = np.concatenate([_run(x[i:i + 1], "block") for i in pick])
    assert np.array_equal(got[pick], alone, equal_nan=True)
    assert np.isnan(got[130]).all() and not np.isnan(got[131]).any()
    assert np.isnan(got[17][[3, 7, 8]]).all() and np.all(got[17][[0, 1, 2, 4, 5, 6] + list(range(9, 18))] == 0)
    want = orc.features18_batch(x[[0, 256, 299]].astype(np.complex128))
    _assert_parity(got[[0, 256, 299]], want, x[[0, 256, 299]], "stream kernel, batch of 300")
    with np.errstate(all="ignore"):
        w277 = orc.features18_batch(x[277:278].astype(np.complex128)).astype(np.float32)
    assert np.array_equal(np.isinf(got[277]), np.isinf(w277[0])) and np.isinf(got[277]).any()


def _run_ws(frames, frame_size, ws_bytes, variant="auto"):
    """features through amcx_features18_c64_ws with a workspace of exactly ws_bytes (0: none) -> numpy (F, 18)."""
    torch = _torch()
    from amcpy_amd import _lib
    lib = _lib.load()
    x = torch.from_numpy(np.ascontiguousarray(frames)).cuda()
    out = torch.full((x.shape[0], 18), float("nan"), dtype=torch.float32, device="cuda")
    ws = torch.empty(max(ws_bytes, 8) // 8, dtype=torch.complex64, device="cuda").fill_(complex(float("nan"), float("nan")))
    _lib.check(lib.amcx_features18_c64_ws(x.data_ptr(), x.shape[0], frame_size, x.shape[1], out.data_ptr(), 18,
                                          torch.cuda.current_stream().cuda_stream, _lib.VARIANTS[variant],
                                          ws.data_ptr() if ws_bytes else None, ws_bytes))
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("N,variant", [(8193, "auto"), (12289, "auto"), (16384, "block"), (16385, "auto"), (32767, "auto"), (32768, "block")])
def test_the_two_forms_of_the_any_size_path_agree(N, variant):
    """Above 8192 samples the spectral term is an FFT through a workspace (Bluestein's chirp-z, M = 32768 / 65536; the plain
    transform at the two powers of two) or, without one, the DFT by its definition.  Both against the oracle, and against
    each other: features 2-18 do not involve the spectral term and are bit-identical, f1 agrees to 2e-6.  A workspace with
    room for fewer workgroups than frames in flight, a workspace one byte short of one workgroup's share (-> the
    workspace-free form) and one polluted with NaNs give the same rows."""
    from amcpy_amd import _lib, synth
    lib = _lib.load()
    F = 7
    x = np.concatenate([synth.host_block(mod, snr, 1, N, seed=8100 + i)
                        for i, (mod, snr) in enumerate([("BPSK", 0.0), ("QPSK", 8.0), ("8PSK", 14.0), ("16QAM", 20.0),
                                                        ("64QAM", 30.0), ("WGN", 0.0), ("BPSK", -12.0)])]).astype(np.complex64)
    x[3] *= np.float32(3e-9)                                      # a frame that is scaled by a power of two on the way
    want = orc.features18_batch(x.astype(np.complex128))
    full = lib.amcx_features18_workspace_bytes(N, F, _lib.VARIANTS[variant])
    M = 1 << int(np.ceil(np.log2(2 * N - 1))) if N & (N - 1) else N
    assert full == (F + (1 if N & (N - 1) else 0)) * M * 8
    fft = _run_ws(x, N, full, variant)
    direct = _run_ws(x, N, 0, variant)
    _assert_parity(fft, want, x, f"any-size N={N} FFT form")
    _assert_parity(direct, want, x, f"any-size N={N} by the definition")
    assert np.array_equal(fft[:, 1:], direct[:, 1:], equal_nan=True)
    assert np.allclose(fft[:, 0], direct[:, 0], rtol=2e-6, atol=0)
    two = (2 + (1 if N & (N - 1) else 0)) * M * 8                 # room for two frames in flight
    assert np.array_equal(_run_ws(x, N, two, variant), fft)
    short = (1 + (1 if N & (N - 1) else 0)) * M * 8 - 1           # not even one: the form that needs none
    assert np.array_equal(_run_ws(x, N, short, variant), direct)
    torch = _torch()                                              # a workspace that is not 8-byte aligned is not used either
    from amcpy_amd import _lib as L
    xs = torch.from_numpy(x).cuda()
    o = torch.zeros((F, 18), dtype=torch.float32, device="cuda")
    w = torch.zeros(full + 16, dtype=torch.uint8, device="cuda")
    L.check(lib.amcx_features18_c64_ws(xs.data_ptr(), F, N, N, o.data_ptr(), 18, None, L.VARIANTS[variant], w.data_ptr() + 4, full))
    torch.cuda.synchronize()
    assert np.array_equal(o.cpu().numpy(), direct)
    assert np.array_equal(_run(x, variant), fft), "amcx_features18_c64_ex takes its workspace from the stream-ordered allocator"


def test_any_size_path_inside_a_graph_capture():
    """The any-size path while its stream is being captured into a graph.  The C entry amcx_features18_c64_ex allocates
    nothing then: the captured node is the workspace-free form (the DFT by its definition), and replaying the graph gives
    that form's rows.  amcpy_amd.features.features18 takes the workspace from TORCH's allocator (round 6: the allocator
    that owns the device memory, and one that is capture-aware -- the graph keeps the bytes), so the node it captures is
    the FFT form: the rows of the eager call, bit for bit."""
    torch = _torch()
    from amcpy_amd import _lib, synth
    from amcpy_amd.features import features18
    N = 9000
    x = torch.from_numpy(synth.host_block("QPSK", 10.0, 5, N, seed=77).astype(np.complex64)).cuda()
    out = torch.zeros((5, 18), dtype=torch.float32, device="cuda")
    features18(x, out=out)                                        # warm (attributes, allocator)
    torch.cuda.synchronize()
    eager = out.clone()
    lib = _lib.load()
    # 1. the C entry under capture: no allocation, the workspace-free form
    out.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        _lib.check(lib.amcx_features18_c64_ex(x.data_ptr(), 5, N, N, out.data_ptr(), 18, torch.cuda.current_stream().cuda_stream,
                                              _lib.VARIANTS["auto"]))
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    direct = _run_ws(x.cpu().numpy(), N, 0)
    assert np.array_equal(out.cpu().numpy(), direct)
    assert np.array_equal(out.cpu().numpy()[:, 1:], eager.cpu().numpy()[:, 1:])
    assert np.allclose(out.cpu().numpy()[:, 0], eager.cpu().numpy()[:, 0], rtol=2e-6, atol=0)
    # 2. the torch entry under capture: the workspace comes from the graph's own pool, the FFT form is captured
    out.zero_()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        features18(x, out=out)
    out.zero_()
    g2.replay()
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), eager.cpu().numpy())
    g2.replay()                                                    # ... and again: the workspace is the graph's for its lifetime
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), eager.cpu().numpy())


def test_quad_launch_cut_into_several(tmp_path):
    """launch_quad cuts an input longer than its workgroups' re-run masks cover (8.4 M frames of 64 KiB: more than a device
    holds) into several launches.  With the cut forced at 52 frames (AMCX_TEST_QUAD_SPLIT, which the library reads once per
    process: the cut run is a process of its own) 211 frames go as five launches: same rows as one launch, an
    out-of-range frame on either side of a cut included."""
    import subprocess
    import sys
    from amcpy_amd import synth
    N, F = 8192, 211
    x = synth.host_block("16QAM", 9.0, F, N, seed=4242).astype(np.complex64)
    x[51] *= np.float32(1e14)
    x[52] *= np.float32(1e-14)
    x[207] *= np.float32(3e12)
    whole = _run(x, "wave")
    np.save(tmp_path / "x.npy", x)
    code = textwrap.dedent(f"""
        import sys, numpy as np, torch
        sys.path.insert(0, {str(REPO)!r})
        from amcpy_amd.features import features18
        x = torch.from_numpy(np.load({str(tmp_path / 'x.npy')!r})).cuda()
        y = features18(x, variant="wave")
        torch.cuda.synchronize()
        np.save({str(tmp_path / 'cut.npy')!r}, y.cpu().numpy())
    """)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, AMCX_TEST_QUAD_SPLIT="52", PYTHONDONTWRITEBYTECODE="1"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    cut = np.load(tmp_path / "cut.npy")
    assert np.array_equal(cut, whole, equal_nan=True)
    assert np.isinf(whole[51]).any() and np.isfinite(whole[50]).all()
