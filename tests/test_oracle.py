"""The oracle against the reference's own pins (CPU only).

1. the reference's known-answer table (features.py:286-305) through both
   oracle evaluators;
2. the golden vectors captured from the imported reference
   (oracle/capture_golden.py): reference-shaped evaluator bit-for-bit (same
   numpy calls), fused evaluator to fp64 round-off;
3. degenerate frames (NaN pattern included).
"""
import numpy as np
import pytest

from oracle import iq_features_oracle as orc
from tests.conftest import load_npz


def _kat_signal(kat):
    return np.array(kat["signal_re"]) + 1j * np.array(kat["signal_im"])


def test_kat_reference_shaped(kat):
    got = orc.calculate_features(range(1, 19), _kat_signal(kat))
    assert np.allclose(got, kat["expected"], rtol=kat["rtol"], atol=0)


def test_kat_fused(kat):
    got = orc.features18_batch(_kat_signal(kat))[0]
    assert np.allclose(got, kat["expected"], rtol=kat["rtol"], atol=0)


def test_kat_spot_values(kat):
    x = _kat_signal(kat)
    iv = orc.instantaneous(x)
    m = orc.mixed_moments(x)
    s = kat["spot"]
    assert len(iv["frequency"]) == s["len_frequency"] == 9
    assert np.isclose(iv["abs"][1], np.sqrt(2), atol=1e-10)
    assert np.isclose(iv["cn_amplitude"][0], -1.0, atol=1e-10)
    assert np.isclose(iv["cn_amplitude"][-1], 1.0, atol=1e-10)
    assert np.isclose(m["m21"], 57.0, atol=1e-10)
    assert np.isclose(m["m42"], 6133.2, atol=1e-6)
    assert np.isclose(np.real(m["m63"]), 782724.0, atol=1e-6)


def test_subset_order_and_unknown_id(kat):
    x = _kat_signal(kat)
    got = orc.calculate_features([14, 2, 2, 7], x)
    exp = [kat["expected"][13], kat["expected"][1], kat["expected"][1], kat["expected"][6]]
    assert np.allclose(got, exp, rtol=1e-5)
    with pytest.raises(KeyError):
        orc.calculate_features([0], x)
    with pytest.raises(KeyError):
        orc.calculate_features([19], x)


def test_reference_shaped_matches_reference_outputs(golden_frames):
    N, g = golden_frames
    x = g["iq"]
    for i in range(0, x.shape[0], 5):           # every 5th frame: keeps CPU time small
        got = np.array(orc.calculate_features(range(1, 19), x[i].astype(np.complex128)))
        assert np.allclose(got, g["golden64_f64"][i], rtol=1e-12, atol=0), (N, i)
        got32 = orc.features18_frame(x[i], dtype=np.complex64)
        # the reference's own complex64 path: same numpy calls -> same bits,
        # modulo summation-order freedom numpy leaves itself; 1e-6 is ample
        assert np.allclose(got32, g["golden32"][i], rtol=1e-6, atol=0), (N, i)


def test_fused_matches_reference_outputs(golden_frames):
    N, g = golden_frames
    got = orc.features18_batch(g["iq"])
    S = orc.conditioning_scales(g["iq"])
    plain, scaled = orc.parity_errors(got, g["golden64_f64"], S)
    assert scaled.max() < 1e-12, scaled.max(axis=0)
    # plain relative: algebraically exact, only conditioning (<=~1e4) shows
    assert plain.max() < 1e-9, plain.max(axis=0)


def test_moments_match_reference(golden_frames):
    N, g = golden_frames
    m = orc.batch_moments(g["iq"])
    keys = ("m20", "m21", "m22", "m40", "m41", "m42", "m43", "m60", "m61", "m62", "m63")
    got = np.stack([m[k] for k in keys], axis=1)
    ref = g["moments"]
    assert np.allclose(got, ref, rtol=1e-11, atol=1e-13)


def test_wrapped_difference_is_diff_unwrap():
    rng = np.random.default_rng(7)
    th = rng.uniform(-np.pi, np.pi, size=(50, 257))
    th[0, :8] = [np.pi, -np.pi, np.pi, 0.0, np.pi, 0, -np.pi, 0]     # exact +-pi steps
    a = orc.wrapped_first_difference(th)
    b = np.diff(np.unwrap(th, axis=-1), axis=-1)
    assert np.allclose(a, b, rtol=0, atol=1e-11)   # unwrap cumsums: O(N eps)


def test_edges_fused_vs_reference(golden_edges):
    N, g = golden_edges
    got = orc.features18_batch(g["iq"])
    ref = g["golden64_f64"]
    names = [str(s) for s in g["names"]]
    for i, name in enumerate(names):
        r, o = ref[i], got[i]
        if name == "one_nan":
            assert np.isnan(r).all() and np.isnan(o).all()
            continue
        # NaN pattern must agree wherever the variance is exactly zero
        if name in ("zeros", "const_pos", "alternating", "impulse"):
            assert (np.isnan(r) == np.isnan(o)).all(), (name, r, o)
        ok = ~np.isnan(r) & ~np.isnan(o)
        # rounding-noise features (kurtosis of a numerically constant series)
        # are not parity targets: SURVEY.md Appendix C
        noisy = np.zeros(18, bool)
        if name in ("tone_k5", "const_neg", "const_neg_negzero", "ramp_phase_pi", "huge", "tiny",
                    "alternating", "real_only", "imag_only"):
            noisy[[7, 8]] = True
        sel = ok & ~noisy
        scale = np.maximum(np.abs(r[sel]), 1e-6 * np.abs(r[sel]).max() + 1e-30)
        assert (np.abs(o[sel] - r[sel]) / scale).max() < 1e-6, (name, o, r)


def test_extract_roundtrip_fixture_is_reference_rows():
    g = load_npz("extract_roundtrip.npz")
    fs = int(g["frame_size"])
    for mod in g["mods"]:
        mod = str(mod)
        x = g[f"in_{mod}"][:, :, :fs].astype(np.complex128)
        out = g[f"out_{mod}"]
        assert out.dtype == np.float32 and out.shape == (2, int(g["n_frames"]), 18)
        got = orc.features18_batch(x.reshape(-1, fs)).astype(np.float32).reshape(out.shape)
        assert np.allclose(got, out, rtol=2e-6, atol=0, equal_nan=True)
        assert str(g[f"label_{mod}"]) == mod


@pytest.mark.parametrize("N", [2048, 4096, 8192, 10000, 16384, 32768])
def test_oracle_at_extreme_scales_matches_reference(N):
    """range_n{N}.npz (captured from the reference): frames at scales 1e-12 ... 1e12, a mixed-scale
    frame and a single 5e7 spike.  Both oracle evaluators on the complex128 cast reproduce the
    reference's float64 values (and therefore its float32-stored inf / 0 pattern)."""
    g = load_npz(f"range_n{N}.npz")
    x, gold = g["iq"].astype(np.complex128), g["golden64_f64"]
    fused = orc.features18_batch(x)
    assert np.allclose(fused, gold, rtol=5e-9, atol=0, equal_nan=True)
    for i in range(0, x.shape[0], 6):
        got = np.array(orc.calculate_features(range(1, 19), x[i]))
        assert np.array_equal(got, gold[i], equal_nan=True), i      # same numpy calls: bit for bit
    with np.errstate(all="ignore"):
        stored = gold.astype(np.float32)
    assert np.array_equal(stored, g["golden64"], equal_nan=True)
    assert np.isinf(stored).any() and (stored == 0).any()           # the fixture does exercise both ends


def test_oracle_on_genuine_doubles_matches_reference_run():
    """extract_roundtrip_f64.npz: what the reference's run_extraction stored (float32) for a container
    of doubles that are not float32-representable equals the oracle on the same doubles."""
    g = load_npz("extract_roundtrip_f64.npz")
    fs = int(g["frame_size"])
    for m in (str(v) for v in g["mods"]):
        x = g[f"in_{m}"]
        assert x.dtype == np.complex128 and not np.array_equal(x, x.astype(np.complex64))
        got = orc.features18_batch(x[:, :, :fs].reshape(-1, fs)).astype(np.float32)
        want = g[f"out_{m}"].reshape(-1, 18)
        assert np.allclose(got, want, rtol=2e-6, atol=0, equal_nan=True), m


@pytest.mark.parametrize("N", [2048, 4096, 8192, 10000, 16384, 32768])
def test_oracle_at_the_ends_of_float32_matches_reference(N):
    g = load_npz(f"range_extreme_n{N}.npz")
    x, gold = g["iq"].astype(np.complex128), g["golden64_f64"]
    assert np.allclose(orc.features18_batch(x), gold, rtol=5e-9, atol=0, equal_nan=True)
    got = np.array(orc.calculate_features(range(1, 19), x[0]))
    assert np.array_equal(got, gold[0], equal_nan=True)


@pytest.mark.parametrize("fixture", ["configs0_reference_run.npz", "configs2_reference_run.npz", "configs4_reference_run.npz"])
def test_oracle_matches_the_references_run_of_baseline_configs0(fixture):
    """configs0_reference_run.npz: BASELINE configs[0] (6 modulations x 2 SNR x 500 frames x 2048 samples, a container of
    MATLAB doubles) through the REFERENCE's own run_extraction (feature_extraction.py:85-99; 16.7 s on this container's
    eight cores).  The inputs are regenerated from their seeds (SHA-256 checked); the oracle reproduces all 6 000 rows
    the reference stored.  configs2_reference_run.npz: the same at BASELINE configs[2]'s frame size (6 x 2 x 50 x 4096)."""
    import hashlib
    from amcpy_amd import synth
    g = load_npz(fixture)
    n_snr, n_frames, fs = int(g["n_snr"]), int(g["n_frames"]), int(g["frame_size"])
    blocks = synth.host_frames(synth.MODS6, n_snr, n_frames, fs)
    for m in synth.MODS6:
        x = np.ascontiguousarray(blocks[m])
        assert hashlib.sha256(x.tobytes()).hexdigest() == str(g[f"sha256_in_{m}"]), f"{m}: the seeded input changed"
        got = orc.features18_batch(x.reshape(-1, fs).astype(np.complex128)).astype(np.float32)
        want = g[f"out_{m}"].reshape(-1, 18)
        assert np.allclose(got, want, rtol=2e-6, atol=0), m
