"""Device-side consumers behind the extraction path
against the numpy / scikit-learn arithmetic the reference calls
(graphics.py:50-62 np.mean/np.std; preprocessing.py:52-62 StandardScaler)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _feats(shape, seed):
    rng = np.random.default_rng(seed)
    base = rng.standard_normal(shape).astype(np.float32)
    scales = np.array([1e3, 1, 1, 0.3, 0.2, 1, 0.02, 3, 10, 1, 1, 2, 2, 2, 10, 10, 10, 10], np.float32)
    offs = np.array([2e3, 0.9, 1.8, 0.5, 0.25, 1, 0.022, 2, 5, 0.5, 1.1, 1, 1, 1, 4, 4, 4, 4], np.float32)
    return base * scales[: shape[-1]] + offs[: shape[-1]]


def test_per_snr_mean_std_matches_numpy():
    import torch
    from amcpy_amd.postprocess import snr_statistics
    x = _feats((5, 16, 1000, 18), 3)                 # (mods, snr, frames, features)
    mean, std = snr_statistics(torch.from_numpy(x).cuda())
    assert mean.shape == (5, 16, 18) and mean.dtype == torch.float64
    # the reference's own loop: float32 np.mean / np.std per (mod, snr, feature)
    want_m = x.astype(np.float64).mean(axis=2)
    want_s = x.astype(np.float64).std(axis=2)
    assert np.allclose(mean.cpu().numpy(), want_m, rtol=1e-12, atol=0)
    assert np.allclose(std.cpu().numpy(), want_s, rtol=1e-10, atol=0)
    ref32_m = np.array([[[np.mean(x[i, j, :, k]) for k in range(18)] for j in range(16)] for i in range(5)])
    ref32_s = np.array([[[np.std(x[i, j, :, k]) for k in range(18)] for j in range(16)] for i in range(5)])
    assert np.allclose(mean.cpu().numpy(), ref32_m, rtol=2e-6, atol=0)
    assert np.allclose(std.cpu().numpy(), ref32_s, rtol=2e-5, atol=0)
    # a strided view (n_frames subset) must not be silently mis-read
    sub = torch.from_numpy(x).cuda()[:, :, :500]
    m2, _ = snr_statistics(sub)
    assert np.allclose(m2.cpu().numpy(), x[:, :, :500].astype(np.float64).mean(axis=2), rtol=1e-12)


def test_select_standardize_matches_sklearn():
    import torch
    from sklearn.preprocessing import StandardScaler
    from amcpy_amd.postprocess import select_standardize
    x = _feats((6 * 6 * 500, 18), 4)
    x[:, 13] = 7.25                                   # a constant column: sklearn gives scale 1
    used = [2, 4, 6, 8, 12, 14, 13]                   # list(FeatureConfig.used) + the constant one
    got, mean, scale = select_standardize(torch.from_numpy(x).cuda(), used)
    sc = StandardScaler()
    want = sc.fit_transform(x[:, used])
    assert np.allclose(mean.cpu().numpy(), sc.mean_, rtol=1e-12, atol=1e-12)
    assert np.allclose(scale.cpu().numpy(), sc.scale_, rtol=1e-10, atol=0)
    g = got.cpu().numpy()
    assert g.dtype == np.float32 and g.shape == want.shape
    assert np.allclose(g, want, rtol=2e-6, atol=2e-6)
    assert (g[:, -1] == 0).all()


def test_statistics_shapes_outliers_and_infinities():
    """Every cut of the chunked single-read statistics: ragged last tile, one row, one column, 32 columns, the whole
    benchmark matrix as ONE group (1 427 chunks merged), a huge outlier in the first row (what a shifted one-pass sum
    would lose), and an inf (numpy: mean inf, std nan)."""
    import torch
    from amcpy_amd import _lib
    from amcpy_amd.postprocess import snr_statistics
    rng = np.random.default_rng(11)
    for shape in [(3, 1, 18), (2, 447, 18), (2, 449, 18), (1, 5000, 1), (4, 700, 32), (1, 638976, 18), (7, 12345, 5)]:
        x = (rng.standard_normal(shape) * 3 + rng.standard_normal(shape[-1]) * 50).astype(np.float32)
        mean, std = snr_statistics(torch.from_numpy(x).cuda())
        x64 = x.astype(np.float64)
        assert np.allclose(mean.cpu().numpy(), x64.mean(axis=1), rtol=1e-12, atol=1e-13), shape
        assert np.allclose(std.cpu().numpy(), x64.std(axis=1), rtol=1e-11, atol=0), shape
    x = rng.standard_normal((2, 3000, 18)).astype(np.float32)
    x[0, 0, 3] = 1e30                                # outlier in the first row of a group
    x[1, 2999, 4] = -3e25                            # and in the last
    x[1, 17, 7] = np.inf
    x[0, 100, 9] = np.inf; x[0, 200, 9] = -np.inf
    mean, std = snr_statistics(torch.from_numpy(x).cuda())
    with np.errstate(invalid="ignore", over="ignore"):
        want_m, want_s = x.astype(np.float64).mean(axis=1), x.astype(np.float64).std(axis=1)
    assert np.allclose(mean.cpu().numpy(), want_m, rtol=1e-12, atol=1e-13, equal_nan=True)
    assert np.allclose(std.cpu().numpy(), want_s, rtol=1e-11, atol=0, equal_nan=True)
    assert np.isposinf(want_m[1, 7]) and np.isnan(want_s[1, 7]) and np.isnan(want_m[0, 9])
    # the entry without a workspace argument (stream-ordered allocation inside) gives the same doubles
    lib = _lib.load()
    xd = torch.from_numpy(x).cuda()
    m2 = torch.empty((2, 18), dtype=torch.float64, device="cuda"); s2 = torch.empty_like(m2)
    assert lib.amcx_group_stats_f32(xd.data_ptr(), 2, 3000, 18, 18, m2.data_ptr(), s2.data_ptr(),
                                    torch.cuda.current_stream().cuda_stream) == _lib.OK
    torch.cuda.synchronize()
    assert np.array_equal(m2.cpu().numpy(), mean.cpu().numpy(), equal_nan=True)
    assert np.array_equal(s2.cpu().numpy(), std.cpu().numpy(), equal_nan=True)
    # a workspace that is too small is refused, not overrun
    ws = torch.empty((64,), dtype=torch.uint8, device="cuda")
    assert lib.amcx_group_stats_ws_f32(xd.data_ptr(), 2, 3000, 18, 18, m2.data_ptr(), s2.data_ptr(), ws.data_ptr(), 64,
                                       None) == _lib.EINVAL


def test_select_standardize_benchmark_matrix_and_near_constant_columns():
    """The 638 976-row matrix of BASELINE configs[1] against sklearn, with a column that is constant up to float32
    noise far below its mean (sklearn's variance bound decides, not a fixed threshold) and a strided input view."""
    import torch
    from sklearn.preprocessing import StandardScaler
    from amcpy_amd.postprocess import select_standardize
    x = _feats((638976, 18), 5)
    x[:, 10] = np.float32(3.0e7)                      # constant, large
    x[::2, 11] = np.float32(1.0); x[1::2, 11] = np.nextafter(np.float32(1.0), np.float32(2.0))   # 1-ulp ripple
    used = [2, 4, 6, 8, 12, 14, 10, 11, 0, 17]
    wide = torch.zeros((638976, 20), dtype=torch.float32, device="cuda")
    wide[:, :18] = torch.from_numpy(x).cuda()
    got, mean, scale = select_standardize(wide[:, :18], used)          # row stride 20
    sc = StandardScaler()
    want = sc.fit_transform(x[:, used])
    assert np.allclose(mean.cpu().numpy(), sc.mean_, rtol=1e-12, atol=1e-12)
    assert np.allclose(scale.cpu().numpy(), sc.scale_, rtol=1e-9, atol=0), (scale.cpu().numpy(), sc.scale_)
    assert scale[6].item() == 1.0
    g = got.cpu().numpy()
    assert np.allclose(g, want, rtol=2e-6, atol=2e-6)


def test_postprocess_argument_errors():
    import torch
    from amcpy_amd import _lib
    from amcpy_amd.postprocess import select_standardize, snr_statistics
    lib = _lib.load()
    x = torch.zeros((4, 18), dtype=torch.float32, device="cuda")
    m = torch.zeros((1, 18), dtype=torch.float64, device="cuda")
    assert lib.amcx_group_stats_f32(x.data_ptr(), 1, 4, 18, 33, m.data_ptr(), m.data_ptr(), None) == _lib.EINVAL
    assert lib.amcx_group_stats_f32(x.data_ptr(), 1, 0, 18, 18, m.data_ptr(), m.data_ptr(), None) == _lib.EINVAL
    assert lib.amcx_group_stats_f32(None, 0, 4, 18, 18, None, None, None) == _lib.OK
    with pytest.raises(IndexError):
        select_standardize(x, [18])
    with pytest.raises(TypeError):
        snr_statistics(x.double())
