"""CPU oracle for the 18 per-frame IQ features  --  TEST INFRASTRUCTURE ONLY.

This module is the *checker* for the HIP path, never the thing shipped or
measured: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it.  Nothing under ``amcpy_amd/`` imports it.

It restates, in plain numpy/fp64, the arithmetic of the reference's
``src/amcpy/features.py`` (file:line cited per function below).  Parity status:
**pinned** -- (a) against the reference's own known-answer table
(features.py:286-305, restated as data in tests/golden/kat_n10.json) and
(b) against outputs of the reference itself, imported in the build container
by ``oracle/capture_golden.py`` and committed as ``tests/golden/*.npz``.

Third-party arithmetic on the path (not vendored in the reference; numpy
>=1.22 / scipy >=1.8 unpinned in pyproject.toml:37,43; 2.2.6 / 1.15.3 here):
``np.fft.fft``, ``np.angle``, ``np.unwrap`` are called exactly where the
reference calls them; ``scipy.stats.kurtosis(fisher=False)`` is restated
(`pearson_kurtosis`) from its published algorithm: biased m4/m2**2 about the
mean with the "m2 <= (eps*mean)**2 -> NaN" degenerate rule.

Two evaluators are provided:

* ``calculate_features(ids, signal)``  -- reference-shaped: one frame, each
  feature re-derives the instantaneous values / moments it needs, exactly the
  redundancy class of the reference (4x instantaneous, 9x moments per frame).
  This is what ``bench.py`` times as the ``cpu_baseline`` ("port").
* ``features18_batch(frames)``        -- fused and vectorised over frames,
  algebraically identical (wrapped first difference instead of unwrap+diff,
  kurt(a) instead of kurt(a/mu-1)); used by tests at sizes where the
  per-frame evaluator would take minutes.
"""

from __future__ import annotations

import math
from typing import Dict, Iterable, List, Sequence

import numpy as np

N_FEATURES = 18
_TWO_PI = 2.0 * math.pi


# --------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------

def instantaneous(signal: np.ndarray) -> Dict[str, np.ndarray]:
    """Envelope / phase / frequency series of one frame (features.py:17-31).

    Returns ``abs`` (N), ``phase`` (N), ``unwrapped`` (N), ``frequency``
    (N-1, cycles per sample) and ``cn_amplitude`` (N, a/mean(a) - 1).
    """
    x = np.asarray(signal)
    a = np.abs(x)                                   # :27
    th = np.angle(x)                                # :28
    un = np.unwrap(th)                              # :29
    with np.errstate(all="ignore"):
        cna = a / np.mean(a) - 1                    # :31
    return {"abs": a, "phase": th, "unwrapped": un,
            "frequency": np.diff(un) / _TWO_PI,     # :30
            "cn_amplitude": cna}


def mixed_moments(signal: np.ndarray) -> Dict[str, complex]:
    """M_pq = mean(x**(p-q) * conj(x)**q) (features.py:39-58).

    m21, m42 and m62 keep the REAL part only, as the reference does
    (features.py:47,53,57); m63 stays complex (features.py:58).
    """
    x = np.asarray(signal)
    c = np.conj(x)
    mean = np.mean
    return {
        "m20": mean(x ** 2), "m21": mean(x * c).real, "m22": mean(c ** 2),
        "m40": mean(x ** 4), "m41": mean(x ** 3 * c),
        "m42": mean(x ** 2 * c ** 2).real, "m43": mean(x * c ** 3),
        "m60": mean(x ** 6), "m61": mean(x ** 5 * c),
        "m62": mean(x ** 4 * c ** 2).real, "m63": mean(x ** 3 * c ** 3),
    }


def pearson_kurtosis(v: np.ndarray) -> float:
    """scipy.stats.kurtosis(v, fisher=False, bias=True) restated
    (call sites features.py:107,113; scipy 1.15.3 _stats_py.kurtosis)."""
    v = np.asarray(v)
    mu = v.mean()
    d = v - mu
    d2 = d * d
    m2 = d2.mean()
    m4 = (d2 * d2).mean()
    with np.errstate(all="ignore"):
        if m2 <= (np.finfo(m2.dtype).eps * mu) ** 2:
            return float("nan")
        return float(m4 / m2 ** 2.0)


def _std1(v: np.ndarray) -> float:
    with np.errstate(all="ignore"):
        return float(np.std(v, ddof=1))


# --------------------------------------------------------------------------
# the 18 features, reference-shaped (one function per id, recomputing inputs)
# --------------------------------------------------------------------------

def f01_gamma_max(x):      # features.py:66-69
    spec = np.abs(np.fft.fft(x))
    return float(np.max(spec ** 2 / len(x)))


def f02_sigma_ap(x):       # features.py:72-74
    return _std1(np.abs(np.angle(x)))


def f03_sigma_dp(x):       # features.py:77-79
    return _std1(np.angle(x))


def f04_sigma_aa(x):       # features.py:82-85
    return _std1(np.abs(instantaneous(x)["cn_amplitude"]))


def f05_sigma_af(x):       # features.py:88-91
    return _std1(instantaneous(x)["frequency"])


def f06_mean_mag(x):       # features.py:94-96
    return float(np.mean(np.abs(x)))


def f07_x2(x):             # features.py:99-101
    return float(np.sqrt(np.sum(np.abs(x))) / len(x))


def f08_kurt_a(x):         # features.py:104-107
    return pearson_kurtosis(instantaneous(x)["cn_amplitude"])


def f09_kurt_f(x):         # features.py:110-113
    return pearson_kurtosis(instantaneous(x)["frequency"])


def cumulant_terms(m: Dict[str, complex]) -> Dict[int, List[complex]]:
    """Additive terms of each cumulant, feature id -> list of terms whose sum
    is the (complex) cumulant; |sum| is the feature (features.py:116-185).
    Also used to form the conditioning scale S = sum(|term|)."""
    m20, m21, m22 = m["m20"], m["m21"], m["m22"]
    m40, m41, m42, m43 = m["m40"], m["m41"], m["m42"], m["m43"]
    m60, m61, m62, m63 = m["m60"], m["m61"], m["m62"], m["m63"]
    return {
        10: [m20],                                                    # :118
        11: [m21],                                                    # :123
        12: [m40, -3 * m20 ** 2],                                     # :129
        13: [m41, -3 * m20 * m21],                                    # :135
        14: [m42, -np.abs(m20) ** 2, -2 * m21 ** 2],                  # :141
        15: [m60, -15 * m20 * m40, 3 * m20 ** 3],                     # :147 (+3, not +30)
        16: [m61, -5 * m21 * m40, -10 * m20 * m41,
             30 * m20 ** 2 * m21],                                    # :154
        17: [m62, -6 * m20 * m42, -8 * m21 * m41, -m22 * m40,
             6 * m20 ** 2 * m22, 24 * m21 ** 2 * m20],                # :163-169
        18: [m63, -9 * m21 * m42, 12 * m21 ** 3, -3 * m20 * m43,
             -3 * m22 * m41, 18 * m20 * m21 * m22],                   # :178-184
    }


def _cumulant(fid: int):
    def fn(x):
        terms = cumulant_terms(mixed_moments(x))[fid]   # moments rebuilt per feature
        acc = terms[0]
        for t in terms[1:]:
            acc = acc + t
        return float(np.abs(acc))
    fn.__name__ = f"f{fid}_cumulant"
    return fn


FEATURE_TABLE = {
    1: f01_gamma_max, 2: f02_sigma_ap, 3: f03_sigma_dp, 4: f04_sigma_aa,
    5: f05_sigma_af, 6: f06_mean_mag, 7: f07_x2, 8: f08_kurt_a, 9: f09_kurt_f,
    **{fid: _cumulant(fid) for fid in range(10, 19)},
}


def calculate_features(feature_ids: Iterable[int], signal: np.ndarray) -> List[float]:
    """Reference-shaped evaluator (features.py:214-232): values in the order
    of ``feature_ids``; an unknown id raises ``KeyError``."""
    with np.errstate(all="ignore"):
        return [FEATURE_TABLE[fid](signal) for fid in feature_ids]


def features18_frame(signal: np.ndarray, dtype=np.complex128) -> np.ndarray:
    """All 18 features of one frame, evaluated in ``dtype`` and stored as
    float32 -- what one `_Worker` row assignment produces
    (feature_extraction.py:35,56)."""
    x = np.asarray(signal).astype(dtype, copy=False)
    return np.asarray(calculate_features(range(1, N_FEATURES + 1), x), dtype=np.float32)


# --------------------------------------------------------------------------
# fused, vectorised evaluator (same algebra, batch over frames)
# --------------------------------------------------------------------------

def wrapped_first_difference(theta: np.ndarray) -> np.ndarray:
    """diff(unwrap(theta)) without the prefix sum: d - 2*pi*rint(d/2pi).
    Half-to-even rounding reproduces numpy's tie rule (+pi stays +pi, -pi
    stays -pi) because |d| <= 2*pi."""
    d = np.diff(theta, axis=-1)
    return d - _TWO_PI * np.rint(d / _TWO_PI)


def batch_moments(frames: np.ndarray) -> Dict[str, np.ndarray]:
    """The 11 mixed moments for a (F, N) batch in complex128."""
    x = np.asarray(frames).astype(np.complex128, copy=False)
    x2 = x * x
    p = (x * np.conj(x)).real
    x4 = x2 * x2
    mean = lambda v: v.mean(axis=-1)   # noqa: E731
    m20, m41 = mean(x2), mean(x2 * p)
    return {
        "m20": m20, "m21": mean(p), "m22": np.conj(m20),
        "m40": mean(x4), "m41": m41, "m42": mean(p * p), "m43": np.conj(m41),
        "m60": mean(x4 * x2), "m61": mean(x4 * p),
        "m62": mean(x2.real * p * p), "m63": mean(p * p * p) + 0j,
    }


def conditioning_scales(frames: np.ndarray, absolute: bool = False) -> np.ndarray:
    """(F, 18) scale S per feature for the tolerance of SURVEY.md section 8c:
    S = sum(|term|) of the cumulant's formula for ids 12..18, S = m21 for id 10,
    S = |value| (i.e. plain relative) for every other id (returned as 0 so the
    caller takes max(|golden|, S)).

    ``absolute=True`` bounds every complex moment by the mean of its summands'
    magnitudes first (|m20| <= m21, |m40|,|m41| <= m42, |m60|,|m61|,|m62| <=
    m63): the conditioning of the moment sums themselves.  Needed only for
    degenerate frames (a pure tone: every complex moment cancels to ~1e-17 in
    fp64), where the plain term sum collapses to rounding dust."""
    m = batch_moments(frames)
    if absolute:
        m21, m42, m63 = np.abs(m["m21"]), np.abs(m["m42"]), np.abs(m["m63"])
        m = {"m20": m21, "m21": m21, "m22": m21, "m40": m42, "m41": m42, "m42": m42,
             "m43": m42, "m60": m63, "m61": m63, "m62": m63, "m63": m63}
    terms = cumulant_terms(m)
    F = np.asarray(frames).shape[0]
    S = np.zeros((F, N_FEATURES))
    S[:, 9] = np.abs(m["m21"])
    for fid in range(12, 19):
        S[:, fid - 1] = sum(np.abs(t) for t in terms[fid])
    return S


def features18_batch(frames: np.ndarray) -> np.ndarray:
    """(F, N) complex -> (F, 18) float64, fused formulation in fp64."""
    x = np.asarray(frames).astype(np.complex128, copy=False)
    if x.ndim == 1:
        x = x[None, :]
    F, N = x.shape
    out = np.empty((F, N_FEATURES))
    with np.errstate(all="ignore"):
        a = np.abs(x)
        th = np.angle(x)
        phi = wrapped_first_difference(th) / _TWO_PI
        mu = a.mean(axis=-1)

        def std1(v):
            return np.sqrt(((v - v.mean(axis=-1, keepdims=True)) ** 2).sum(axis=-1)
                           / (v.shape[-1] - 1))

        def kurt(v, mean_for_rule):
            d = v - v.mean(axis=-1, keepdims=True)
            m2 = (d ** 2).mean(axis=-1)
            m4 = (d ** 4).mean(axis=-1)
            bad = m2 <= (np.finfo(np.float64).eps * mean_for_rule) ** 2
            return np.where(bad, np.nan, m4 / m2 ** 2)

        out[:, 0] = (np.abs(np.fft.fft(x, axis=-1)) ** 2).max(axis=-1) / N
        out[:, 1] = std1(np.abs(th))
        out[:, 2] = std1(th)
        cna = a / mu[:, None] - 1
        out[:, 3] = std1(np.abs(cna))
        out[:, 4] = std1(phi)
        out[:, 5] = mu
        out[:, 6] = np.sqrt(a.sum(axis=-1)) / N
        out[:, 7] = kurt(cna, cna.mean(axis=-1))
        out[:, 8] = kurt(phi, phi.mean(axis=-1))
        terms = cumulant_terms(batch_moments(x))
        for fid in range(10, 19):
            out[:, fid - 1] = np.abs(sum(terms[fid]))
    return out


# --------------------------------------------------------------------------
# tolerance helper shared by every parity test
# --------------------------------------------------------------------------

def parity_errors(got: np.ndarray, golden: np.ndarray, scales: np.ndarray):
    """Return (plain_rel, scaled_rel), both (F, 18).

    scaled_rel = |got-golden| / max(|golden|, S): S-scaled for the
    cancellation-dominated cumulants, plain relative elsewhere.  Positions
    where golden is NaN must be NaN in ``got`` (error 0) -- otherwise inf.
    """
    got = np.asarray(got, dtype=np.float64)
    golden = np.asarray(golden, dtype=np.float64)
    diff = np.abs(got - golden)
    with np.errstate(all="ignore"):
        plain = diff / np.abs(golden)
        scaled = diff / np.maximum(np.abs(golden), scales)
    both_nan = np.isnan(got) & np.isnan(golden)
    one_nan = np.isnan(got) ^ np.isnan(golden)
    exact = diff == 0
    for e in (plain, scaled):
        e[both_nan | exact] = 0.0
        e[one_nan] = np.inf
    return plain, scaled
