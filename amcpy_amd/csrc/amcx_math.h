// Per-sample fp32 device math shared by every amcx kernel, plus the fp64
// per-frame finaliser that turns reduced sums into the 18 features.
// gfx950 (CDNA4) only.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace amcx {

constexpr float kPi = 3.14159265358979323846f;
constexpr float kHalfPi = 1.57079632679489661923f;
constexpr float kTwoPi = 6.28318530717958647692f;
constexpr float kInvTwoPi = 0.15915494309189533577f;
constexpr double kTwoPiD = 6.283185307179586476925286766559;

// ---------------------------------------------------------------------------
// angle(x) = atan2(im, re) in (-pi, pi], numpy semantics on the axes:
// angle(0) = 0, angle(-1+0j) = +pi, angle(-1-0j) = -pi  (reference
// features.py:28 -> np.angle).
//
// Half-angle form on the envelope the caller already has (a = |x|):
//   u = im / (a + |re|) = tan(phi/2),  phi = atan2(im, |re|) in [-pi/2, pi/2], |u| <= 1
//   theta = phi                      (re >= 0)
//         = copysign(pi, im) - phi   (re <  0, incl. -0)
// -- no min/max/swap, the odd polynomial carries the sign: ~16 plain VALU ops
// plus one v_rcp_f32, against ~23 for the octant-reduction form (min/max,
// compares and selects issue at 2/3 the rate of mul/add/fma on gfx950,
// profiles/r1_valu_issue_rates.txt).  The denominator is a sum of
// non-negatives, so nothing cancels; max abs error 2.5e-7 rad.
// `a` must be > 0: callers form it as sqrt(re^2 + im^2 + kTinyPower), which
// leaves every normal input unchanged and makes angle(0) = 0 fall out.
// NaN inputs are not tracked here: a frame with a non-finite sample is turned
// into 18 NaNs by the finaliser, from the NaN its power sum carries.
// ---------------------------------------------------------------------------
constexpr float kTinyPower = 1.0e-37f;

__device__ __forceinline__ float fast_angle(float re, float im, float a) {
  const float den = a + __builtin_fabsf(re);
  const float u = im * __builtin_amdgcn_rcpf(den);
  const float s = u * u;
  // 2*atan(u)/u, 8-term minimax on [0,1] (tools/fit_atan.py coefficients x 2)
  float q = -8.109134424e-03f;
  q = __builtin_fmaf(q, s, 4.372591574e-02f);
  q = __builtin_fmaf(q, s, -1.118246535e-01f);
  q = __builtin_fmaf(q, s, 1.928439466e-01f);
  q = __builtin_fmaf(q, s, -2.781725910e-01f);
  q = __builtin_fmaf(q, s, 3.989313130e-01f);
  q = __builtin_fmaf(q, s, -6.665971279e-01f);   // +1 ulp: makes 2*atan(1) evaluate to fl(pi/2) exactly
  q = __builtin_fmaf(q, s, 1.999998671e+00f);
  const float phi = q * u;
  const float flip = __builtin_copysignf(kPi, im) - phi;
  return __builtin_signbitf(re) ? flip : phi;
}

// diff(unwrap(theta))[i] from two neighbouring angles: d - 2*pi*rint(d/2pi),
// half-to-even reproduces numpy's tie rule since |d| <= 2*pi
// (reference features.py:29-30 -> np.unwrap/np.diff).
// rint is spelled (t + 1.5*2^23) - 1.5*2^23 -- an fma and a subtract on the
// full-rate fp32 pipe instead of v_rndne_f32, which issues at 2/3 of that rate
// (profiles/r1_valu_issue_rates.txt); exact for |t| < 2^22, and |t| < 1 here.
__device__ __forceinline__ float wrapped_step(float th_next, float th) {
  constexpr float kMagic = 12582912.0f;
  const float d = th_next - th;
  const float k = __builtin_fmaf(d, kInvTwoPi, kMagic) - kMagic;
  return __builtin_fmaf(-kTwoPi, k, d);
}

// |step| > pi - kTieBand: rounding of the two fp32 angles (<= 1e-6 together) may
// have decided the sign of the wrapped step.  The fp64 reference resolves such a
// step from the 17th digit; the throughput kernel's sweep only FLAGS the frame (the
// finaliser sees -f5, and f5 >= 0) and the finaliser recomputes f5/f9 of flagged frames
// with exact_step below (wave_exact_frequency).  About 1 frame in 400 is flagged at low
// SNR; every frame of noiseless axis-aligned data is.
constexpr float kTieBand = 2.0e-6f;

// Wrapped step p -> q with an exact tie decision: sign(sin(step)) = sign(Re p Im q -
// Im p Re q), the cross product evaluated with two error-free products, so the
// decision matches the fp64 reference down to ~1e-14 rad; an exact zero
// (antiparallel samples) follows numpy's unwrap rule, the sign of the raw difference.
__device__ __forceinline__ float exact_step(float th_q, float th_p, float re_p, float im_p,
                                            float re_q, float im_q) {
  const float w = wrapped_step(th_q, th_p);
  if (!(__builtin_fabsf(w) > kPi - kTieBand)) return w;
  const float p1 = re_p * im_q, p2 = im_p * re_q;
  const float e1 = __builtin_fmaf(re_p, im_q, -p1), e2 = __builtin_fmaf(im_p, re_q, -p2);
  const float c = (p1 - p2) + (e1 - e2);
  const float sgn = (c != 0.f) ? c : (th_q - th_p);
  return __builtin_copysignf(__builtin_fabsf(w), sgn);
}

// ---------------------------------------------------------------------------
// Reduced per-frame sums, everything the 18 features need.
// A = re^2 - im^2, Bh = re*im (so x^2 = A + 2i*Bh), P = re^2 + im^2.
// Re(x^4) = A^2 - 4 Bh^2 is summed as such, not as two large positive sums that
// cancel: for noise-like frames m40 and m61 are near zero and would otherwise
// inherit the rounding of sums the size of m42 / m63.
// ---------------------------------------------------------------------------
struct FrameSums {
  // mixed-moment sums over the N samples
  double sA, sBh, sP;            // -> m20, m21
  double sAA, sX4, sAB;          // A^2, A^2 - 4 Bh^2 (= Re x^4), A*Bh -> m40, m42
  double sAP, sBP;               // A*P, Bh*P                -> m41
  double sAAA, sABB, sAAB, sBBB; // A^3, A*Bh^2, A^2*Bh, Bh^3 -> m60, m62
  double sAAP, sX4P, sABP;       // A^2*P, Re(x^4)*P, A*Bh*P  -> m61, m63
  // envelope a = |x| : exact mean, then centred sums about it
  double sa;                     // sum a
  double sad1, sad2, sad4;       // sum |a-mu|, (a-mu)^2, (a-mu)^4
  // phase theta, shifted by Kt: d = theta - Kt
  double Kt, std1, std2;         // shift, sum d, sum d^2
  double Ka, sab1, sab2;         // |theta| shifted by Ka: sum e, sum e^2 with e = |theta| - Ka
  // wrapped phase step w (N-1 values), shifted by Kw: d = w - Kw
  double Kw, swd1, swd2, swd3, swd4;
  bool pi_tie = false;           // some step within kTieBand of +-pi: f5 is stored negated as the flag
  double gmax_raw;               // max_k |X_k|^2 (unnormalised FFT)
};

// Range in which the throughput kernel's fp32 sums are trusted, as the frame's mean power
// sum|x|^2 / N: sixth-order products of a frame at 1e10 stay below 3.4e38 unless one sample
// holds most of its energy (then a sum overflows and is_outside_fp32_range sees the inf), and at
// 1e-10 they are still 8 orders above the smallest normal float.  Frames outside -- and frames
// any of whose sums is not finite -- are flagged (f5 = -inf) and recomputed with fp64 sums by
// the range pass: the reference evaluates in complex128 (features.py:46-58) and is finite
// over the whole complex64 range, overflowing only in its float32 store.
constexpr double kRangeLoPower = 1.0e-10, kRangeHiPower = 1.0e10;

__device__ inline bool is_outside_fp32_range(const FrameSums& s, int N) {
  if (!(s.sP == s.sP)) return false;            // a NaN sample: 18 NaNs on either path
  double z = 0.0;                               // 0 * x is NaN for x = +-inf or NaN
  z = __builtin_fma(s.sA, 0.0, z);   z = __builtin_fma(s.sBh, 0.0, z);  z = __builtin_fma(s.sP, 0.0, z);
  z = __builtin_fma(s.sAA, 0.0, z);  z = __builtin_fma(s.sX4, 0.0, z);  z = __builtin_fma(s.sAB, 0.0, z);
  z = __builtin_fma(s.sAP, 0.0, z);  z = __builtin_fma(s.sBP, 0.0, z);  z = __builtin_fma(s.sAAA, 0.0, z);
  z = __builtin_fma(s.sABB, 0.0, z); z = __builtin_fma(s.sAAB, 0.0, z); z = __builtin_fma(s.sBBB, 0.0, z);
  z = __builtin_fma(s.sAAP, 0.0, z); z = __builtin_fma(s.sX4P, 0.0, z); z = __builtin_fma(s.sABP, 0.0, z);
  z = __builtin_fma(s.sa, 0.0, z);   z = __builtin_fma(s.sad2, 0.0, z); z = __builtin_fma(s.sad4, 0.0, z);
  z = __builtin_fma(s.gmax_raw, 0.0, z);
  const double n = (double)N;
  // An all-zero frame shows as N samples of power kTinyPower with every angle exactly 0; samples below
  // ~3e-19, whose squares underflow, show the same power but not the same angles (only a frame of
  // non-negative reals that small is indistinguishable from zeros in fp32).  The former stays, the
  // latter is flagged like any other out-of-range frame.
  const bool zero_frame = s.sP <= 2.0 * n * (double)kTinyPower && s.std2 == 0.0 && s.sab2 == 0.0 && s.Kt == 0.0;
  const double pbar = s.sP / n;
  return (z != z) || (!zero_frame && !(pbar >= kRangeLoPower && pbar <= kRangeHiPower));
}

// The finaliser's results are stored as float32, so its square roots and quotients
// need ~1e-7, not 1e-16: the raw v_sqrt_f64 / v_rcp_f64 (accurate to about float
// precision) replace the ~20-instruction IEEE expansions hipcc emits for sqrt() and
// '/', cutting the finaliser from ~500 to ~200 instructions per batch.  Special values
// behave the same way (0 * rcp(0) = NaN, x * rcp(0) = inf).
__device__ __forceinline__ double q_sqrt(double x) { return __builtin_amdgcn_sqrt(x); }
__device__ __forceinline__ double q_div(double a, double b) { return a * __builtin_amdgcn_rcp(b); }
constexpr double kInvTwoPiD = 1.0 / 6.283185307179586476925286766559;

// ---------------------------------------------------------------------------
// CANCELLATION.  The cumulants |C40|, |C41|, |C60|, |C61|, |C62| (ids 12, 13, 15, 16, 17) are sums of terms that can cancel
// to any degree: over thousands of noise-like frames a few have a whole sixth-order moment 1000x below the size of its
// summands, by chance.  The throughput kernels' fp32 sums are good to a few 1e-9 of the SUMMANDS' scale (measured tail
// over 135 000 cancelled cumulants at N = 2048: 99 % below 1.4e-8, 99.9 % below 2.2e-8, largest 5.2e-8 --
// tests/manual/cancel_study.py, profiles/r6_cancel_study_*.txt), not of such a sum; the parity contract is 1e-5 of
// S = sum |terms| (the reference evaluates in complex128, features.py:144-185).  So the finaliser asks,
// per frame, whether any of the five has  S < kappa E,  E the first-order error scale of that cumulant (every moment's
// absolute error taken as one unit of the mean of its summands' magnitudes: m21, m42, m63), and a frame that does gets
// its 15 moment sums again from an fp64 sweep by the whole wave, in the same launch (wave_exact_moments,
// amcx_wave_kernel.h) -- the mechanism the +-pi ties already use for f5 / f9.  Ids 10, 11, 14, 18 cannot cancel below
// ~1/7 of their E (m21, m42 + 2 m21^2, m63 + 9 m21 m42 + 12 m21^3 are sums of non-negatives).
// kappa = 4e-3 (2048 / N)^(1/3): the measured tail shrinks slowly with the frame size while the share of frames below a
// fixed kappa grows like N; at this kappa 0.3 ... 0.6 % of the BASELINE configs' frames are flagged (1.0 % at N = 4096),
// and the chance that an UNFLAGGED frame misses 1e-5 S, extrapolating the measured tail, is ~1e-8 per frame.
// The predicate is conservative: the leading moment enters exactly (as a square), the other complex moments through
// max(|re|, |im|) <= |z| <= |re| + |im| (|m40|: the octagon, within 8 %) -- S from below, E from above.  NaN / inf moments compare false: those frames have
// their own paths.
// ---------------------------------------------------------------------------
constexpr double cancel_kappa(int N) {
  double k = 4.0e-3;
  for (int n = N; n < 2048; n *= 2) k *= 1.2599210498948732;
  for (int n = N; n > 2048; n /= 2) k /= 1.2599210498948732;
  return k;
}

// Evaluated in fp32 on the 15 reduced sums as they lie in the stash (sA, sBh, sP, sAA, sX4, sAB, sAP, sBP, sAAA, sABB, sAAB,
// sBBB, sAAP, sX4P, sABP), in units of the frame's mean power (m21 = 1: m20 / m21 = sA / sP, m4x / m21^2 = n s / sP^2,
// m6x / m21^3 = n^2 s / sP^3 -- nothing overflows anywhere in the range the fp32 sums are trusted in), ahead of the fp64
// algebra and independent of it: ~85 fp32 instructions per finaliser batch and a dozen registers (as part of
// finalize_features, in fp64, the 128-register wave kernels spilled five doubles per batch).
__device__ __forceinline__ bool cancellation_suspect(const float (&s)[15], float n, float kappa) {
  const float c2 = __builtin_amdgcn_rcpf(s[2]);
  const float c4 = n * c2 * c2, c6 = c4 * (n * c2);
  const float b20r = s[0] * c2, b20i = 2.0f * s[1] * c2;
  const float n20 = __builtin_fmaf(b20r, b20r, b20i * b20i), a20 = __builtin_amdgcn_sqrtf(n20);
  const float b40r = s[4] * c4, b40i = 4.0f * s[5] * c4, b41r = s[6] * c4, b41i = 2.0f * s[7] * c4;
  const float b42 = __builtin_fmaf(2.0f, s[3], -s[4]) * c4;
  const float b60r = __builtin_fmaf(-12.0f, s[9], s[8]) * c6, b60i = __builtin_fmaf(6.0f, s[10], -8.0f * s[11]) * c6;
  const float b61r = s[13] * c6, b61i = 4.0f * s[14] * c6;
  const float b62 = __builtin_fmaf(4.0f, s[9], s[8]) * c6, b63 = __builtin_fmaf(2.0f, s[12], -s[13]) * c6;
  // |m40| sits in three of the five tests: the octagon max(max(|re|, |im|), (|re| + |im|) / sqrt 2) <= |z| <= 1.0824 x that
  const float s40 = __builtin_fabsf(b40r) + __builtin_fabsf(b40i);
  const float a40l = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(b40r), __builtin_fabsf(b40i)), 0.70710678f * s40), a40h = 1.0823922f * a40l;
  const float a41l = __builtin_fmaxf(__builtin_fabsf(b41r), __builtin_fabsf(b41i)), a41h = __builtin_fabsf(b41r) + __builtin_fabsf(b41i);
  const float g = a20 * b42, u = a20 * a40l, c3 = a20 * n20, v = a20 * a41l;
  // id 12: m40 - 3 m20^2                       S >= |m40| + 3 |m20|^2,  E = m42 + 6 |m20| m21
  const float t12 = __builtin_fmaf(kappa, __builtin_fmaf(6.0f, a20, b42), -3.0f * n20);
  // id 13: m41 - 3 m20 m21                     E = m42 + 3 m21^2 + 3 |m20| m21
  const float t13 = __builtin_fmaf(kappa, __builtin_fmaf(3.0f, a20, b42 + 3.0f), -3.0f * a20);
  // id 15: m60 - 15 m20 m40 + 3 m20^3          E = m63 + 15 (|m20| m42 + |m40| m21) + 9 |m20|^2 m21
  const float e15 = __builtin_fmaf(9.0f, n20, __builtin_fmaf(15.0f, g + a40h, b63));
  const float t15 = __builtin_fmaf(kappa, e15, __builtin_fmaf(-15.0f, u, -3.0f * c3));
  // id 16: m61 - 5 m21 m40 - 10 m20 m41 + 30 m20^2 m21
  //        E = m63 + 5 (m21 m42 + |m40| m21) + 10 (|m20| m42 + |m41| m21) + 30 (2 |m20| m21^2 + |m20|^2 m21)
  const float e16 = __builtin_fmaf(60.0f, a20, __builtin_fmaf(30.0f, n20, __builtin_fmaf(10.0f, g + a41h, __builtin_fmaf(5.0f, b42 + a40h, b63))));
  const float t16 = __builtin_fmaf(kappa, e16, __builtin_fmaf(-5.0f, a40l, __builtin_fmaf(-10.0f, v, -30.0f * n20)));
  // id 17: m62 - 6 m20 m42 - 8 m21 m41 - m22 m40 + 6 m20^2 m22 + 24 m21^2 m20
  //        E = m63 + 6 (|m20| m42 + m42 m21) + 8 (m21 m42 + |m41| m21) + (|m20| m42 + |m40| m21) + 18 |m20|^2 m21 + 24 (2 m21^2 |m20| + m21^3)
  const float e17 = __builtin_fmaf(48.0f, a20, __builtin_fmaf(18.0f, n20, __builtin_fmaf(8.0f, a41h, __builtin_fmaf(7.0f, g,
                    __builtin_fmaf(14.0f, b42, b63 + a40h + 24.0f)))));
  const float t17 = __builtin_fmaf(kappa, e17, __builtin_fmaf(-6.0f, g + c3, __builtin_fmaf(-8.0f, a41l, __builtin_fmaf(-24.0f, a20, -u))));
  bool f = t12 > 0.f && __builtin_fmaf(b40r, b40r, b40i * b40i) < t12 * t12;
  f |= t13 > 0.f && __builtin_fmaf(b41r, b41r, b41i * b41i) < t13 * t13;
  f |= t15 > 0.f && __builtin_fmaf(b60r, b60r, b60i * b60i) < t15 * t15;
  f |= t16 > 0.f && __builtin_fmaf(b61r, b61r, b61i * b61i) < t16 * t16;
  f |= __builtin_fabsf(b62) < t17;
  return f;
}

// f5 = std1(phi), f9 = kurt(phi) of phi = w / 2pi from sums of d = w - Kw over the N-1 steps
__device__ inline void frequency_features(double Kw, double swd1, double swd2, double swd3, double swd4,
                                          int N, float& f5, float& f9) {
  const double n1 = (double)N - 1.0, inv1 = 1.0 / n1;
  const double d1 = swd1 * inv1;                        // mean of shifted w
  const double r2 = swd2 * inv1, r3 = swd3 * inv1, r4 = swd4 * inv1;
  double c2 = r2 - d1 * d1;                             // central moments of w
  if (c2 < 0) c2 = 0;
  const double c4 = r4 - 4.0 * d1 * r3 + 6.0 * d1 * d1 * r2 - 3.0 * d1 * d1 * d1 * d1;
  f5 = (float)(q_sqrt(c2 * (n1 * (1.0 / (n1 - 1.0)))) * kInvTwoPiD);   // N is a constant where it matters
  const double wbar = (Kw + d1) * kInvTwoPiD;            // mean phi, for scipy's rule
  const double m2phi = c2 * (kInvTwoPiD * kInvTwoPiD);
  const double eps_mean = 2.220446049250313e-16 * wbar;
  f9 = (m2phi <= eps_mean * eps_mean) ? __builtin_nanf("") : (float)q_div(c4, c2 * c2);
}

// All 18 features from the sums, fp64 (the reference evaluates in fp64 and
// stores float32: feature_extraction.py:35,56).  Formulas: features.py:66-185;
// C60 uses +3*m20^3 and m62 is real-only, as the reference has them
// (features.py:147,57).
// SCALED (the wave kernel's re-run of an out-of-range frame): the sums are those of the frame multiplied by 2^-ex, ex even;
// feature j of the frame itself is the scaled one times 2^(ex * order_j) with order = 2, 0, 0, 0, 0, 1, 1/2, 0, 0, 2, 2,
// 4, 4, 4, 6, 6, 6, 6 -- applied in fp64 before the float32 store, so that store overflows / underflows exactly where the
// reference's does (feature_extraction.py:35,56).
// Returns whether the frame is an ordinary one (false: a non-finite sample, or all zeros -- no cumulant to refine).
template <bool SCALED = false>
__device__ inline bool finalize_features(const FrameSums& s, int N, float* __restrict__ out, int ex = 0) {
  [[maybe_unused]] const int h = ex / 2;                 // ex is even: 2^(h * twice_order) is exact
  auto put = [&](int j, int twice_order, double v) {
    if constexpr (SCALED) v = __builtin_ldexp(v, h * twice_order);
    out[j] = (float)v;
  };
  const double n = (double)N;
  const double inv = 1.0 / n, inv_nm1 = 1.0 / (n - 1.0);
  // non-finite input anywhere -> the reference's numpy arithmetic yields NaN
  // in every feature (SURVEY.md Appendix C "one NaN sample")
  if (!(__builtin_fabs(s.sP) <= 1.79e308) || !(s.gmax_raw == s.gmax_raw)) {
#pragma unroll
    for (int j = 0; j < 18; ++j) out[j] = __builtin_nanf("");
    return false;
  }
  // ---- f1: gamma_max
  put(0, 4, s.gmax_raw * inv);

  // ---- phase: f2 = std1(|theta|), f3 = std1(theta); both from sums about a shift
  // close to the mean (no E[v^2] - E[v]^2 on raw values: |theta| can have a tiny
  // variance around pi/2 while theta itself spans +-pi)
  const double md = s.std1 * inv;                       // mean of shifted theta
  double ct2 = s.std2 - n * md * md;                    // sum (theta-mean)^2
  if (ct2 < 0) ct2 = 0;
  const double ma = s.sab1 * inv;                       // mean of shifted |theta|
  double cabs2 = s.sab2 - n * ma * ma;                  // sum (|theta|-mean)^2
  if (cabs2 < 0) cabs2 = 0;
  put(1, 0, q_sqrt(cabs2 * inv_nm1));
  put(2, 0, q_sqrt(ct2 * inv_nm1));

  // ---- envelope: f4, f6, f7, f8
  // an all-zero frame reaches here as N samples of power kTinyPower (the angle
  // guard): restore the exact zeros so that f4 is 0/0 = NaN as in the reference
  const bool zero_frame = s.sP <= 2.0 * n * (double)kTinyPower;
  const double mu = zero_frame ? 0.0 : s.sa * inv;
  {
    const double mad = s.sad1 * inv;                    // mean |a-mu|
    double v = s.sad2 - n * mad * mad;                  // sum (|a-mu| - mad)^2
    if (v < 0) v = 0;
    put(3, 0, q_div(q_sqrt(v * inv_nm1), mu));            // 0/0 -> NaN for a zero frame
    put(5, 2, mu);
    put(6, 1, q_sqrt(zero_frame ? 0.0 : s.sa) * inv);
    const double m2 = s.sad2 * inv, m4 = s.sad4 * inv;
    put(7, 0, q_div(m4, m2 * m2));                      // m2 == 0 -> NaN (scipy rule)
  }

  // ---- frequency phi = w / 2pi over N-1 values: f5, f9
  frequency_features(s.Kw, s.swd1, s.swd2, s.swd3, s.swd4, N, out[4], out[8]);

  if (s.pi_tie) out[4] = -out[4];   // picked up by the wave finaliser (wave_exact_frequency)

  // ---- mixed moments (complex as (re, im) pairs)
  if (zero_frame) {   // the guard's kTinyPower must not leak into |C20| ... |C63| of a zero frame
#pragma unroll
    for (int j = 9; j < 18; ++j) out[j] = 0.f;
    return false;
  }
  const double m20r = s.sA * inv, m20i = 2.0 * s.sBh * inv;
  const double m21 = s.sP * inv;
  const double m40r = s.sX4 * inv, m40i = 4.0 * s.sAB * inv;
  const double m41r = s.sAP * inv, m41i = 2.0 * s.sBP * inv;
  const double m42 = (2.0 * s.sAA - s.sX4) * inv;                     // mean P^2 = A^2 + 4 Bh^2
  const double m60r = (s.sAAA - 12.0 * s.sABB) * inv;                 // Re (A+iB)^3, B = 2Bh
  const double m60i = (6.0 * s.sAAB - 8.0 * s.sBBB) * inv;
  const double m61r = s.sX4P * inv, m61i = 4.0 * s.sABP * inv;
  const double m62 = (s.sAAA + 4.0 * s.sABB) * inv;                   // mean A*P^2 (real only)
  const double m63 = (2.0 * s.sAAP - s.sX4P) * inv;                   // mean P^3
  // m22 = conj(m20), m43 = conj(m41)

  auto cabs = [](double r, double i) { return q_sqrt(r * r + i * i); };
  const double q20r = m20r * m20r - m20i * m20i, q20i = 2.0 * m20r * m20i;   // m20^2
  const double n20 = m20r * m20r + m20i * m20i;                              // |m20|^2

  put(9, 4, cabs(m20r, m20i));                                               // C20
  put(10, 4, __builtin_fabs(m21));                                           // C21
  put(11, 8, cabs(m40r - 3.0 * q20r, m40i - 3.0 * q20i));                    // C40
  put(12, 8, cabs(m41r - 3.0 * m20r * m21, m41i - 3.0 * m20i * m21));        // C41
  put(13, 8, __builtin_fabs(m42 - n20 - 2.0 * m21 * m21));                   // C42
  {  // C60 = m60 - 15 m20 m40 + 3 m20^3
    const double pr = m20r * m40r - m20i * m40i, pi = m20r * m40i + m20i * m40r;
    const double cr = q20r * m20r - q20i * m20i, ci = q20r * m20i + q20i * m20r;
    put(14, 12, cabs(m60r - 15.0 * pr + 3.0 * cr, m60i - 15.0 * pi + 3.0 * ci));
  }
  {  // C61 = m61 - 5 m21 m40 - 10 m20 m41 + 30 m20^2 m21
    const double pr = m20r * m41r - m20i * m41i, pi = m20r * m41i + m20i * m41r;
    put(15, 12, cabs(m61r - 5.0 * m21 * m40r - 10.0 * pr + 30.0 * q20r * m21,
                     m61i - 5.0 * m21 * m40i - 10.0 * pi + 30.0 * q20i * m21));
  }
  {  // C62 = m62 - 6 m20 m42 - 8 m21 m41 - m22 m40 + 6 m20^2 m22 + 24 m21^2 m20
    // m22 m40 = conj(m20) m40 ; m20^2 m22 = m20 |m20|^2
    const double ar = m20r * m40r + m20i * m40i, ai = m20r * m40i - m20i * m40r;
    const double re = m62 - 6.0 * m20r * m42 - 8.0 * m21 * m41r - ar + 6.0 * m20r * n20 +
                      24.0 * m21 * m21 * m20r;
    const double im = -6.0 * m20i * m42 - 8.0 * m21 * m41i - ai + 6.0 * m20i * n20 +
                      24.0 * m21 * m21 * m20i;
    put(16, 12, cabs(re, im));
  }
  {  // C63 = m63 - 9 m21 m42 + 12 m21^3 - 3 m20 m43 - 3 m22 m41 + 18 m20 m21 m22
    // m20 conj(m41) + conj(m20) m41 = 2 Re(m20 conj(m41)), real
    const double cross = 2.0 * (m20r * m41r + m20i * m41i);
    put(17, 12, __builtin_fabs(m63 - 9.0 * m21 * m42 + 12.0 * m21 * m21 * m21 -
                               3.0 * cross + 18.0 * m21 * n20));
  }
  return true;
}

// finalize_features in PIECES, each from the sums it needs alone, for a caller whose sums come from several places
// (amcx_group_kernel.h: W stash rows per frame) and who fetches them group by group instead of holding all thirty at once
// -- at 128 registers the monolithic form spilled ~100 values per call there.  The same formulas, statement for
// statement; finalize_features itself is left exactly as it is, because re-expressing it through these pieces changes the
// machine code of every wave kernel (tools/codeobj_gate.py --kernels), and those are tuned to the register.
// put(j, twice_order, v) stores feature j (0-based); twice_order: see finalize_features<SCALED>.
//
// f2 = std1(|theta|), f3 = std1(theta); both from sums about a shift close to the mean (no E[v^2] - E[v]^2 on raw values:
// |theta| can have a tiny variance around pi/2 while theta itself spans +-pi)
template <class Put>
__device__ __forceinline__ void phase_features(double std1, double std2, double sab1, double sab2, double n, Put&& put) {
  const double inv = 1.0 / n, inv_nm1 = 1.0 / (n - 1.0);
  const double md = std1 * inv;                         // mean of shifted theta
  double ct2 = std2 - n * md * md;                      // sum (theta-mean)^2
  if (ct2 < 0) ct2 = 0;
  const double ma = sab1 * inv;                         // mean of shifted |theta|
  double cabs2 = sab2 - n * ma * ma;                    // sum (|theta|-mean)^2
  if (cabs2 < 0) cabs2 = 0;
  put(1, 0, q_sqrt(cabs2 * inv_nm1));
  put(2, 0, q_sqrt(ct2 * inv_nm1));
}

// envelope: f4, f6, f7, f8.  An all-zero frame reaches here as N samples of power kTinyPower (the angle guard): the exact
// zeros are restored so that f4 is 0/0 = NaN as in the reference
template <class Put>
__device__ __forceinline__ void envelope_features(double sa, double sad1, double sad2, double sad4, bool zero_frame, double n,
                                                  Put&& put) {
  const double inv = 1.0 / n, inv_nm1 = 1.0 / (n - 1.0);
  const double mu = zero_frame ? 0.0 : sa * inv;
  const double mad = sad1 * inv;                        // mean |a-mu|
  double v = sad2 - n * mad * mad;                      // sum (|a-mu| - mad)^2
  if (v < 0) v = 0;
  put(3, 0, q_div(q_sqrt(v * inv_nm1), mu));            // 0/0 -> NaN for a zero frame
  put(5, 2, mu);
  put(6, 1, q_sqrt(zero_frame ? 0.0 : sa) * inv);
  const double m2 = sad2 * inv, m4 = sad4 * inv;
  put(7, 0, q_div(m4, m2 * m2));                        // m2 == 0 -> NaN (scipy rule)
}

// |C20| ... |C63| from the 15 mixed-moment sums (complex as (re, im) pairs).  Formulas: features.py:116-185; C60 uses
// +3*m20^3 and m62 is real-only, as the reference has them (features.py:147,57).
template <class Put>
__device__ __forceinline__ void moment_features(double sA, double sBh, double sP, double sAA, double sX4, double sAB, double sAP,
                                                double sBP, double sAAA, double sABB, double sAAB, double sBBB, double sAAP,
                                                double sX4P, double sABP, double n, Put&& put) {
  const double inv = 1.0 / n;
  const double m20r = sA * inv, m20i = 2.0 * sBh * inv;
  const double m21 = sP * inv;
  const double m40r = sX4 * inv, m40i = 4.0 * sAB * inv;
  const double m41r = sAP * inv, m41i = 2.0 * sBP * inv;
  const double m42 = (2.0 * sAA - sX4) * inv;                         // mean P^2 = A^2 + 4 Bh^2
  const double m60r = (sAAA - 12.0 * sABB) * inv;                     // Re (A+iB)^3, B = 2Bh
  const double m60i = (6.0 * sAAB - 8.0 * sBBB) * inv;
  const double m61r = sX4P * inv, m61i = 4.0 * sABP * inv;
  const double m62 = (sAAA + 4.0 * sABB) * inv;                       // mean A*P^2 (real only)
  const double m63 = (2.0 * sAAP - sX4P) * inv;                       // mean P^3
  // m22 = conj(m20), m43 = conj(m41)

  auto cabs = [](double r, double i) { return q_sqrt(r * r + i * i); };
  const double q20r = m20r * m20r - m20i * m20i, q20i = 2.0 * m20r * m20i;   // m20^2
  const double n20 = m20r * m20r + m20i * m20i;                              // |m20|^2

  put(9, 4, cabs(m20r, m20i));                                               // C20
  put(10, 4, __builtin_fabs(m21));                                           // C21
  put(11, 8, cabs(m40r - 3.0 * q20r, m40i - 3.0 * q20i));                    // C40
  put(12, 8, cabs(m41r - 3.0 * m20r * m21, m41i - 3.0 * m20i * m21));        // C41
  put(13, 8, __builtin_fabs(m42 - n20 - 2.0 * m21 * m21));                   // C42
  {  // C60 = m60 - 15 m20 m40 + 3 m20^3
    const double pr = m20r * m40r - m20i * m40i, pi = m20r * m40i + m20i * m40r;
    const double cr = q20r * m20r - q20i * m20i, ci = q20r * m20i + q20i * m20r;
    put(14, 12, cabs(m60r - 15.0 * pr + 3.0 * cr, m60i - 15.0 * pi + 3.0 * ci));
  }
  {  // C61 = m61 - 5 m21 m40 - 10 m20 m41 + 30 m20^2 m21
    const double pr = m20r * m41r - m20i * m41i, pi = m20r * m41i + m20i * m41r;
    put(15, 12, cabs(m61r - 5.0 * m21 * m40r - 10.0 * pr + 30.0 * q20r * m21,
                     m61i - 5.0 * m21 * m40i - 10.0 * pi + 30.0 * q20i * m21));
  }
  {  // C62 = m62 - 6 m20 m42 - 8 m21 m41 - m22 m40 + 6 m20^2 m22 + 24 m21^2 m20
    // m22 m40 = conj(m20) m40 ; m20^2 m22 = m20 |m20|^2
    const double ar = m20r * m40r + m20i * m40i, ai = m20r * m40i - m20i * m40r;
    const double re = m62 - 6.0 * m20r * m42 - 8.0 * m21 * m41r - ar + 6.0 * m20r * n20 +
                      24.0 * m21 * m21 * m20r;
    const double im = -6.0 * m20i * m42 - 8.0 * m21 * m41i - ai + 6.0 * m20i * n20 +
                      24.0 * m21 * m21 * m20i;
    put(16, 12, cabs(re, im));
  }
  {  // C63 = m63 - 9 m21 m42 + 12 m21^3 - 3 m20 m43 - 3 m22 m41 + 18 m20 m21 m22
    // m20 conj(m41) + conj(m20) m41 = 2 Re(m20 conj(m41)), real
    const double cross = 2.0 * (m20r * m41r + m20i * m41i);
    put(17, 12, __builtin_fabs(m63 - 9.0 * m21 * m42 + 12.0 * m21 * m21 * m21 -
                               3.0 * cross + 18.0 * m21 * n20));
  }
}

}  // namespace amcx
