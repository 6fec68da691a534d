// AMCX_VARIANT_BLOCK: one 256-thread workgroup per frame.
//
// The literal shape BASELINE.json's north_star describes: the complex frame
// is staged into LDS once with coalesced loads, every moment is a
// wavefront-level (shuffle) reduction followed by a 4-wave LDS combine, and
// the spectral term is a shared-memory radix-2 FFT.  It is the accuracy-first,
// any-frame-size kernel: sums accumulate in fp64, centred statistics are true
// two-pass.  The reference's np.fft.fft accepts any N (its own known-answer
// test uses N = 10, features.py:240-255), so a frame size that is not a power
// of two takes Bluestein's chirp-z form of the same DFT,
//   X_k = w_k * sum_n (x_n w_n) conj(w_(k-n)),  w_n = exp(-i pi n^2 / N),
// as a circular convolution of length M = 2^ceil(log2(2N-1)) through the same
// LDS FFT (forward DIF, product with the workgroup's precomputed FFT of the
// chirp, inverse DIT; |w_k| = 1 drops out of the peak), or, for N <= 64 and for
// 4096 < N < 8192 (M would not fit LDS), a direct O(N^2) DFT in fp64.
// The throughput kernel is amcx_wave_kernel.h.
//
// LDS per workgroup: 16*N bytes (frame + (|x|, angle) stash / twiddles), or
// 16*M for Bluestein (chirp spectrum + convolution buffer, which the frame and the
// stash alias), + 1.5 KiB of reduction scratch and two-level twiddle tables.
// Algorithmic HBM bytes per frame: 8*N + 72.
#pragma once

#include "amcx_math.h"

namespace amcx {

constexpr int kBlockThreads = 256;
constexpr int kBlockMaxN = 8192;                 // a staged frame + its (|x|, angle) stash: 16 N bytes of LDS; larger: amcx_stream_kernel.h
constexpr int kBlockWaves = kBlockThreads / 64;
constexpr int kMaxReduce = 16;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Sum K per-thread doubles over the workgroup; every thread gets the totals.
template <int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double* scratch) {
  static_assert(K <= kMaxReduce, "scratch too small");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double w = wave_sum(v[k]);
    if (lane == 0) scratch[wave * kMaxReduce + k] = w;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double t = 0;
#pragma unroll
    for (int w = 0; w < kBlockWaves; ++w) t += scratch[w * kMaxReduce + k];
    v[k] = t;
  }
  __syncthreads();
}

__device__ __forceinline__ float block_max(float v, double* scratch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // NaN must survive the reduction: fmaxf drops it, so carry a flag
  float bad = (v == v) ? 0.f : 1.f;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    v = __builtin_fmaxf(v, __shfl_xor(v, off, 64));
    bad = __builtin_fmaxf(bad, __shfl_xor(bad, off, 64));
  }
  float* s = reinterpret_cast<float*>(scratch);
  if (lane == 0) { s[wave * 2] = v; s[wave * 2 + 1] = bad; }
  __syncthreads();
  float m = s[0], b = s[1];
#pragma unroll
  for (int w = 1; w < kBlockWaves; ++w) { m = __builtin_fmaxf(m, s[w * 2]); b = __builtin_fmaxf(b, s[w * 2 + 1]); }
  __syncthreads();
  return b > 0.f ? __builtin_nanf("") : m;
}

constexpr int kBlockDirect = 0, kBlockPow2 = 1, kBlockBluestein = 2, kBlockBluesteinBig = 3;   // spectral-term variants
constexpr int kBluesteinMinN = 65, kBluesteinMaxN = 4096;
// 4096 < N < 8192, not a power of two: the convolution length is 16384 and two LDS arrays of it would be 256 KB.
// kBlockBluesteinBig keeps the convolution buffer in LDS (128 KB, one workgroup per CU) and the chirp's spectrum in
// REGISTERS: thread t multiplies entries t + 256 j, j < 64, and holds exactly those 64 complex values (128 VGPRs)
// for the kernel's lifetime.
constexpr int kBluesteinBigM = 16384, kBluesteinBigPerThread = kBluesteinBigM / kBlockThreads;
constexpr int kBlockScratchBytes = (int)sizeof(double) * kBlockWaves * kMaxReduce;   // 512
constexpr int kBlockTwiddleBytes = (64 + 128) * 8;                                    // T_lo[64], T_hi[128]: exponents < 8192

__host__ __device__ inline int bluestein_length(int n) {          // smallest power of two >= 2n - 1
  int m = 1;
  while (m < 2 * n - 1) m <<= 1;
  return m;
}

// W_M^m = exp(-2 pi i m / M), m < M/2, from two small tables: T_hi[m >> 6] * T_lo[m & 63] (T_hi: up to 128 entries)
__device__ __forceinline__ float2 twiddle2(const float2* tlo, const float2* thi, int m) {
  const float2 a = thi[m >> 6], b = tlo[m & 63];
  return make_float2(__builtin_fmaf(a.x, b.x, -(a.y * b.y)), __builtin_fmaf(a.x, b.y, a.y * b.x));
}

// In-place M-point FFTs over an LDS array by the whole workgroup (trailing barrier included).
// Forward: decimation in frequency, natural order in, bit-reversed order out.
__device__ __forceinline__ void lds_fft_dif(float2* v, int M, const float2* tlo, const float2* thi) {
  const int half_m = M >> 1;
  for (int half = half_m, step = 1; half >= 1; half >>= 1, step <<= 1) {
    for (int b = threadIdx.x; b < half_m; b += kBlockThreads) {
      const int j = b & (half - 1);
      const int i0 = ((b - j) << 1) + j, i1 = i0 + half;
      const float2 p = v[i0], q = v[i1];
      const float2 w = twiddle2(tlo, thi, j * step);
      const float dr = p.x - q.x, di = p.y - q.y;
      v[i0] = make_float2(p.x + q.x, p.y + q.y);
      v[i1] = make_float2(__builtin_fmaf(dr, w.x, -(di * w.y)), __builtin_fmaf(dr, w.y, di * w.x));
    }
    __syncthreads();
  }
}
// Inverse (unscaled): decimation in time with conjugate twiddles, bit-reversed in, natural out.
__device__ __forceinline__ void lds_ifft_dit(float2* v, int M, const float2* tlo, const float2* thi) {
  const int half_m = M >> 1;
  for (int half = 1, step = half_m; half <= half_m; half <<= 1, step >>= 1) {
    for (int b = threadIdx.x; b < half_m; b += kBlockThreads) {
      const int j = b & (half - 1);
      const int i0 = ((b - j) << 1) + j, i1 = i0 + half;
      const float2 p = v[i0], q = v[i1];
      const float2 w = twiddle2(tlo, thi, j * step);            // conj(w) = (w.x, -w.y)
      const float tr = __builtin_fmaf(q.x, w.x, q.y * w.y), ti = __builtin_fmaf(q.y, w.x, -(q.x * w.y));
      v[i0] = make_float2(p.x + tr, p.y + ti);
      v[i1] = make_float2(p.x - tr, p.y - ti);
    }
    __syncthreads();
  }
}

// exp(sign * i pi n^2 / N): n^2 mod 2N in integers (n < 4096), then one sincospi
__device__ __forceinline__ float2 chirp(int n, int N, float sign) {
  const int r = (n * n) % (2 * N);
  float sn, cs;
  sincospif((float)r / (float)N, &sn, &cs);
  return make_float2(cs, sign * sn);
}

// LDS regions of one workgroup (see the kernel below for how they are laid out)
struct BlockLds {
  float2* bh;        // Bluestein: FFT of the chirp, M entries
  float2* xs;        // frame, later FFT workspace (M entries for Bluestein)
  float2* at;        // (|x|, angle), later twiddles / workspace
  double* scratch;   // kBlockWaves * kMaxReduce doubles
  float2* tlo;       // two-level twiddle tables (Bluestein only)
  float2* thi;
};

// All 18 features of ONE frame by the whole 256-thread workgroup; out_row gets the 18 floats.
// Also the slow path behind the wave kernel for frames outside its fp32 range (amcx_fixup_kernel.h).
template <int MODE>
__device__ __forceinline__ void block_frame(const float2* __restrict__ src, int N, int M, const BlockLds& L,
                                            float* __restrict__ out_row, const float2* bh_regs = nullptr) {
  float2* const bh = L.bh;
  float2* const xs = L.xs;
  float2* const at = L.at;
  double* const scratch = L.scratch;
  const float2* const tlo = L.tlo;
  const float2* const thi = L.thi;
  const int tid = threadIdx.x;
  (void)bh; (void)tlo; (void)thi; (void)M;
  {
    // The frame is staged times 2^-ex, ex the even-rounded exponent of its largest component (exact; 1 for
    // ordinary data up to a factor < 4), and the finaliser un-scales in fp64 through the features' scaling
    // laws: every fp32 intermediate (|x|^2, the spectrum) then stays inside float32 for any normal float32
    // input, as the reference's complex128 evaluation does (features.py:46-58).
    float mx = 0.f;
    for (int n = tid; n < N; n += kBlockThreads) {
      const float2 x = src[n];
      xs[n] = x;
      mx = __builtin_fmaxf(mx, __builtin_fmaxf(__builtin_fabsf(x.x), __builtin_fabsf(x.y)));
    }
    mx = block_max(mx, scratch);                       // NaN if the frame holds one (no scaling then); barriers inside
    int ex = 0;
    if (mx >= 0x1p-125f && mx <= 3.4028235e38f) ex = (((__builtin_bit_cast(int, mx) >> 23) & 0xff) - 127) & ~1;
    if (ex != 0) {
      const float sc = __builtin_bit_cast(float, (127 - ex) << 23);
      for (int n = tid; n < N; n += kBlockThreads) xs[n] = make_float2(xs[n].x * sc, xs[n].y * sc);
      __syncthreads();
    }

    FrameSums S;
    // N <= 64 (kBlockDirect -- the size class of the reference's own known-answer vector, N = 10, features.py:240-255):
    // envelope, angle and wrapped step in fp64 (atan2, IEEE sqrt and division), as the reference computes them
    // (features.py:27-30).  A std or kurtosis over three or nine values shows every rounding of an fp32 angle (2e-7 rad):
    // until round 6 these sizes were held to 2e-4 instead of 1e-5.  At most one sample per thread, so the cost is nil.
    [[maybe_unused]] auto env_phase_d = [&](int n, double& a, double& th) {
      const double re = xs[n].x, im = xs[n].y;
      a = __builtin_sqrt(re * re + im * im);
      th = atan2(im, re);
    };
    [[maybe_unused]] auto step_d = [&](int n) -> double {             // diff(unwrap(theta))[n]: d - 2 pi rint(d / 2 pi), half to even
      double a0, t0, a1, t1;
      env_phase_d(n, a0, t0);
      env_phase_d(n + 1, a1, t1);
      const double d = t1 - t0;
      return d - kTwoPiD * __builtin_rint(d / kTwoPiD);
    };
    // ---- pass A: mixed moments, envelope and phase first sums ------------
    {
      double m[15];
#pragma unroll
      for (int k = 0; k < 15; ++k) m[k] = 0;
      double e[3] = {0, 0, 0};  // sum a, sum theta, sum |theta|
      for (int n = tid; n < N; n += kBlockThreads) {
        const float2 x = xs[n];
        const double re = x.x, im = x.y;
        const double A = re * re - im * im, Bh = re * im, P = re * re + im * im;
        const double AA = A * A, BB = Bh * Bh, AP = A * P;
        m[0] += A; m[1] += Bh; m[2] += P;
        const double X4 = AA - 4.0 * BB;
        m[3] += AA; m[4] += X4; m[5] += A * Bh;
        m[6] += AP; m[7] += Bh * P;
        m[8] += AA * A; m[9] += A * BB; m[10] += AA * Bh; m[11] += BB * Bh;
        m[12] += AA * P; m[13] += X4 * P; m[14] += AP * Bh;
        if constexpr (MODE == kBlockDirect) {
          double a, th;
          env_phase_d(n, a, th);
          e[0] += a; e[1] += th; e[2] += __builtin_fabs(th);
        } else {
          const float a = __builtin_amdgcn_sqrtf(__builtin_fmaf(x.x, x.x, __builtin_fmaf(x.y, x.y, kTinyPower)));
          const float th = fast_angle(x.x, x.y, a);
          at[n] = make_float2(a, th);
          e[0] += a; e[1] += th; e[2] += __builtin_fabsf(th);
        }
      }
      block_sum(m, scratch);
      block_sum(e, scratch);   // the barrier inside also publishes `at`
      S.sA = m[0]; S.sBh = m[1]; S.sP = m[2]; S.sAA = m[3]; S.sX4 = m[4]; S.sAB = m[5];
      S.sAP = m[6]; S.sBP = m[7]; S.sAAA = m[8]; S.sABB = m[9]; S.sAAB = m[10]; S.sBBB = m[11];
      S.sAAP = m[12]; S.sX4P = m[13]; S.sABP = m[14];
      S.sa = e[0]; S.Kt = e[1] / N; S.Ka = e[2] / N;
    }
    // ---- pass B: centred envelope / phase sums, first sum of the steps ---
    {
      const double mu = S.sa / N;
      double c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int n = tid; n < N; n += kBlockThreads) {
        double va, vth;
        if constexpr (MODE == kBlockDirect) {
          env_phase_d(n, va, vth);
        } else {
          const float2 v = at[n];
          va = v.x; vth = v.y;
        }
        const double d = va - mu, d2 = d * d;
        c[0] += __builtin_fabs(d); c[1] += d2; c[2] += d2 * d2;
        const double dt = vth - S.Kt;
        c[3] += dt; c[4] += dt * dt;
        const double da = __builtin_fabs(vth) - S.Ka;
        c[6] += da; c[7] += da * da;
        if (n + 1 < N) {
          if constexpr (MODE == kBlockDirect) c[5] += step_d(n);
          else c[5] += exact_step(at[n + 1].y, (float)vth, xs[n].x, xs[n].y, xs[n + 1].x, xs[n + 1].y);
        }
      }
      block_sum(c, scratch);
      S.sad1 = c[0]; S.sad2 = c[1]; S.sad4 = c[2]; S.std1 = c[3]; S.std2 = c[4];
      S.Kw = c[5] / (N - 1);
      S.sab1 = c[6]; S.sab2 = c[7];
    }
    // ---- pass C: centred sums of the wrapped phase step ------------------
    {
      double c[4] = {0, 0, 0, 0};
      for (int n = tid; n + 1 < N; n += kBlockThreads) {
        double w;
        if constexpr (MODE == kBlockDirect) w = step_d(n);
        else w = (double)exact_step(at[n + 1].y, at[n].y, xs[n].x, xs[n].y, xs[n + 1].x, xs[n + 1].y);
        const double d = w - S.Kw, d2 = d * d;
        c[0] += d; c[1] += d2; c[2] += d2 * d; c[3] += d2 * d2;
      }
      block_sum(c, scratch);   // trailing barrier: `at` may now be overwritten
      S.swd1 = c[0]; S.swd2 = c[1]; S.swd3 = c[2]; S.swd4 = c[3];
    }
    // ---- spectral peak ---------------------------------------------------
    // (a non-finite sample is caught by the finaliser through the power sum,
    //  so the maximum itself need not carry NaNs)
    float peak = 0.f;
    if constexpr (MODE == kBlockBluestein || MODE == kBlockBluesteinBig) {
      // a_n = x_n w_n in place over the frame, zero padding over the dead (|x|, angle) stash
      for (int n = tid; n < M; n += kBlockThreads) {
        float2 a = make_float2(0.f, 0.f);
        if (n < N) {
          const float2 x = xs[n], w = chirp(n, N, -1.0f);
          a = make_float2(__builtin_fmaf(x.x, w.x, -(x.y * w.y)), __builtin_fmaf(x.x, w.y, x.y * w.x));
        }
        xs[n] = a;
      }
      __syncthreads();
      lds_fft_dif(xs, M, tlo, thi);
      if constexpr (MODE == kBlockBluesteinBig) {
#pragma unroll
        for (int j = 0; j < kBluesteinBigPerThread; ++j) {        // the chirp's spectrum: this thread's 64 entries, in registers
          const int n = tid + j * kBlockThreads;
          const float2 a = xs[n], b = bh_regs[j];
          xs[n] = make_float2(__builtin_fmaf(a.x, b.x, -(a.y * b.y)), __builtin_fmaf(a.x, b.y, a.y * b.x));
        }
      } else {
        for (int n = tid; n < M; n += kBlockThreads) {            // both spectra are in bit-reversed order
          const float2 a = xs[n], b = bh[n];
          xs[n] = make_float2(__builtin_fmaf(a.x, b.x, -(a.y * b.y)), __builtin_fmaf(a.x, b.y, a.y * b.x));
        }
      }
      __syncthreads();
      lds_ifft_dit(xs, M, tlo, thi);
      const float inv_m2 = 1.0f / ((float)M * (float)M);          // the inverse transform is unscaled
      for (int n = tid; n < N; n += kBlockThreads) {
        const float2 X = xs[n];
        peak = __builtin_fmaxf(peak, __builtin_fmaf(X.x, X.x, X.y * X.y) * inv_m2);
      }
    } else if constexpr (MODE == kBlockPow2) {
      // in-place radix-2 decimation-in-frequency; output order is bit-reversed,
      // which a maximum does not care about.  Twiddles W_N^m, m < N/2, are tabulated
      // once per frame in the (|x|, angle) stash, which is dead by now: N/512
      // sincospif per thread instead of one per butterfly.
      const int half_n = N >> 1;
      for (int mI = tid; mI < half_n; mI += kBlockThreads) {
        float sn, cs;
        sincospif((float)mI / (float)half_n, &sn, &cs);        // W_N^m = cs - i*sn
        at[mI] = make_float2(cs, sn);
      }
      __syncthreads();
      for (int half = half_n, step = 1; half >= 1; half >>= 1, step <<= 1) {
        for (int b = tid; b < half_n; b += kBlockThreads) {
          const int j = b & (half - 1);
          const int i0 = ((b - j) << 1) + j, i1 = i0 + half;
          const float2 u = xs[i0], v = xs[i1];
          const float2 w = at[j * step];                       // W_(2 half)^j = W_N^(j N/(2 half))
          const float cs = w.x, sn = w.y;
          const float dr = u.x - v.x, di = u.y - v.y;
          xs[i0] = make_float2(u.x + v.x, u.y + v.y);
          xs[i1] = make_float2(__builtin_fmaf(dr, cs, di * sn), __builtin_fmaf(di, cs, -dr * sn));
        }
        __syncthreads();
      }
      for (int n = tid; n < N; n += kBlockThreads) {
        const float2 X = xs[n];
        peak = __builtin_fmaxf(peak, __builtin_fmaf(X.x, X.x, X.y * X.y));
      }
    } else {
      // direct DFT with an exact-index twiddle table: tw[m] = exp(-2*pi*i*m/N)
      for (int mI = tid; mI < N; mI += kBlockThreads) {
        double sn, cs;
        sincospi(2.0 * (double)mI / (double)N, &sn, &cs);
        at[mI] = make_float2((float)cs, (float)(-sn));
      }
      __syncthreads();
      for (int k = tid; k < N; k += kBlockThreads) {
        double ar = 0, ai = 0;
        int idx = 0;
        for (int n = 0; n < N; ++n) {
          const float2 x = xs[n], w = at[idx];
          ar += (double)x.x * w.x - (double)x.y * w.y;
          ai += (double)x.x * w.y + (double)x.y * w.x;
          idx += k;
          if (idx >= N) idx -= N;
        }
        peak = __builtin_fmaxf(peak, (float)(ar * ar + ai * ai));
      }
    }
    peak = block_max(peak, scratch);   // barriers inside: LDS free for the next frame
    S.gmax_raw = peak;
    if (tid == 0) finalize_features<true>(S, N, out_row, ex);
  }
}

template <int MODE>
__global__ __launch_bounds__(kBlockThreads, MODE == kBlockBluesteinBig ? 1 : 2) void amcx_features18_block_kernel(
    const float2* __restrict__ iq, long long n_frames, int N, long long row_stride,
    float* __restrict__ out, long long out_stride) {
  extern __shared__ float4 amcx_block_smem[];
  float2* const base = reinterpret_cast<float2*>(amcx_block_smem);
  constexpr bool kChirp = MODE == kBlockBluestein || MODE == kBlockBluesteinBig;
  const int M = MODE == kBlockBluestein ? bluestein_length(N) : MODE == kBlockBluesteinBig ? kBluesteinBigM : 0;
  float2* bh = base;                                              // Bluestein: FFT of the chirp, M entries (Big: in registers)
  float2* xs = MODE == kBlockBluestein ? base + M : base;         // frame, later FFT workspace (M entries for Bluestein)
  float2* at = xs + N;                                            // (|x|, angle), later twiddles / workspace
  double* scratch = reinterpret_cast<double*>(kChirp ? xs + M : at + N);   // kBlockWaves * kMaxReduce doubles
  float2* tlo = reinterpret_cast<float2*>(reinterpret_cast<char*>(scratch) + kBlockScratchBytes);
  float2* thi = tlo + 64;
  const int tid = threadIdx.x;
  [[maybe_unused]] float2 bh_regs[MODE == kBlockBluesteinBig ? kBluesteinBigPerThread : 1];

  if constexpr (kChirp) {
    // once per workgroup: two-level twiddles of the M-point FFT and the chirp's spectrum
    if (tid < 192) {
      const int k = tid < 64 ? tid : tid - 64;
      float sn, cs;
      sincospif((float)(tid < 64 ? k : 64 * k) * (2.0f / (float)M), &sn, &cs);
      tlo[tid] = make_float2(cs, -sn);                            // tlo[k] = W_M^k (k < 64), thi[k] = W_M^(64 k) (k < 128)
    }
    float2* const spec = MODE == kBlockBluesteinBig ? xs : bh;    // Big: built in the convolution buffer, then taken into registers
    for (int n = tid; n < M; n += kBlockThreads) {
      const int d = n < N ? n : (M - n < N ? M - n : -1);         // conj(w) is even in n: wrap it around
      spec[n] = d >= 0 ? chirp(d, N, 1.0f) : make_float2(0.f, 0.f);
    }
    __syncthreads();
    lds_fft_dif(spec, M, tlo, thi);
    if constexpr (MODE == kBlockBluesteinBig) {
#pragma unroll
      for (int j = 0; j < kBluesteinBigPerThread; ++j) bh_regs[j] = spec[tid + j * kBlockThreads];
      __syncthreads();                                            // the buffer is the frames' from here on
    }
  }

  BlockLds L{bh, xs, at, scratch, tlo, thi};
  for (long long f = blockIdx.x; f < n_frames; f += gridDim.x)
    block_frame<MODE>(iq + f * row_stride, N, M, L, out + f * out_stride, bh_regs);
}

}  // namespace amcx
