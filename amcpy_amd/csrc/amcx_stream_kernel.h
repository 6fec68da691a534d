// AMCX_VARIANT_BLOCK above 8192 samples: one 1024-thread workgroup per frame, the frame left where it lies.
//
// The reference takes any frame_size (config.py:96 is a free integer, np.fft.fft, features.py:68, any N).  The block
// kernel (amcx_block_kernel.h) stages a frame in LDS and ends at 8192 samples; the throughput kernels above it
// (amcx_group_kernel.h) exist for the powers of two 16384 and 32768.  This kernel serves every OTHER size up to 32768 --
// and those two as the accuracy-first cross-check of the group kernels -- with the block kernel's arithmetic:
//   * a 256 KB frame does not fit LDS, so it is read from global memory four times (maximum, moments, centred passes)
//     -- the re-reads hit L2 / MALL, one frame per CU being 256 x 256 KB in flight at most -- and scaled by the exact
//     power of two 2^-ex AT THE READ (the block kernel scales its staged copy);
//   * the instantaneous phase is kept in LDS (4 N bytes: 128 KB at N = 32768), the envelope is taken again where it is
//     needed; sums in fp64, centred statistics true two-pass, phase steps decided exactly (exact_step);
//   * the spectral term is the DFT by its definition, X_k = sum_n x_n W_N^(k n).  The samples are staged (scaled) in the
//     LDS the phase has left free, 16384 at a time; thread t takes the bins t + 1024 j, four at a time.  The twiddle of
//     a bin runs as a recurrence w <- w W_N^k (W_N^k from an fp64 sincospi, rounded once) that is RE-SEEDED EXACTLY
//     every 16 terms from a two-level table in LDS by the index (k n) mod N, kept incrementally in integers: a rounding
//     of the recurrence lives for fifteen steps.  Products in fp32, every run of 16 folded into fp64 sums.
//     O(N^2): 1.07e9 complex multiply-adds at N = 32768, ~9 VALU instructions each.  It is the fallback that makes the
//     frame-size domain whole, not a throughput path: a Bluestein form would need a 65536-point convolution buffer
//     (512 KB) per frame in global scratch, and the C ABI's device entries allocate nothing.
// LDS: max(4 N, min(8 N, 128 KB)) (phase, then samples) + 2.1 KB of reduction scratch + (64 + 512) twiddles = 137.9 KB
// from N = 16384 on: one workgroup per CU.  Algorithmic HBM bytes per frame: 8 N + 72 (the re-reads are L2 traffic).
#pragma once

#include "amcx_math.h"

namespace amcx {
namespace stream {

constexpr int kThreads = 1024, kWaves = kThreads / 64, kMaxReduce = 16;
constexpr int kScratchBytes = (int)sizeof(double) * (kWaves + 1) * kMaxReduce;      // the waves' partial sums + the totals: 2176
constexpr int kTwLo = 64, kTwHi = 512;                                       // exponents < 64 * 512 = 32768
constexpr int kTwBytes = (kTwLo + kTwHi) * 8;
constexpr int kMaxN = 32768;
constexpr int kRun = 16;                                                     // DFT terms per exact re-seed of the twiddle recurrence / fp64 fold
constexpr int kBins = 4;                                                     // bins a thread carries at a time
constexpr int kChunk = 16384;                                                // samples staged in LDS at a time (128 KB)

// the area that holds the phase (4 N bytes) in passes A-C and the staged samples (8 bytes each, padded to whole runs) afterwards
__host__ __device__ constexpr size_t area_bytes(int N) {
  return N <= kChunk ? (size_t)8 * ((N + kRun - 1) / kRun * kRun) : (size_t)8 * kChunk;
}
__host__ __device__ constexpr size_t lds_bytes(int N) { return area_bytes(N) + kScratchBytes + kTwBytes; }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Sum K per-thread doubles over the workgroup; every thread gets the totals (barriers inside).  The sixteen waves'
// partial sums are added by K threads and handed out through LDS: every thread adding 16 x K values itself is 240 reads
// in flight at 128 registers.
template <int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double* scratch) {
  static_assert(K <= kMaxReduce, "scratch too small");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* const totals = scratch + kWaves * kMaxReduce;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double w = wave_sum(v[k]);
    if (lane == 0) scratch[wave * kMaxReduce + k] = w;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    double t = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) t += scratch[w * kMaxReduce + threadIdx.x];
    totals[threadIdx.x] = t;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = totals[k];
  __syncthreads();
}

// Maximum over the workgroup; a NaN survives (fmaxf drops it, so a flag is carried).
__device__ __forceinline__ float block_max(float v, double* scratch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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
  for (int w = 1; w < kWaves; ++w) { m = __builtin_fmaxf(m, s[w * 2]); b = __builtin_fmaxf(b, s[w * 2 + 1]); }
  __syncthreads();
  return b > 0.f ? __builtin_nanf("") : m;
}

// The fp64 finaliser wants ~200 registers; a 1024-thread workgroup has 128.  As a function of its own (one thread calls it
// once per frame) its allocation does not weigh on the loops around it.
__device__ __attribute__((noinline)) void finalise_frame(const FrameSums& S, int N, float* __restrict__ out_row, int ex) {
  finalize_features<true>(S, N, out_row, ex);
}

__global__ __launch_bounds__(kThreads, 4) void amcx_features18_stream_kernel(
    const float2* __restrict__ iq, long long n_frames, int N, long long row_stride,
    float* __restrict__ out, long long out_stride) {
  extern __shared__ float4 amcx_stream_smem[];
  float* const th = reinterpret_cast<float*>(amcx_stream_smem);                 // instantaneous phase, N floats
  float2* const xs = reinterpret_cast<float2*>(amcx_stream_smem);               // ... and, behind pass C, the staged samples
  double* const scratch = reinterpret_cast<double*>(reinterpret_cast<char*>(amcx_stream_smem) + area_bytes(N));
  float2* const tlo = reinterpret_cast<float2*>(reinterpret_cast<char*>(scratch) + kScratchBytes);
  float2* const thi = tlo + kTwLo;
  const int tid = threadIdx.x;

  // W_N^m = thi[m >> 6] * tlo[m & 63], m < N <= 32768: the exact angle of each entry in fp64, rounded once
  for (int e = tid; e < kTwLo + kTwHi; e += kThreads) {
    const int m = e < kTwLo ? e : (e - kTwLo) * kTwLo;
    double sn = 0.0, cs = 1.0;
    if (m < N) sincospi(2.0 * (double)m / (double)N, &sn, &cs);
    tlo[e] = make_float2((float)cs, (float)(-sn));
  }
  __syncthreads();

  for (long long f = blockIdx.x; f < n_frames; f += gridDim.x) {
    const float2* __restrict__ src = iq + f * row_stride;
    // ---- the frame's scale: 2^-ex, ex the even-rounded exponent of its largest component (amcx_block_kernel.h) ----
    float mx = 0.f;
    for (int n = tid; n < N; n += kThreads) {
      const float2 x = src[n];
      mx = __builtin_fmaxf(mx, __builtin_fmaxf(__builtin_fabsf(x.x), __builtin_fabsf(x.y)));
    }
    mx = block_max(mx, scratch);                         // NaN if the frame holds one (no scaling then)
    int ex = 0;
    if (mx >= 0x1p-125f && mx <= 3.4028235e38f) ex = (((__builtin_bit_cast(int, mx) >> 23) & 0xff) - 127) & ~1;
    const float sc = __builtin_bit_cast(float, (127 - ex) << 23);              // exact; 1 for ordinary data
    auto sample = [&](int n) -> float2 {
      const float2 x = src[n];
      return make_float2(x.x * sc, x.y * sc);
    };

    FrameSums S;
    // ---- pass A: mixed moments, envelope and phase first sums; the phase goes to LDS ----
    {
      double m[15];
#pragma unroll
      for (int k = 0; k < 15; ++k) m[k] = 0;
      double e[3] = {0, 0, 0};                            // sum a, sum theta, sum |theta|
      for (int n = tid; n < N; n += kThreads) {
        const float2 x = sample(n);
        const double re = x.x, im = x.y;
        const double A = re * re - im * im, Bh = re * im, P = re * re + im * im;
        const double AA = A * A, BB = Bh * Bh, AP = A * P;
        m[0] += A; m[1] += Bh; m[2] += P;
        const double X4 = AA - 4.0 * BB;
        m[3] += AA; m[4] += X4; m[5] += A * Bh;
        m[6] += AP; m[7] += Bh * P;
        m[8] += AA * A; m[9] += A * BB; m[10] += AA * Bh; m[11] += BB * Bh;
        m[12] += AA * P; m[13] += X4 * P; m[14] += AP * Bh;
        const float a = __builtin_amdgcn_sqrtf(__builtin_fmaf(x.x, x.x, __builtin_fmaf(x.y, x.y, kTinyPower)));
        const float t = fast_angle(x.x, x.y, a);
        th[n] = t;
        e[0] += a; e[1] += t; e[2] += __builtin_fabsf(t);
      }
      block_sum(m, scratch);
      block_sum(e, scratch);                              // the barrier inside also publishes `th`
      S.sA = m[0]; S.sBh = m[1]; S.sP = m[2]; S.sAA = m[3]; S.sX4 = m[4]; S.sAB = m[5];
      S.sAP = m[6]; S.sBP = m[7]; S.sAAA = m[8]; S.sABB = m[9]; S.sAAB = m[10]; S.sBBB = m[11];
      S.sAAP = m[12]; S.sX4P = m[13]; S.sABP = m[14];
      S.sa = e[0]; S.Kt = e[1] / N; S.Ka = e[2] / N;
    }
    // ---- pass B: centred envelope / phase sums, first sum of the steps ----
    {
      const double mu = S.sa / N;
      double c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int n = tid; n < N; n += kThreads) {
        const float2 x = sample(n);
        const float a = __builtin_amdgcn_sqrtf(__builtin_fmaf(x.x, x.x, __builtin_fmaf(x.y, x.y, kTinyPower)));
        const float t = th[n];
        const double d = (double)a - mu, d2 = d * d;
        c[0] += __builtin_fabs(d); c[1] += d2; c[2] += d2 * d2;
        const double dt = (double)t - S.Kt;
        c[3] += dt; c[4] += dt * dt;
        const double da = __builtin_fabs((double)t) - S.Ka;
        c[6] += da; c[7] += da * da;
        if (n + 1 < N) {
          const float2 y = sample(n + 1);
          c[5] += exact_step(th[n + 1], t, x.x, x.y, y.x, y.y);
        }
      }
      block_sum(c, scratch);
      S.sad1 = c[0]; S.sad2 = c[1]; S.sad4 = c[2]; S.std1 = c[3]; S.std2 = c[4];
      S.Kw = c[5] / (N - 1);
      S.sab1 = c[6]; S.sab2 = c[7];
    }
    // ---- pass C: centred sums of the wrapped phase step ----
    {
      double c[4] = {0, 0, 0, 0};
      for (int n = tid; n + 1 < N; n += kThreads) {
        const float2 x = sample(n), y = sample(n + 1);
        const double d = (double)exact_step(th[n + 1], th[n], x.x, x.y, y.x, y.y) - S.Kw, d2 = d * d;
        c[0] += d; c[1] += d2; c[2] += d2 * d; c[3] += d2 * d2;
      }
      block_sum(c, scratch);                              // trailing barrier: `th` may be overwritten by the next frame
      S.swd1 = c[0]; S.swd2 = c[1]; S.swd3 = c[2]; S.swd4 = c[3];
    }
    // ---- spectral peak: the DFT by its definition, kBins bins per thread at a time ----
    // (a non-finite sample is caught by the finaliser through the power sum: the maximum need not carry NaNs)
    float peak = 0.f;
    const int n_chunks = (N + kChunk - 1) / kChunk;
    for (int k0 = 0; k0 < N; k0 += kThreads * kBins) {
      int kk[kBins], idx[kBins], step[kBins];
      float rr[kBins], ri[kBins];                         // W_N^k: the recurrence's factor
      double ar[kBins], ai[kBins];
#pragma unroll
      for (int j = 0; j < kBins; ++j) {
        const int k = k0 + j * kThreads + tid;
        kk[j] = k < N ? k : 0;                            // a bin past the end computes bin 0 again (and is not looked at)
        step[j] = (int)(((unsigned)kk[j] * (unsigned)kRun) % (unsigned)N);
        double sn, cs;
        sincospi(2.0 * (double)kk[j] / (double)N, &sn, &cs);
        rr[j] = (float)cs; ri[j] = (float)(-sn);
        ar[j] = 0; ai[j] = 0;
      }
      for (int c = 0; c < n_chunks; ++c) {
        const int c0 = c * kChunk;
        const int len = N - c0 < kChunk ? N - c0 : kChunk;
        const int padded = (len + kRun - 1) / kRun * kRun;
        if (n_chunks > 1 || k0 == 0) {                    // a frame that fits is staged once
          __syncthreads();                                // the previous chunk has been read by everyone
          for (int i = tid; i < padded; i += kThreads) xs[i] = i < len ? sample(c0 + i) : make_float2(0.f, 0.f);
          __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < kBins; ++j) idx[j] = (int)(((unsigned)kk[j] * (unsigned)c0) % (unsigned)N);   // (k c0) mod N: < 2^30
        for (int r0 = 0; r0 < padded; r0 += kRun) {
          float wr[kBins], wi[kBins], pr[kBins], pi[kBins];
#pragma unroll
          for (int j = 0; j < kBins; ++j) {               // the exact twiddle at the head of the run
            const float2 a = thi[idx[j] >> 6], b = tlo[idx[j] & 63];
            wr[j] = __builtin_fmaf(a.x, b.x, -(a.y * b.y));
            wi[j] = __builtin_fmaf(a.x, b.y, a.y * b.x);
            pr[j] = 0.f; pi[j] = 0.f;
            idx[j] += step[j];
            if (idx[j] >= N) idx[j] -= N;
          }
#pragma unroll
          for (int t = 0; t < kRun; ++t) {
            const float2 x = xs[r0 + t];                  // the same address in every lane: one broadcast read
#pragma unroll
            for (int j = 0; j < kBins; ++j) {
              pr[j] = __builtin_fmaf(x.x, wr[j], __builtin_fmaf(-x.y, wi[j], pr[j]));
              pi[j] = __builtin_fmaf(x.x, wi[j], __builtin_fmaf(x.y, wr[j], pi[j]));
              if (t + 1 < kRun) {
                const float nr = __builtin_fmaf(wr[j], rr[j], -(wi[j] * ri[j]));
                wi[j] = __builtin_fmaf(wr[j], ri[j], wi[j] * rr[j]);
                wr[j] = nr;
              }
            }
          }
#pragma unroll
          for (int j = 0; j < kBins; ++j) { ar[j] += (double)pr[j]; ai[j] += (double)pi[j]; }
        }
      }
#pragma unroll
      for (int j = 0; j < kBins; ++j)
        if (k0 + j * kThreads + tid < N) peak = __builtin_fmaxf(peak, (float)(ar[j] * ar[j] + ai[j] * ai[j]));
    }
    peak = block_max(peak, scratch);                      // barriers inside
    S.gmax_raw = peak;
    if (tid == 0) finalise_frame(S, N, out + f * out_stride, ex);
  }
}

}  // namespace stream
}  // namespace amcx
