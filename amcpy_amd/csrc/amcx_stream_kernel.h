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
//   * WITH A WORKSPACE (amcx_features18_c64_ws, or the stream-ordered allocator behind amcx_features18_c64[_ex]) the
//     spectral term is an FFT instead (the kernel's <true> form): Bluestein's chirp-z for sizes that are not powers of two,
//     X_k = w_k sum_n (x_n w_n) conj(w_(k-n)), w_n = exp(-i pi n^2 / N), as a circular convolution of length M = 32768 /
//     65536 (the FFT of the chirp once per launch, amcx_stream_chirp_kernel), and the plain FFT for 16384 / 32768.  The
//     M-point transforms run IN PLACE ON GLOBAL MEMORY -- one private M x 8 byte buffer per workgroup, L2 / MALL traffic --
//     in the four-step form M = R x 256: tiles of 32 columns (then 32 rows) go through LDS, 32 radix-2 transforms of R
//     (256) points side by side, the twist W_M^(c kr) from a two-level table; forward decimation in frequency
//     (bit-reversed out), the product with the chirp's spectrum in that order, inverse decimation in time (natural out), the
//     peak taken from the last tile without a store.  O(M log M): ~100 x the rate of the definition at N = 32767.
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

// the FFT form: M = R x kC, tiles of kTile transforms side by side in LDS, kPitch complex elements apart (odd: the
// transposing tile loads are conflict-free)
constexpr int kC = 256, kLogC = 8, kTile = 32, kPitch = 257;
constexpr int kTileBytes = kTile * kPitch * 8;                                // 65 792
constexpr int kW256 = 128, kMLo = 64, kMHi = 1024;                           // W_256^m (m < 128); W_M^e = mhi[e >> 6] * mlo[e & 63], e < 65536
constexpr int kFftTabBytes = (kW256 + kMLo + kMHi) * 8;                       // 9 728

// the area that holds the phase (4 N bytes) in passes A-C and the staged samples (8 bytes each, padded to whole runs) or
// the FFT's tile afterwards
__host__ __device__ constexpr size_t area_bytes(int N) {
  const size_t a = N <= kChunk ? (size_t)8 * ((N + kRun - 1) / kRun * kRun) : (size_t)8 * kChunk;
  return a > (size_t)kTileBytes ? a : (size_t)kTileBytes;
}
__host__ __device__ constexpr size_t lds_bytes(int N) { return area_bytes(N) + kScratchBytes + kTwBytes; }
__host__ __device__ constexpr size_t lds_bytes_fft(int N) { return area_bytes(N) + kScratchBytes + kFftTabBytes; }

// length of the transform the FFT form runs: N itself for a power of two, else Bluestein's M = 2^ceil(log2(2N - 1))
__host__ __device__ inline int conv_length(int N) {
  if ((N & (N - 1)) == 0) return N;
  int m = 1;
  while (m < 2 * N - 1) m <<= 1;
  return m;
}

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

// ---- the FFT form ---------------------------------------------------------------------------------------------------
struct FftLds {
  float2* tile;          // kTile transforms, kPitch apart
  const float2* w256;    // W_256^m, m < 128
  const float2* mlo;     // W_M^l, l < 64
  const float2* mhi;     // W_M^(64 h), h < 1024
};

__device__ __forceinline__ void build_fft_tables(float2* tab, int M) {
  for (int e = threadIdx.x; e < kW256 + kMLo + kMHi; e += kThreads) {
    double turn;                                           // the entry's angle in turns
    if (e < kW256) turn = (double)e / 256.0;
    else if (e < kW256 + kMLo) turn = (double)(e - kW256) / (double)M;
    else turn = (double)((e - kW256 - kMLo) * 64 % M) / (double)M;
    double sn, cs;
    sincospi(2.0 * turn, &sn, &cs);
    tab[e] = make_float2((float)cs, (float)(-sn));
  }
}

__device__ __forceinline__ float2 twiddle_m(const FftLds& L, int e) {
  const float2 a = L.mhi[e >> 6], b = L.mlo[e & 63];
  return make_float2(__builtin_fmaf(a.x, b.x, -(a.y * b.y)), __builtin_fmaf(a.x, b.y, a.y * b.x));
}

// kTile transforms of len = 2^log_len (<= 256) points in the tile, by the whole workgroup; trailing barrier.
// Forward: decimation in frequency, natural order in, bit-reversed out.  Inverse (unscaled): decimation in time with
// conjugate twiddles, bit-reversed in, natural out (the block kernel's lds_fft_dif / lds_ifft_dit, side by side).
template <bool INV>
__device__ __forceinline__ void tile_fft(float2* tile, int log_len, const float2* w256) {
  const int log_half = log_len - 1, half_len = 1 << log_half, total = kTile << log_half;
  for (int s = 0; s <= log_half; ++s) {
    const int lh = INV ? s : log_half - s, half = 1 << lh;
    for (int b = threadIdx.x; b < total; b += kThreads) {
      const int t = b >> log_half, bb = b & (half_len - 1);
      const int j = bb & (half - 1);
      const int i0 = ((bb - j) << 1) + j, i1 = i0 + half;
      float2* v = tile + t * kPitch;
      const float2 p = v[i0], q = v[i1];
      const float2 w = w256[j << (7 - lh)];                // W_(2 half)^j
      if constexpr (!INV) {
        const float dr = p.x - q.x, di = p.y - q.y;
        v[i0] = make_float2(p.x + q.x, p.y + q.y);
        v[i1] = make_float2(__builtin_fmaf(dr, w.x, -(di * w.y)), __builtin_fmaf(dr, w.y, di * w.x));
      } else {
        const float tr = __builtin_fmaf(q.x, w.x, q.y * w.y), ti = __builtin_fmaf(q.y, w.x, -(q.x * w.y));
        v[i0] = make_float2(p.x + tr, p.y + ti);
        v[i1] = make_float2(p.x - tr, p.y - ti);
      }
    }
    __syncthreads();
  }
}

// Forward M-point transform of g (global, private to this workgroup), in place, M = R x 256, R = 2^log_r (64 ... 256):
// element [p][q] (p < R, q < 256, at p * 256 + q) leaves as X[brev_R(p) + R brev_256(q)].
// MODE 0: store; 1: store times the same element of `other` (the chirp's spectrum, same order); 2: no store, the largest
// |X|^2 is returned (a frame size that is a power of two needs nothing else).
template <int MODE>
__device__ __forceinline__ float fft_forward(float2* g, int log_r, const FftLds& L, const float2* __restrict__ other) {
  const int R = 1 << log_r, tid = threadIdx.x;
  float peak = 0.f;
  for (int c0 = 0; c0 < kC; c0 += kTile) {                  // step 1: the columns, R points each, then the twist W_M^(c kr)
    for (int i = tid; i < kTile * R; i += kThreads) {
      const int t = i & (kTile - 1), r = i >> 5;
      L.tile[t * kPitch + r] = g[r * kC + c0 + t];
    }
    __syncthreads();
    tile_fft<false>(L.tile, log_r, L.w256);
    for (int i = tid; i < kTile * R; i += kThreads) {
      const int t = i & (kTile - 1), p = i >> 5;
      const int kr = (int)(__brev((unsigned)p) >> (32 - log_r));
      const float2 w = twiddle_m(L, (c0 + t) * kr), v = L.tile[t * kPitch + p];
      g[p * kC + c0 + t] = make_float2(__builtin_fmaf(v.x, w.x, -(v.y * w.y)), __builtin_fmaf(v.x, w.y, v.y * w.x));
    }
    __syncthreads();
  }
  for (int p0 = 0; p0 < R; p0 += kTile) {                   // step 2: the rows, 256 points each
    for (int i = tid; i < kTile * kC; i += kThreads) {
      const int c = i & (kC - 1), t = i >> kLogC;
      L.tile[t * kPitch + c] = g[(p0 + t) * kC + c];
    }
    __syncthreads();
    tile_fft<false>(L.tile, kLogC, L.w256);
    for (int i = tid; i < kTile * kC; i += kThreads) {
      const int q = i & (kC - 1), t = i >> kLogC;
      float2 v = L.tile[t * kPitch + q];
      const int at = (p0 + t) * kC + q;
      if constexpr (MODE == 2) {
        peak = __builtin_fmaxf(peak, __builtin_fmaf(v.x, v.x, v.y * v.y));
      } else {
        if constexpr (MODE == 1) {
          const float2 b = other[at];
          v = make_float2(__builtin_fmaf(v.x, b.x, -(v.y * b.y)), __builtin_fmaf(v.x, b.y, v.y * b.x));
        }
        g[at] = v;
      }
    }
    __syncthreads();
  }
  return peak;
}

// Inverse (unscaled) of fft_forward's order, in place; returns the largest |y_n|^2 over n < n_out (nothing is stored
// by the last step).
__device__ __forceinline__ float fft_inverse_peak(float2* g, int log_r, int n_out, const FftLds& L) {
  const int R = 1 << log_r, tid = threadIdx.x;
  for (int p0 = 0; p0 < R; p0 += kTile) {                   // the rows back, then the conjugate twist
    for (int i = tid; i < kTile * kC; i += kThreads) {
      const int q = i & (kC - 1), t = i >> kLogC;
      L.tile[t * kPitch + q] = g[(p0 + t) * kC + q];
    }
    __syncthreads();
    tile_fft<true>(L.tile, kLogC, L.w256);
    for (int i = tid; i < kTile * kC; i += kThreads) {
      const int c = i & (kC - 1), t = i >> kLogC;
      const int kr = (int)(__brev((unsigned)(p0 + t)) >> (32 - log_r));
      const float2 w = twiddle_m(L, c * kr), v = L.tile[t * kPitch + c];      // times conj(w)
      g[(p0 + t) * kC + c] = make_float2(__builtin_fmaf(v.x, w.x, v.y * w.y), __builtin_fmaf(v.y, w.x, -(v.x * w.y)));
    }
    __syncthreads();
  }
  float peak = 0.f;
  for (int c0 = 0; c0 < kC; c0 += kTile) {                  // the columns back: natural order, n = r * 256 + c
    for (int i = tid; i < kTile * R; i += kThreads) {
      const int t = i & (kTile - 1), p = i >> 5;
      L.tile[t * kPitch + p] = g[p * kC + c0 + t];
    }
    __syncthreads();
    tile_fft<true>(L.tile, log_r, L.w256);
    for (int i = tid; i < kTile * R; i += kThreads) {
      const int t = i & (kTile - 1), r = i >> 5;
      if (r * kC + c0 + t < n_out) {
        const float2 v = L.tile[t * kPitch + r];
        peak = __builtin_fmaxf(peak, __builtin_fmaf(v.x, v.x, v.y * v.y));
      }
    }
    __syncthreads();
  }
  return peak;
}

// exp(sign * i pi n^2 / N): n^2 mod 2N in integers (n < 32768: n^2 < 2^30), then one sincospi
__device__ __forceinline__ float2 chirp(int n, int N, float sign) {
  const int r = (n * n) % (2 * N);
  float sn, cs;
  sincospif((float)r / (float)N, &sn, &cs);
  return make_float2(cs, sign * sn);
}

// Once per launch, one workgroup: the spectrum of Bluestein's filter conj(w), wrapped around M, in fft_forward's order.
__global__ __launch_bounds__(kThreads, 4) void amcx_stream_chirp_kernel(float2* __restrict__ bspec, int N, int M) {
  extern __shared__ float4 amcx_stream_smem[];
  float2* const tile = reinterpret_cast<float2*>(amcx_stream_smem);
  float2* const tab = tile + kTile * kPitch;
  build_fft_tables(tab, M);
  for (int n = threadIdx.x; n < M; n += kThreads) {
    const int d = n < N ? n : (M - n < N ? M - n : -1);           // conj(w) is even in n
    bspec[n] = d >= 0 ? chirp(d, N, 1.0f) : make_float2(0.f, 0.f);
  }
  __syncthreads();
  const FftLds L{tile, tab, tab + kW256, tab + kW256 + kMLo};
  fft_forward<0>(bspec, 31 - __builtin_clz((unsigned)(M / kC)), L, nullptr);
}

// The fp64 finaliser wants ~200 registers; a 1024-thread workgroup has 128.  As a function of its own (one thread calls it
// once per frame) its allocation does not weigh on the loops around it.
__device__ __attribute__((noinline)) void finalise_frame(const FrameSums& S, int N, float* __restrict__ out_row, int ex) {
  finalize_features<true>(S, N, out_row, ex);
}

// FFT = false: the spectral term by the definition (no workspace).  FFT = true: through `bufs` (gridDim.x buffers of M
// complex values) and, for sizes that are not powers of two, `bspec` (amcx_stream_chirp_kernel's output; else null).
template <bool FFT>
__global__ __launch_bounds__(kThreads, 4) void amcx_features18_stream_kernel(
    const float2* __restrict__ iq, long long n_frames, int N, long long row_stride,
    float* __restrict__ out, long long out_stride, const float2* __restrict__ bspec, float2* __restrict__ bufs, int M) {
  extern __shared__ float4 amcx_stream_smem[];
  float* const th = reinterpret_cast<float*>(amcx_stream_smem);                 // instantaneous phase, N floats
  float2* const xs = reinterpret_cast<float2*>(amcx_stream_smem);               // ... and, behind pass C, the staged samples
  double* const scratch = reinterpret_cast<double*>(reinterpret_cast<char*>(amcx_stream_smem) + area_bytes(N));
  float2* const tlo = reinterpret_cast<float2*>(reinterpret_cast<char*>(scratch) + kScratchBytes);
  float2* const thi = tlo + kTwLo;
  const int tid = threadIdx.x;

  if constexpr (FFT) {
    build_fft_tables(tlo, M);                             // W_256^m, then the two levels of W_M^e
  } else {
    // W_N^m = thi[m >> 6] * tlo[m & 63], m < N <= 32768: the exact angle of each entry in fp64, rounded once
    for (int e = tid; e < kTwLo + kTwHi; e += kThreads) {
      const int m = e < kTwLo ? e : (e - kTwLo) * kTwLo;
      double sn = 0.0, cs = 1.0;
      if (m < N) sincospi(2.0 * (double)m / (double)N, &sn, &cs);
      tlo[e] = make_float2((float)cs, (float)(-sn));
    }
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
    float peak = 0.f;
    if constexpr (FFT) {
      // ---- spectral peak through the workgroup's buffer in global memory (the phase's LDS is the FFT's tile now) ----
      float2* const buf = bufs + (size_t)blockIdx.x * (size_t)M;
      const FftLds L{xs, tlo, tlo + kW256, tlo + kW256 + kMLo};
      const int log_r = 31 - __builtin_clz((unsigned)(M / kC));
      if (bspec != nullptr) {                             // Bluestein: a_n = x_n w_n, zero-padded to M
        for (int n = tid; n < M; n += kThreads) {
          float2 a = make_float2(0.f, 0.f);
          if (n < N) {
            const float2 x = sample(n), w = chirp(n, N, -1.0f);
            a = make_float2(__builtin_fmaf(x.x, w.x, -(x.y * w.y)), __builtin_fmaf(x.x, w.y, x.y * w.x));
          }
          buf[n] = a;
        }
        __syncthreads();
        fft_forward<1>(buf, log_r, L, bspec);
        peak = fft_inverse_peak(buf, log_r, N, L) * (1.0f / ((float)M * (float)M));   // the inverse is unscaled; |w_k| = 1 drops out
      } else {                                            // a power of two: M = N, the transform itself
        for (int n = tid; n < N; n += kThreads) buf[n] = sample(n);
        __syncthreads();
        peak = fft_forward<2>(buf, log_r, L, nullptr);
      }
    } else {
      // ---- spectral peak: the DFT by its definition, kBins bins per thread at a time ----
      // (a non-finite sample is caught by the finaliser through the power sum: the maximum need not carry NaNs)
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
    }
    peak = block_max(peak, scratch);                      // barriers inside
    S.gmax_raw = peak;
    if (tid == 0) finalise_frame(S, N, out + f * out_stride, ex);
  }
}

}  // namespace stream
}  // namespace amcx
