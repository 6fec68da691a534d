// The small launches behind the wave kernel.  The throughput kernel does its per-sample
// arithmetic and its sums in fp32 and never takes a slow path itself (an in-kernel re-sweep
// cost 60-760 spilled VGPRs, profiles/README.md); instead its finaliser FLAGS, in band, the
// frames it cannot finish exactly -- feature 5 (a standard deviation: >= 0 or NaN) is stored
//   -infinity           : the frame is outside the range in which fp32 sixth-order sums are
//                         trustworthy (mean power outside [kRangeLoPower, kRangeHiPower], or a
//                         sum overflowed: |x| >~ 1e5 or <~ 1e-5, a single huge sample, an
//                         infinite sample).  Redone by the wave kernel's own range pass on a
//                         power-of-two pre-scaled copy (amcx_range_wave_kernel, N = 1024, 2048, 4096)
//                         or, at the other wave sizes, by amcx_range_fixup_kernel below with the
//                         block kernel's fp64-sum frame routine -- the reference evaluates in
//                         complex128 (features.py:46-58) and stays finite over the whole
//                         complex64 range, overflowing only in its float32 store.
//   negative and finite : some phase step lay within an fp32 ulp of +-pi (amcx_math.h kTieBand);
//                         amcx_fixup_kernel recomputes f5 and f9 with the sign of every such
//                         step decided exactly (exact_step).  Typically < 0.3 % of frames.  It runs
//                         last: a range-pass frame can carry a tie flag too.
// Each scan reads 4 bytes per frame.  The kernels are separate because they want different shapes:
// a flagged tie frame costs ~20 us of latency, so the 0.3 % of them need many small workgroups
// in flight (8 N bytes of LDS each, 7 per CU at N = 2048: 30 us per 639 k frames); the range
// routines need the wave kernel's or the block kernel's footprint and normally find nothing (~10 us).
// One kernel with the larger footprint took 50-60 us.  A caller that reads `out` on another
// stream BETWEEN the launches of one amcx_features18_c64 call sees the flags (include/amcx.h).
#pragma once

#include "amcx_block_kernel.h"

namespace amcx {

constexpr int kFixListCap = 1024;   // flagged frames handled per round per workgroup

// list <- indices k < lim (relative to base) of frames whose f5 satisfies pred; returns the count
template <class Pred>
__device__ __forceinline__ int scan_flags(const float* __restrict__ out, long long out_stride, long long base,
                                          long long lim, int* list, int* count, Pred pred) {
  const int tid = threadIdx.x;
  if (tid == 0) *count = 0;
  __syncthreads();
  for (long long k = tid; k < lim; k += kBlockThreads) {
    const float v = out[(base + k) * out_stride + 4];
    if (pred(v)) {
      const int slot = __hip_atomic_fetch_add(count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      list[slot] = (int)k;
    }
  }
  __syncthreads();
  return *count;
}

__global__ __launch_bounds__(kBlockThreads) void amcx_fixup_kernel(
    const float2* __restrict__ iq, long long n_frames, int N, long long row_stride,
    float* __restrict__ out, long long out_stride) {
  extern __shared__ float4 amcx_fix_smem[];
  float* th = reinterpret_cast<float*>(amcx_fix_smem);              // angles of the frame
  float* wv = th + N;                                                // wrapped steps
  double* scratch = reinterpret_cast<double*>(wv + N + (N & 1));     // block_sum scratch
  int* list = reinterpret_cast<int*>(scratch + kBlockWaves * kMaxReduce);
  int* count = list + kFixListCap;
  const int tid = threadIdx.x;

  const long long per = (n_frames + gridDim.x - 1) / gridDim.x;
  const long long r0 = (long long)blockIdx.x * per;
  long long r1 = r0 + per;
  if (r1 > n_frames) r1 = n_frames;

  for (long long base = r0; base < r1; base += kFixListCap) {
    const long long lim = (r1 - base) < kFixListCap ? (r1 - base) : kFixListCap;
    // negative and finite (or -0): a tie flag; -inf belongs to the range kernel, NaN is a result
    const int n_flagged = scan_flags(out, out_stride, base, lim, list, count, [](float v) {
      return __builtin_signbitf(v) && v == v && v != -__builtin_inff();
    });
    for (int q = 0; q < n_flagged; ++q) {
      const long long f = base + list[q];
      const float2* src = iq + f * row_stride;
      // angles and cross products are scale-free, their fp32 evaluation is not (|x|^2 must stay inside float32): work
      // on the frame times 2^-ex, ex the exponent of its largest component -- exact, and a no-op for ordinary data
      float m = 0.f;
      for (int n = tid; n < N; n += kBlockThreads) {
        const float2 x = src[n];
        m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(x.x), __builtin_fabsf(x.y)));
      }
      m = block_max(m, scratch);                                      // NaN if the frame holds one: no scaling then
      int ex = 0;
      if (m >= 0x1p-125f && m <= 3.4028235e38f) ex = ((__builtin_bit_cast(int, m) >> 23) & 0xff) - 127;
      if (ex > 126) ex = 126;
      const float sc = __builtin_bit_cast(float, (127 - ex) << 23);
      for (int n = tid; n < N; n += kBlockThreads) {
        const float2 x = make_float2(src[n].x * sc, src[n].y * sc);
        const float a = __builtin_amdgcn_sqrtf(__builtin_fmaf(x.x, x.x, __builtin_fmaf(x.y, x.y, kTinyPower)));
        th[n] = fast_angle(x.x, x.y, a);
      }
      __syncthreads();
      double s1[1] = {0};
      for (int n = tid; n + 1 < N; n += kBlockThreads) {
        const float2 p = make_float2(src[n].x * sc, src[n].y * sc), r = make_float2(src[n + 1].x * sc, src[n + 1].y * sc);
        const float w = exact_step(th[n + 1], th[n], p.x, p.y, r.x, r.y);
        wv[n] = w;
        s1[0] += w;
      }
      block_sum(s1, scratch);
      const double Kw = s1[0] / (N - 1);
      double c[4] = {0, 0, 0, 0};
      for (int n = tid; n + 1 < N; n += kBlockThreads) {
        const double d = (double)wv[n] - Kw, d2 = d * d;
        c[0] += d; c[1] += d2; c[2] += d2 * d; c[3] += d2 * d2;
      }
      block_sum(c, scratch);      // trailing barrier: th/wv free for the next frame
      if (tid == 0) {
        float f5, f9;
        frequency_features(Kw, c[0], c[1], c[2], c[3], N, f5, f9);
        out[f * out_stride + 4] = f5;
        out[f * out_stride + 8] = f9;
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(kBlockThreads, 2) void amcx_range_fixup_kernel(
    const float2* __restrict__ iq, long long n_frames, int N, long long row_stride,
    float* __restrict__ out, long long out_stride) {
  extern __shared__ float4 amcx_fix_smem[];
  float2* const xs = reinterpret_cast<float2*>(amcx_fix_smem);       // block_frame: frame / FFT workspace
  float2* const at = xs + N;                                         // block_frame: (|x|, angle) / twiddles
  double* const scratch = reinterpret_cast<double*>(at + N);
  int* const list = reinterpret_cast<int*>(scratch + kBlockWaves * kMaxReduce);
  int* const count = list + kFixListCap;
  const BlockLds L{nullptr, xs, at, scratch, nullptr, nullptr};

  const long long per = (n_frames + gridDim.x - 1) / gridDim.x;
  const long long r0 = (long long)blockIdx.x * per;
  long long r1 = r0 + per;
  if (r1 > n_frames) r1 = n_frames;

  for (long long base = r0; base < r1; base += kFixListCap) {
    const long long lim = (r1 - base) < kFixListCap ? (r1 - base) : kFixListCap;
    const int n_flagged = scan_flags(out, out_stride, base, lim, list, count,
                                     [](float v) { return v == -__builtin_inff(); });
    for (int q = 0; q < n_flagged; ++q) {
      const long long f = base + list[q];
      block_frame<kBlockPow2>(iq + f * row_stride, N, 0, L, out + f * out_stride);   // barriers inside; LDS free on return
    }
    __syncthreads();
  }
}

// block_range: also launch amcx_range_fixup_kernel (frame sizes without a range pass of the wave kernel);
// it goes FIRST, so that a range-flagged frame that also holds a near-pi step gets its tie fix afterwards
inline hipError_t launch_fixup(const float2* iq, int64_t n_frames, int32_t N, int64_t row_stride,
                               float* out, int64_t out_stride, hipStream_t stream, int cus, bool block_range) {
  const size_t tail = sizeof(double) * kBlockWaves * kMaxReduce + sizeof(int) * (kFixListCap + 4);
  const size_t lds_tie = (size_t)8 * (N + (N & 1)) + tail, lds_range = (size_t)16 * N + tail;
  // > 64 KiB of dynamic LDS needs the attribute: set once per device to the most any N asks for
  static bool attr_set[64] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    const int max_tail = (int)tail;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(amcx_fixup_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 8192 + max_tail);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(amcx_range_fixup_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 8192 + max_tail);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) attr_set[dev] = true;   // benign race: idempotent
  }
  const int64_t max_grid = (n_frames + 63) / 64;
  int64_t grid;
  if (block_range) {
    int per_cu = (int)((160 * 1024) / lds_range);
    per_cu = per_cu < 1 ? 1 : per_cu > 2 ? 2 : per_cu;      // launch bound: two workgroups per CU
    grid = (int64_t)cus * per_cu;
    if (grid > max_grid) grid = max_grid;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(amcx_range_fixup_kernel, dim3((unsigned)grid), dim3(kBlockThreads), lds_range, stream, iq,
                       (long long)n_frames, (int)N, (long long)row_stride, out, (long long)out_stride);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  grid = (int64_t)cus * 16;     // few flagged frames per workgroup: they are handled one at a time
  if (grid > max_grid) grid = max_grid;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(amcx_fixup_kernel, dim3((unsigned)grid), dim3(kBlockThreads), lds_tie, stream, iq,
                     (long long)n_frames, (int)N, (long long)row_stride, out, (long long)out_stride);
  return hipGetLastError();
}

}  // namespace amcx
