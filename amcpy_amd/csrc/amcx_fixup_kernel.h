// Second, tiny launch behind the wave kernel.  The throughput kernel does its per-sample
// arithmetic and its sums in fp32 and never takes a slow path itself (an in-kernel re-sweep
// cost 60-760 spilled VGPRs, profiles/README.md); instead its finaliser FLAGS, in band, the
// frames it cannot finish exactly -- feature 5 (a standard deviation: >= 0 or NaN) is stored
//   negative and finite : some phase step lay within an fp32 ulp of +-pi (amcx_math.h kTieBand);
//                         f5 and f9 are recomputed here with the sign of every such step decided
//                         exactly (exact_step).  Typically < 0.3 % of frames.
//   -infinity           : the frame is outside the range in which fp32 sixth-order sums are
//                         trustworthy (mean power outside [kRangeLoPower, kRangeHiPower], or a
//                         sum overflowed: |x| >~ 1e5 or <~ 1e-5, a single huge sample, an
//                         infinite sample).  All 18 features are recomputed here by the block
//                         kernel's frame routine, whose sums are fp64 -- the reference evaluates in
//                         complex128 (features.py:46-58) and stays finite over the whole
//                         complex64 range, overflowing only in its float32 store.
// The scan itself reads 4 bytes per frame.  A caller that reads `out` on another stream BETWEEN
// the two launches of one amcx_features18_c64 call sees those flags (include/amcx.h).
#pragma once

#include "amcx_block_kernel.h"

namespace amcx {

constexpr int kFixListCap = 1024;   // flagged frames handled per round per workgroup
constexpr int kFixFullFrame = 1 << 16; // list entry bit: recompute all 18 features (range flag), not only f5 / f9

__global__ __launch_bounds__(kBlockThreads) void amcx_fixup_kernel(
    const float2* __restrict__ iq, long long n_frames, int N, long long row_stride,
    float* __restrict__ out, long long out_stride) {
  extern __shared__ float4 amcx_fix_smem[];
  float2* const xs = reinterpret_cast<float2*>(amcx_fix_smem);       // block_frame: frame / FFT workspace
  float2* const at = xs + N;                                         // block_frame: (|x|, angle) / twiddles
  float* const th = reinterpret_cast<float*>(amcx_fix_smem);         // tie path: angles of the frame ...
  float* const wv = th + N;                                          // ... and wrapped steps (alias xs)
  double* const scratch = reinterpret_cast<double*>(at + N);         // block_sum scratch
  int* const list = reinterpret_cast<int*>(scratch + kBlockWaves * kMaxReduce);
  int* const count = list + kFixListCap;
  const BlockLds L{nullptr, xs, at, scratch, nullptr, nullptr};
  const int tid = threadIdx.x;

  const long long per = (n_frames + gridDim.x - 1) / gridDim.x;
  const long long r0 = (long long)blockIdx.x * per;
  long long r1 = r0 + per;
  if (r1 > n_frames) r1 = n_frames;

  for (long long base = r0; base < r1; base += kFixListCap) {
    if (tid == 0) *count = 0;
    __syncthreads();
    const long long lim = (r1 - base) < kFixListCap ? (r1 - base) : kFixListCap;
    for (long long k = tid; k < lim; k += kBlockThreads) {
      const float v = out[(base + k) * out_stride + 4];
      if (__builtin_signbitf(v) && v == v) {                          // -x, -0 or -inf; never NaN
        const int slot = __hip_atomic_fetch_add(count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        list[slot] = (int)k | (v == -__builtin_inff() ? kFixFullFrame : 0);
      }
    }
    __syncthreads();
    const int n_flagged = *count;
    for (int q = 0; q < n_flagged; ++q) {
      const int entry = list[q];                                      // LDS: uniform over the workgroup
      const long long f = base + (entry & (kFixFullFrame - 1));
      const float2* src = iq + f * row_stride;
      if (entry & kFixFullFrame) {
        block_frame<kBlockPow2>(src, N, 0, L, out + f * out_stride);  // barriers inside; LDS free on return
        continue;
      }
      for (int n = tid; n < N; n += kBlockThreads) {
        const float2 x = src[n];
        const float a = __builtin_amdgcn_sqrtf(__builtin_fmaf(x.x, x.x, __builtin_fmaf(x.y, x.y, kTinyPower)));
        th[n] = fast_angle(x.x, x.y, a);
      }
      __syncthreads();
      double s1[1] = {0};
      for (int n = tid; n + 1 < N; n += kBlockThreads) {
        const float2 p = src[n], r = src[n + 1];
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

inline hipError_t launch_fixup(const float2* iq, int64_t n_frames, int32_t N, int64_t row_stride,
                               float* out, int64_t out_stride, hipStream_t stream, int cus) {
  const size_t lds = (size_t)16 * N + sizeof(double) * kBlockWaves * kMaxReduce + sizeof(int) * (kFixListCap + 4);
  // > 64 KiB of dynamic LDS (N >= 4096) needs the attribute: set once per device to the most any N asks for
  static bool attr_set[64] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    constexpr int kMaxLds = 16 * 8192 + (int)sizeof(double) * kBlockWaves * kMaxReduce + (int)sizeof(int) * (kFixListCap + 4);
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(amcx_fixup_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) attr_set[dev] = true;   // benign race: idempotent
  }
  int64_t grid = (int64_t)cus * 16;     // few flagged frames per workgroup: they are handled one at a time
  const int64_t max_grid = (n_frames + 63) / 64;
  if (grid > max_grid) grid = max_grid;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(amcx_fixup_kernel, dim3((unsigned)grid), dim3(kBlockThreads), lds, stream, iq,
                     (long long)n_frames, (int)N, (long long)row_stride, out, (long long)out_stride);
  return hipGetLastError();
}

}  // namespace amcx
