// Second, tiny launch behind the wave kernel: frames whose frequency features were
// flagged (f5 stored negated, amcx_math.h kTieBand) get f5 and f9 recomputed with
// the sign of every near-+-pi phase step decided exactly (exact_step).  Typically
// <0.3 % of frames; the scan itself reads 4 bytes per frame.  Keeping this out of
// the throughput kernel keeps that kernel free of a slow path's register pressure
// (an in-kernel re-sweep cost 60-760 spilled VGPRs, profiles/README.md).
#pragma once

#include "amcx_block_kernel.h"

namespace amcx {

constexpr int kFixListCap = 1024;   // flagged frames handled per round per workgroup

__global__ __launch_bounds__(kBlockThreads) void amcx_step_tie_fix_kernel(
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
    if (tid == 0) *count = 0;
    __syncthreads();
    const long long lim = (r1 - base) < kFixListCap ? (r1 - base) : kFixListCap;
    for (long long k = tid; k < lim; k += kBlockThreads) {
      const float v = out[(base + k) * out_stride + 4];
      if (__builtin_signbitf(v) && v == v) {
        const int slot = __hip_atomic_fetch_add(count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        list[slot] = (int)k;
      }
    }
    __syncthreads();
    const int n_flagged = *count;
    for (int q = 0; q < n_flagged; ++q) {
      const long long f = base + list[q];
      const float2* src = iq + f * row_stride;
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

inline hipError_t launch_tie_fix(const float2* iq, int64_t n_frames, int32_t N, int64_t row_stride,
                                 float* out, int64_t out_stride, hipStream_t stream, int cus) {
  const size_t lds = (size_t)8 * (N + (N & 1)) + sizeof(double) * kBlockWaves * kMaxReduce +
                     sizeof(int) * (kFixListCap + 4);
  if (lds > 64 * 1024) {   // N = 8192: 68.5 KiB of dynamic LDS needs the attribute (idempotent, no allocation)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(amcx_step_tie_fix_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  int64_t grid = (int64_t)cus * 16;     // few flagged frames per workgroup: they are handled one at a time
  const int64_t max_grid = (n_frames + 63) / 64;
  if (grid > max_grid) grid = max_grid;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(amcx_step_tie_fix_kernel, dim3((unsigned)grid), dim3(kBlockThreads), lds, stream, iq,
                     (long long)n_frames, (int)N, (long long)row_stride, out, (long long)out_stride);
  return hipGetLastError();
}

}  // namespace amcx
