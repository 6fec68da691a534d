// EXPERIMENT SUPPORT ONLY (compiled with -DAMCX_EXP_PAIR4096): the launch behind the pair kernel, which does not re-run
// out-of-range frames itself.  Every product kernel does -- the wave kernels behind each batch (amcx_wave_kernel.h), the
// N = 8192 quad kernel in a pass at the end of its launch (amcx_quad_kernel.h) -- so the product library has no second
// launch and does not contain this kernel.
// The throughput kernel does its per-sample arithmetic and its sums in fp32; its finaliser marks, in
// band, the frames it cannot finish -- feature 5 (a standard deviation: >= 0 or NaN) is stored
//   -infinity : the frame is outside the range in which fp32 sixth-order sums are trustworthy (mean
//               power outside [kRangeLoPower, kRangeHiPower], or a sum overflowed: |x| >~ 1e5 or
//               <~ 1e-5, a single huge sample, an infinite sample).  Redone by amcx_range_fixup_kernel
//               below with the block kernel's fp64-sum frame routine -- the reference evaluates in complex128
//               (features.py:46-58) and stays finite over the whole complex64 range, overflowing only
//               in its float32 store.
// Frames with a phase step within an fp32 ulp of +-pi (amcx_math.h kTieBand) no longer leave the wave
// kernel flagged: its finaliser recomputes their f5 / f9 itself (wave_exact_frequency, round 3; rounds
// 1-2 had a third launch, amcx_fixup_kernel, scan the result matrix for them).
// The scan reads 4 bytes per frame and normally finds nothing (~10 us per 639 k frames).  A caller that
// reads `out` on another stream BETWEEN the two launches of one amcx_features18_c64 call sees the -inf
// marks (include/amcx.h).
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

inline hipError_t launch_range_fixup(const float2* iq, int64_t n_frames, int32_t N, int64_t row_stride,
                                      float* out, int64_t out_stride, hipStream_t stream, int cus) {
  const size_t tail = sizeof(double) * kBlockWaves * kMaxReduce + sizeof(int) * (kFixListCap + 4);
  const size_t lds_range = (size_t)16 * N + tail;
  // > 64 KiB of dynamic LDS needs the attribute: set once per device to the most any N asks for
  static bool attr_set[64] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(amcx_range_fixup_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 8192 + (int)tail);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) attr_set[dev] = true;   // benign race: idempotent
  }
  const int64_t max_grid = (n_frames + 63) / 64;
  int per_cu = (int)((160 * 1024) / lds_range);
  per_cu = per_cu < 1 ? 1 : per_cu > 2 ? 2 : per_cu;      // launch bound: two workgroups per CU
  int64_t grid = (int64_t)cus * per_cu;
  if (grid > max_grid) grid = max_grid;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(amcx_range_fixup_kernel, dim3((unsigned)grid), dim3(kBlockThreads), lds_range, stream, iq,
                     (long long)n_frames, (int)N, (long long)row_stride, out, (long long)out_stride);
  return hipGetLastError();
}

}  // namespace amcx
