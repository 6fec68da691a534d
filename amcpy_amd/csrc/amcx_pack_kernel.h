// Layout kernels in front of the feature kernels: whatever order a host container keeps its
// samples in, the feature kernels read frame-major complex64 -- [n_frames][row_stride], a frame's
// samples contiguous (include/amcx.h).
//
// scipy.io.loadmat hands the reference's (n_snr, n_frames, L) variable back Fortran-ordered
// (feature_extraction.py:46-48): the snr axis is the fastest, a frame's samples are
// n_snr * n_frames elements apart, and what IS contiguous is a sample PLANE -- all (snr, frame)
// positions of one sample index.  The upload path therefore moves planes as they lie (long
// contiguous runs for the host threads and for the copy engine) and this kernel does the
// transposition [sample][position] -> [frame][sample] on the device, where it costs 16 bytes of HBM
// traffic per sample (~1 us per MB) instead of a strided gather by host threads.  Source elements
// are complex64, or complex128 rounded to nearest even here (== numpy's astype).
//
// HBM-bound by construction: every byte read once, written once; 64 x 64 tiles through LDS so that
// both the reads (along positions) and the writes (along samples) are 512-byte runs per wave.
#pragma once

#include <hip/hip_runtime.h>

namespace amcx {

constexpr int kPackTile = 64;

__device__ __forceinline__ float2 pack_load(const float2* p) { return *p; }
__device__ __forceinline__ float2 pack_load(const double2* p) {
  const double2 v = *p;
  return make_float2((float)v.x, (float)v.y);
}

// slab [n_planes][plane_stride] of SRC elements: plane r holds sample n0 + r of every position j < P.
// Position j is frame g(j):  inner_snr != 0: j = k * S + s  ->  g = s * K + k   (snr the fastest axis, as loadmat returns it)
//                            inner_snr == 0: j = g
// dst[g * dst_stride + n0 + r] <- slab[r * plane_stride + j]
template <class SRC>
__global__ __launch_bounds__(256) void amcx_pack_planes_kernel(
    const SRC* __restrict__ slab, int n_planes, long long P, long long plane_stride, int S, long long K,
    int inner_snr, float2* __restrict__ dst, long long dst_stride, int n0) {
  __shared__ float2 tile[kPackTile][kPackTile + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long j0 = (long long)blockIdx.x * kPackTile;
  const int r0 = blockIdx.y * kPackTile;
#pragma unroll 4
  for (int r = wave; r < kPackTile; r += 4)
    if (r0 + r < n_planes && j0 + lane < P) tile[r][lane] = pack_load(slab + (long long)(r0 + r) * plane_stride + j0 + lane);
  __syncthreads();
#pragma unroll 4
  for (int c = wave; c < kPackTile; c += 4) {
    const long long j = j0 + c;
    if (j >= P) break;
    long long g = j;
    if (inner_snr) {
      const long long k = j / S;
      g = (j - k * S) * K + k;
    }
    if (r0 + lane < n_planes) dst[g * dst_stride + n0 + r0 + lane] = tile[lane][c];
  }
}

// complex128 -> complex64, row-packed: dst[f][n] = (float2) src[f][n], n < N
__global__ __launch_bounds__(256) void amcx_c128_to_c64_kernel(const double2* __restrict__ src,
                                                              long long n_frames, int N,
                                                              long long src_stride, float2* __restrict__ dst) {
  const long long total = n_frames * N;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long f = i / N;
    const int n = (int)(i - f * N);
    const double2 v = src[f * src_stride + n];
    dst[i] = make_float2((float)v.x, (float)v.y);
  }
}

template <class SRC>
inline hipError_t launch_pack_planes(const SRC* slab, int n_planes, long long P, long long plane_stride, int S,
                                     long long K, int inner_snr, float2* dst, long long dst_stride, int n0,
                                     hipStream_t stream) {
  if (n_planes <= 0 || P <= 0) return hipSuccess;
  const long long gx = (P + kPackTile - 1) / kPackTile, gy = (n_planes + kPackTile - 1) / kPackTile;
  if (gx > 0x7fffffffLL || gy > 65535) return hipErrorInvalidValue;
  hipLaunchKernelGGL(amcx_pack_planes_kernel<SRC>, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, stream, slab,
                     n_planes, P, plane_stride, S, K, inner_snr, dst, dst_stride, n0);
  return hipGetLastError();
}

}  // namespace amcx
