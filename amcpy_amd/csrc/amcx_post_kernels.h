// The two consumers right behind the extraction path,
// as device-side reductions over the (frames x 18) float32 feature matrix so that the
// matrix need not leave HBM:
//   * per-group column mean / population std: the reference's per-SNR statistics
//     (graphics.py:50-62, np.mean / np.std over frames) and the fit of its
//     StandardScaler (preprocessing.py:59-61);
//   * column select + (x - mean) / scale: the transform (preprocessing.py:55,62).
// Both are tiny next to the feature kernel (46 MB per 638 976 frames) and HBM-bound
// by construction; sums are fp64.
#pragma once

#include "amcx_block_kernel.h"

namespace amcx {

constexpr int kStatMaxCols = 32;

// one workgroup per group; thread t: column t % 32, row lane t / 32 (8 row lanes)
__global__ __launch_bounds__(kBlockThreads) void amcx_group_stats_kernel(
    const float* __restrict__ x, long long rows_per_group, long long row_stride, int n_cols,
    double* __restrict__ mean_out, double* __restrict__ std_out) {
  __shared__ double red[8][kStatMaxCols];
  __shared__ double mean_s[kStatMaxCols];
  const int col = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const float* base = x + (long long)blockIdx.x * rows_per_group * row_stride;
  const bool on = col < n_cols;
  double s = 0;
  if (on)
    for (long long r = rl; r < rows_per_group; r += 8) s += (double)base[r * row_stride + col];
  red[rl][col] = s;
  __syncthreads();
  if (rl == 0) {
    double t = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][col];
    mean_s[col] = t / (double)rows_per_group;
  }
  __syncthreads();
  const double mu = mean_s[col];
  double v = 0;
  if (on)
    for (long long r = rl; r < rows_per_group; r += 8) {
      const double d = (double)base[r * row_stride + col] - mu;
      v += d * d;
    }
  __syncthreads();
  red[rl][col] = v;
  __syncthreads();
  if (rl == 0 && on) {
    double t = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][col];
    mean_out[(long long)blockIdx.x * n_cols + col] = mu;
    std_out[(long long)blockIdx.x * n_cols + col] = __builtin_sqrt(t / (double)rows_per_group);
  }
}

// out[r][j] = float(float(x[r][cols[j]] - mean[j]) / scale[j]): the two roundings of
// numpy's in-place float32 `X -= mean_; X /= scale_` (sklearn StandardScaler.transform)
__global__ __launch_bounds__(kBlockThreads) void amcx_select_scale_kernel(
    const float* __restrict__ x, long long n_rows, long long row_stride, const int* __restrict__ cols,
    int n_sel, const double* __restrict__ mean, const double* __restrict__ scale,
    float* __restrict__ out, long long out_stride) {
  const long long total = n_rows * n_sel;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / n_sel;
    const int j = (int)(i - r * n_sel);
    const float c = (float)((double)x[r * row_stride + cols[j]] - mean[j]);
    out[r * out_stride + j] = (float)((double)c / scale[j]);
  }
}

}  // namespace amcx
