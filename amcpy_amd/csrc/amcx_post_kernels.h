// The two consumers right behind the extraction path,
// as device-side reductions over the (frames x 18) float32 feature matrix so that the
// matrix need not leave HBM:
//   * per-group column mean / population std: the reference's per-SNR statistics
//     (graphics.py:50-62, np.mean / np.std over frames) and the fit of its
//     StandardScaler (preprocessing.py:59-61);
//   * column select + (x - mean) / scale: the transform (preprocessing.py:55,62).
// Both are HBM-bound by construction (46 MB per 638 976 frames, read once).
//
// Statistics: a group's rows are cut into chunks, one workgroup per chunk; a workgroup
// walks its chunk in tiles it holds in registers (16 rows per thread, all loads of a
// tile in flight before the first add).  (n, sum, M2) triples are pooled upwards --
// thread -> tile -> chunk -> group -- each time about the mean of the level above
// (M2 = sum_i [M2_i + n_i (mean_i - mean)^2], Chan, Golub & LeVeque's update): the
// matrix is read ONCE, every deviation is taken about a mean of nearby rows (robust to
// an outlier anywhere, unlike a shifted one-pass sum), arithmetic is fp64, and the mean
// is the plain fp64 sum over n -- so an inf in a column gives mean +-inf / std nan
// exactly as numpy's two passes do.  The group level is a second, tiny launch.
#pragma once

#include "amcx_block_kernel.h"

namespace amcx {

constexpr int kStatMaxCols = 32;
constexpr int kStatThreads = 256;
constexpr int kStatPerThread = 16;     // rows of a tile held by one thread
constexpr int kStatMaxRowLanes = 32;   // row lanes of a tile (threads per column)

// row lanes / rows of one tile for a column count (host and device agree through this)
__host__ __device__ inline int stat_row_lanes(int n_cols) {
  const int l = kStatThreads / n_cols;
  return l < kStatMaxRowLanes ? l : kStatMaxRowLanes;
}
__host__ __device__ inline int stat_tile_rows(int n_cols) { return stat_row_lanes(n_cols) * kStatPerThread; }

// grid n_groups * chunks_per_group (chunk fastest); thread t: column t % n_cols, row lane t / n_cols,
// so the 252 active threads of an 18-column tile read 252 consecutive floats per load.
// A thread reduces the 16 rows it holds of a tile to (n, sum, M2 about THEIR mean) in
// registers and folds that into its running triple (pairwise update) while the next
// tile's loads are in flight; one barrier at the end of the chunk, then the row-lane-0
// thread of each column pools the <= 32 thread triples about the chunk mean.
// part: [group][col][3][chunk] doubles = (n, sum, M2) -- chunk fastest, so the pooling
// launch reads it coalesced.
__global__ __launch_bounds__(kStatThreads) void amcx_stats_part_kernel(
    const float* __restrict__ x, long long rows_per_group, long long row_stride, int n_cols,
    long long rows_per_chunk, int chunks_per_group, double* __restrict__ part) {
  __shared__ double ln[kStatMaxRowLanes][kStatMaxCols + 1], ls[kStatMaxRowLanes][kStatMaxCols + 1],
      lm[kStatMaxRowLanes][kStatMaxCols + 1], lq[kStatMaxRowLanes][kStatMaxCols + 1];
  const int lanes = stat_row_lanes(n_cols);
  const int rl = (int)threadIdx.x / n_cols, col = (int)threadIdx.x - rl * n_cols;
  const bool on = rl < lanes;
  const long long group = blockIdx.x / (unsigned)chunks_per_group;
  const int chunk = (int)(blockIdx.x - group * chunks_per_group);
  const long long row0 = (long long)chunk * rows_per_chunk;
  const long long row_end = row0 + rows_per_chunk < rows_per_group ? row0 + rows_per_chunk : rows_per_group;
  const float* xg = x + group * rows_per_group * row_stride;      // uniform
  const int tile_rows = lanes * kStatPerThread;
  // the thread's 16 element offsets inside a tile do not depend on the tile: 32-bit, formed once
  // (the host refuses row strides for which tile_rows * row_stride would not fit)
  unsigned off[kStatPerThread];
#pragma unroll
  for (int k = 0; k < kStatPerThread; ++k)
    off[k] = on ? (unsigned)(rl + k * lanes) * (unsigned)row_stride + (unsigned)col : 0u;
  // loads are unconditional (a full tile: the offsets above; the ragged last one: rows clamped to the last valid
  // row, values masked afterwards), so that a tile is 16 straight-line loads off one uniform base
  auto load_tile = [&](long long t0, float (&v)[kStatPerThread]) {
    const long long left = row_end - t0;
    if (left <= 0) {                       // uniform: past the chunk, nothing is read
#pragma unroll
      for (int k = 0; k < kStatPerThread; ++k) v[k] = 0.0f;
      return;
    }
    const float* tb = xg + t0 * row_stride;
    if (left >= tile_rows) {
#pragma unroll
      for (int k = 0; k < kStatPerThread; ++k) v[k] = tb[off[k]];
    } else {
      const unsigned last = (unsigned)left - 1u;
#pragma unroll
      for (int k = 0; k < kStatPerThread; ++k) {
        const unsigned r = (unsigned)(rl + k * lanes);
        const float f = tb[on ? (r < last ? r : last) * (unsigned)row_stride + (unsigned)col : 0u];
        v[k] = r <= last ? f : 0.0f;
      }
    }
  };
  double run_n = 0, run_s = 0, run_m = 0, run_q = 0;   // the sum is kept beside the mean: the mean out is sum / n
  float v[kStatPerThread], w[kStatPerThread];
  load_tile(row0, v);
  for (long long t0 = row0; t0 < row_end; t0 += tile_rows) {
    load_tile(t0 + tile_rows, w);
    const long long left = row_end - t0;
    const int tile_n = left < tile_rows ? (int)left : tile_rows;
    // rows of this tile the thread holds: r = rl + k * lanes < tile_n  <=>  k < mine
    const int mine = on && rl < tile_n ? (tile_n - rl + lanes - 1) / lanes : 0;
    if (mine) {
      double s = 0;
#pragma unroll
      for (int k = 0; k < kStatPerThread; ++k) s += k < mine ? (double)v[k] : 0.0;
      const double nb = (double)mine, m = s / nb;
      double q = 0;
#pragma unroll
      for (int k = 0; k < kStatPerThread; ++k) {
        const double d = (double)v[k] - m;
        if (k < mine) q += d * d;
      }
      const double d = m - run_m, n = run_n + nb;      // run_n == 0: d * d * 0, run_m <- m
      run_q += q + d * d * (run_n * nb / n);
      run_s += s;
      run_m = run_s / n;
      run_n = n;
    }
#pragma unroll
    for (int k = 0; k < kStatPerThread; ++k) v[k] = w[k];
  }
  if (on) {
    ln[rl][col] = run_n; ls[rl][col] = run_s; lm[rl][col] = run_m; lq[rl][col] = run_q;
  }
  __syncthreads();
  if (rl == 0) {
    double tn = 0, ts = 0;
    for (int k = 0; k < lanes; ++k) {
      tn += ln[k][col];
      ts += ls[k][col];
    }
    const double mu = ts / tn;
    double tq = 0;
    for (int k = 0; k < lanes; ++k) {
      const double nk = ln[k][col], d = lm[k][col] - mu;
      if (nk != 0.0) tq += lq[k][col] + nk * (d * d);
    }
    double* o = part + (group * n_cols + col) * 3 * chunks_per_group + chunk;
    o[0] = tn; o[chunks_per_group] = ts; o[2 * (long long)chunks_per_group] = tq;
  }
}

// sum over the workgroup (<= 4 waves), returned to every thread
__device__ inline double stat_block_sum(double v, double* slot) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();                         // slot may still be read from the previous sum
  if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += slot[w];
  return t;
}

// grid (n_cols, n_groups), 64..256 threads over the chunks of one column of one group:
// pools the chunk triples about the group mean, mean = sum_c s_c / N, M2 = sum_c [q_c +
// n_c (s_c / n_c - mean)^2] -- two plain parallel sums (no serial chain of divisions).
__global__ __launch_bounds__(256) void amcx_stats_combine_kernel(
    const double* __restrict__ part, int chunks_per_group, int n_cols,
    double* __restrict__ mean_out, double* __restrict__ std_out) {
  __shared__ double slot[4];
  const int col = blockIdx.x;
  const long long group = blockIdx.y;
  const double* pn = part + (group * n_cols + col) * 3 * chunks_per_group;
  const double* ps = pn + chunks_per_group;
  const double* pq = ps + chunks_per_group;
  double n = 0, s = 0;
  for (int c = threadIdx.x; c < chunks_per_group; c += blockDim.x) {
    n += pn[c];
    s += ps[c];
  }
  n = stat_block_sum(n, slot);
  s = stat_block_sum(s, slot);
  const double mu = s / n;
  double q = 0;
  for (int c = threadIdx.x; c < chunks_per_group; c += blockDim.x) {
    const double nc = pn[c], d = ps[c] / nc - mu;
    q += pq[c] + nc * (d * d);
  }
  q = stat_block_sum(q, slot);
  if (threadIdx.x == 0) {
    mean_out[group * n_cols + col] = mu;
    std_out[group * n_cols + col] = __builtin_sqrt(q / n);
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

// fit + transform in one launch behind the statistics: picks the selected columns'
// mean / std out of the all-column statistics, turns std into sklearn's scale_ (a
// column indistinguishable from constant by the two-pass error bound gets 1,
// sklearn/preprocessing/_data.py _is_constant_feature), writes both for the caller
// (workgroup 0) and transforms kSelectRows rows per workgroup.
constexpr int kSelectRows = 1024;
struct SelectCols {
  int n;
  int c[kStatMaxCols];
};

__global__ __launch_bounds__(kBlockThreads) void amcx_select_fit_scale_kernel(
    const float* __restrict__ x, long long n_rows, long long row_stride, SelectCols sel,
    const double* __restrict__ mean_all, const double* __restrict__ std_all,
    float* __restrict__ out, long long out_stride, double* __restrict__ mean_out,
    double* __restrict__ scale_out) {
  __shared__ double mu_s[kStatMaxCols], sc_s[kStatMaxCols];
  __shared__ int col_s[kStatMaxCols];
  if ((int)threadIdx.x < sel.n) {
    const int c = sel.c[threadIdx.x];
    const double m = mean_all[c], sd = std_all[c], var = sd * sd, n = (double)n_rows;
    const double eps = 2.220446049250313e-16, nm = n * m * eps;
    const double sc = (var <= n * eps * var + nm * nm) ? 1.0 : sd;
    mu_s[threadIdx.x] = m; sc_s[threadIdx.x] = sc; col_s[threadIdx.x] = c;
    if (blockIdx.x == 0) {
      mean_out[threadIdx.x] = m;
      scale_out[threadIdx.x] = sc;
    }
  }
  __syncthreads();
  const long long row0 = (long long)blockIdx.x * kSelectRows;
  const long long left = n_rows - row0;
  const unsigned rows = left < kSelectRows ? (unsigned)left : (unsigned)kSelectRows;
  const unsigned total = rows * (unsigned)sel.n, ns = (unsigned)sel.n;
  for (unsigned i = threadIdx.x; i < total; i += kBlockThreads) {
    const unsigned r = i / ns, j = i - r * ns;
    const float c = (float)((double)x[(row0 + r) * row_stride + col_s[j]] - mu_s[j]);
    out[(row0 + r) * out_stride + j] = (float)((double)c / sc_s[j]);
  }
}

}  // namespace amcx
