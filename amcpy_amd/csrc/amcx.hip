// libamcx.so -- C ABI (include/amcx.h) over the gfx950 feature kernels.
// Build: hipcc --offload-arch=gfx950 -O3 -fPIC -shared (see build.py).
#include "../../include/amcx.h"

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <new>

#include "amcx_block_kernel.h"
#include "amcx_wave_kernel.h"
#include "amcx_fixup_kernel.h"
#include "amcx_post_kernels.h"

namespace {

thread_local char g_hip_err[256] = "";

int hip_fail(hipError_t e, const char* what) {
  snprintf(g_hip_err, sizeof g_hip_err, "%s: %s", what, hipGetErrorString(e));
  return AMCX_EHIP;
}

#define AMCX_HIP(call)                                     \
  do {                                                     \
    hipError_t e_ = (call);                                \
    if (e_ != hipSuccess) return hip_fail(e_, #call);      \
  } while (0)

bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

int resolve_variant(int32_t frame_size, int32_t variant) {
  if (frame_size < AMCX_MIN_FRAME_SIZE || frame_size > AMCX_MAX_FRAME_SIZE) return AMCX_EINVAL;
  switch (variant) {
    case AMCX_VARIANT_AUTO:
      return amcx::wave_supports(frame_size) ? AMCX_VARIANT_WAVE : AMCX_VARIANT_BLOCK;
    case AMCX_VARIANT_BLOCK:
      return AMCX_VARIANT_BLOCK;
    case AMCX_VARIANT_WAVE:
      return amcx::wave_supports(frame_size) ? AMCX_VARIANT_WAVE : AMCX_ENOTSUP;
    default:
      return AMCX_EINVAL;
  }
}

int cu_count() {
  static int cus = 0;   // benign race: every writer stores the same value
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      cus = n;
    else
      return 256;
  }
  return cus;
}

// spectral-term variant of the block kernel for this frame size
int block_mode(int N) {
  if (is_pow2(N)) return amcx::kBlockPow2;
  return (N >= amcx::kBluesteinMinN && N <= amcx::kBluesteinMaxN) ? amcx::kBlockBluestein : amcx::kBlockDirect;
}

int launch_block(const float2* iq, int64_t n_frames, int32_t N, int64_t row_stride, float* out,
                 int64_t out_stride, hipStream_t stream) {
  const int mode = block_mode(N);
  const size_t lds = (mode == amcx::kBlockBluestein ? (size_t)16 * amcx::bluestein_length(N) : (size_t)16 * N) +
                     amcx::kBlockScratchBytes + amcx::kBlockTwiddleBytes;
  auto kern = mode == amcx::kBlockPow2        ? amcx::amcx_features18_block_kernel<amcx::kBlockPow2>
              : mode == amcx::kBlockBluestein ? amcx::amcx_features18_block_kernel<amcx::kBlockBluestein>
                                              : amcx::amcx_features18_block_kernel<amcx::kBlockDirect>;
  // > 64 KiB of dynamic LDS needs the attribute.  It is set once per (kernel, device) to the most any
  // frame size can ask for, never per launch: two host threads launching different N would otherwise
  // race between one's attribute and the other's launch.
  {
    static bool attr_set[3][64] = {};
    constexpr int kMaxLds = 16 * AMCX_MAX_FRAME_SIZE + amcx::kBlockScratchBytes + amcx::kBlockTwiddleBytes;
    int dev = 0;
    AMCX_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[mode][dev]) {
      AMCX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds));
      if (dev >= 0 && dev < 64) attr_set[mode][dev] = true;   // benign race: idempotent
    }
  }
  // enough workgroups to fill every CU at the occupancy LDS allows, grid-stride beyond
  const int per_cu = (int)((160 * 1024) / lds) < 1 ? 1 : (int)((160 * 1024) / lds);
  int64_t grid = (int64_t)cu_count() * (per_cu > 8 ? 8 : per_cu) * 4;
  if (grid > n_frames) grid = n_frames;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(amcx::kBlockThreads), lds, stream, iq,
                     (long long)n_frames, (int)N, (long long)row_stride, out, (long long)out_stride);
  AMCX_HIP(hipGetLastError());
  return AMCX_OK;
}

__global__ __launch_bounds__(256) void amcx_probe_read_kernel(const float4* __restrict__ src,
                                                             long long n_vec, float* partial) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  float acc = 0.f;
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  // four independent 16-byte loads in flight per lane and iteration
  for (; i + 3 * stride < n_vec; i += 4 * stride) {
    const v4f a = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(src + i));
    const v4f b = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(src + i + stride));
    const v4f c = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(src + i + 2 * stride));
    const v4f d = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(src + i + 3 * stride));
    acc += ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w)) + ((c.x + c.y) + (c.z + c.w)) +
           ((d.x + d.y) + (d.z + d.w));
  }
  for (; i < n_vec; i += stride) {
    const float4 v = src[i];
    acc += (v.x + v.y) + (v.z + v.w);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
  __shared__ float s[4];
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
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

}  // namespace

extern "C" {

int amcx_abi_version(void) { return AMCX_ABI_VERSION; }

const char* amcx_strerror(int code) {
  switch (code) {
    case AMCX_OK: return "ok";
    case AMCX_EINVAL: return "invalid argument (null pointer, negative count, stride or frame_size out of range)";
    case AMCX_ENOTSUP: return "kernel variant does not support this frame_size";
    case AMCX_EHIP: return "HIP runtime error (see amcx_last_hip_error)";
    case AMCX_ENODEV: return "no usable gfx950 device";
    case AMCX_ENOMEM: return "device memory allocation failed";
    default: return "unknown amcx error code";
  }
}

const char* amcx_last_hip_error(void) { return g_hip_err; }

int amcx_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e == hipErrorNoDevice) return 0;
  if (e != hipSuccess) return hip_fail(e, "hipGetDeviceCount");
  int ok = 0;
  for (int d = 0; d < n; ++d) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, d) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
  }
  return ok;
}

int amcx_features18_c64_ex(const void* iq_dev, int64_t n_frames, int32_t frame_size,
                           int64_t row_stride_elems, float* out_dev, int64_t out_row_stride,
                           void* hip_stream, int32_t variant) {
  if (n_frames < 0 || row_stride_elems < frame_size || out_row_stride < AMCX_NUM_FEATURES)
    return AMCX_EINVAL;
  const int v = resolve_variant(frame_size, variant);
  if (v < 0) return v;
  if (n_frames == 0) return AMCX_OK;
  if (iq_dev == nullptr || out_dev == nullptr) return AMCX_EINVAL;
  if ((reinterpret_cast<uintptr_t>(iq_dev) & 7u) || (reinterpret_cast<uintptr_t>(out_dev) & 3u))
    return AMCX_EINVAL;
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  const float2* iq = static_cast<const float2*>(iq_dev);
  if (v == AMCX_VARIANT_WAVE) {
    hipError_t e = amcx::launch_wave(iq, n_frames, frame_size, row_stride_elems, out_dev,
                                     out_row_stride, stream, cu_count());
    if (e != hipSuccess) return hip_fail(e, "wave kernel launch");
    // frames the fp32 kernel flagged as outside its range (f5 = -inf): the range pass of the same wave machine
    // on a power-of-two pre-scaled copy (N = 1024, 2048, 4096), or the block kernel's fp64-sum routine (other N);
    // then frames with a phase step within an angle rounding of +-pi (f5 < 0): exact f5 / f9 (amcx_fixup_kernel.h)
    const bool wave_range = amcx::wave_has_range_pass(frame_size);
    if (wave_range) {
      e = amcx::launch_wave_range(iq, n_frames, frame_size, row_stride_elems, out_dev, out_row_stride, stream, cu_count());
      if (e != hipSuccess) return hip_fail(e, "range pass launch");
    }
    e = amcx::launch_fixup(iq, n_frames, frame_size, row_stride_elems, out_dev, out_row_stride, stream,
                           cu_count(), !wave_range);
    if (e != hipSuccess) return hip_fail(e, "fix-up kernel launch");
    return AMCX_OK;
  }
  return launch_block(iq, n_frames, frame_size, row_stride_elems, out_dev, out_row_stride, stream);
}

int amcx_features18_c64(const void* iq_dev, int64_t n_frames, int32_t frame_size,
                        int64_t row_stride_elems, float* out_dev, int64_t out_row_stride,
                        void* hip_stream) {
  return amcx_features18_c64_ex(iq_dev, n_frames, frame_size, row_stride_elems, out_dev,
                                out_row_stride, hip_stream, AMCX_VARIANT_AUTO);
}

// ---- host-buffer entry points over a reusable context --------------------------------------
// The context owns a stream and device scratch that only ever grows, so a loop of per-frame
// calls (the reference's usage pattern, features.py:214-232 called once per queue item) pays
// two small copies and the launches, not hipMalloc/hipFree/stream creation per call.
struct amcx_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  void* d_in = nullptr;   size_t in_cap = 0;     // uploaded rows (complex64 or complex128)
  void* d_c64 = nullptr;  size_t c64_cap = 0;    // complex128 rows rounded to complex64
  float* d_out = nullptr; size_t out_cap = 0;
};

}  // extern "C"

namespace {

int ctx_reserve(void** p, size_t* cap, size_t bytes) {
  if (*cap >= bytes) return AMCX_OK;
  if (*p) { (void)hipFree(*p); *p = nullptr; *cap = 0; }
  // grow geometrically so a slowly growing batch size does not reallocate every call
  size_t want = bytes < (size_t(1) << 20) ? bytes : bytes + bytes / 4;
  if (hipMalloc(p, want) != hipSuccess) {
    (void)hipGetLastError();
    if (want == bytes || hipMalloc(p, bytes) != hipSuccess) { (void)hipGetLastError(); *p = nullptr; return AMCX_ENOMEM; }
    want = bytes;
  }
  *cap = want;
  return AMCX_OK;
}

struct DeviceGuard {
  int prev = -1;
  hipError_t enter(int dev) {
    hipError_t e = hipGetDevice(&prev);
    if (e != hipSuccess) { prev = -1; return e; }
    return hipSetDevice(dev);
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

int ctx_run(amcx_ctx* c, const void* iq_host, bool is_c128, int64_t n_frames, int32_t frame_size,
            int64_t row_stride_elems, float* out_host, int64_t out_row_stride, int32_t variant) {
  if (c == nullptr) return AMCX_EINVAL;
  if (n_frames < 0 || row_stride_elems < frame_size || out_row_stride < AMCX_NUM_FEATURES)
    return AMCX_EINVAL;
  const int v = resolve_variant(frame_size, variant);
  if (v < 0) return v;
  if (n_frames == 0) return AMCX_OK;
  if (iq_host == nullptr || out_host == nullptr) return AMCX_EINVAL;
  DeviceGuard guard;
  AMCX_HIP(guard.enter(c->device));
  const size_t elem = is_c128 ? 16 : 8;
  const size_t row_in = (size_t)frame_size * elem;
  // rows are packed on the device (row stride == frame_size), so rows longer than frame_size
  // (feature_extraction.py:68) cost no HBM or PCIe bytes; at most ~512 MiB go up at a time
  int64_t per = (int64_t)((512ull << 20) / row_in);
  if (per < 1) per = 1;
  if (per > n_frames) per = n_frames;
  int rc = ctx_reserve(&c->d_in, &c->in_cap, row_in * (size_t)per);
  if (rc == AMCX_OK && is_c128) rc = ctx_reserve(&c->d_c64, &c->c64_cap, (size_t)frame_size * 8 * (size_t)per);
  if (rc == AMCX_OK)
    rc = ctx_reserve(reinterpret_cast<void**>(&c->d_out), &c->out_cap,
                     sizeof(float) * AMCX_NUM_FEATURES * (size_t)per);
  if (rc != AMCX_OK) return rc;
  const char* src = static_cast<const char*>(iq_host);
  hipError_t e = hipSuccess;
  for (int64_t f0 = 0; f0 < n_frames; f0 += per) {
    const int64_t nf = (n_frames - f0) < per ? (n_frames - f0) : per;
    e = hipMemcpy2DAsync(c->d_in, row_in, src + (size_t)f0 * (size_t)row_stride_elems * elem,
                         (size_t)row_stride_elems * elem, row_in, (size_t)nf, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) break;
    const void* d_frames = c->d_in;
    if (is_c128) {
      hipLaunchKernelGGL(amcx_c128_to_c64_kernel, dim3(2048), dim3(256), 0, c->stream,
                         static_cast<const double2*>(c->d_in), (long long)nf, (int)frame_size,
                         (long long)frame_size, static_cast<float2*>(c->d_c64));
      e = hipGetLastError();
      if (e != hipSuccess) break;
      d_frames = c->d_c64;
    }
    rc = amcx_features18_c64_ex(d_frames, nf, frame_size, frame_size, c->d_out, AMCX_NUM_FEATURES, c->stream, v);
    if (rc != AMCX_OK) break;
    e = hipMemcpy2DAsync(out_host + (size_t)f0 * (size_t)out_row_stride, sizeof(float) * (size_t)out_row_stride,
                         c->d_out, sizeof(float) * AMCX_NUM_FEATURES, sizeof(float) * AMCX_NUM_FEATURES,
                         (size_t)nf, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);   // the scratch is reused by the next chunk / call
    if (e != hipSuccess) break;
  }
  if (rc == AMCX_OK && e != hipSuccess) rc = hip_fail(e, "amcx_ctx_features18 host entry");
  return rc;
}

}  // namespace

extern "C" {

int amcx_ctx_create(int32_t device, amcx_ctx** ctx_out) {
  if (ctx_out == nullptr) return AMCX_EINVAL;
  *ctx_out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    (void)hipGetLastError();
    return AMCX_ENODEV;
  }
  DeviceGuard guard;
  AMCX_HIP(guard.enter(device));
  amcx_ctx* c = new (std::nothrow) amcx_ctx();
  if (c == nullptr) return AMCX_ENOMEM;
  c->device = device;
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete c; return hip_fail(e, "hipStreamCreateWithFlags"); }
  *ctx_out = c;
  return AMCX_OK;
}

int amcx_ctx_destroy(amcx_ctx* c) {
  if (c == nullptr) return AMCX_OK;
  DeviceGuard guard;
  (void)guard.enter(c->device);
  if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
  if (c->d_in) (void)hipFree(c->d_in);
  if (c->d_c64) (void)hipFree(c->d_c64);
  if (c->d_out) (void)hipFree(c->d_out);
  delete c;
  return AMCX_OK;
}

int amcx_ctx_features18_c64_host(amcx_ctx* ctx, const void* iq_host, int64_t n_frames, int32_t frame_size,
                                 int64_t row_stride_elems, float* out_host, int64_t out_row_stride,
                                 int32_t variant) {
  return ctx_run(ctx, iq_host, false, n_frames, frame_size, row_stride_elems, out_host, out_row_stride, variant);
}

int amcx_ctx_features18_c128_host(amcx_ctx* ctx, const void* iq_host, int64_t n_frames, int32_t frame_size,
                                  int64_t row_stride_elems, float* out_host, int64_t out_row_stride,
                                  int32_t variant) {
  return ctx_run(ctx, iq_host, true, n_frames, frame_size, row_stride_elems, out_host, out_row_stride, variant);
}

// one-shot forms: a context for the duration of the call
static int one_shot(const void* iq_host, bool is_c128, int64_t n_frames, int32_t frame_size,
                    int64_t row_stride_elems, float* out_host, int64_t out_row_stride, int32_t device,
                    int32_t variant) {
  // argument errors are reported before a device is looked for (tests/test_host_cpu.py runs without one)
  if (n_frames < 0 || row_stride_elems < frame_size || out_row_stride < AMCX_NUM_FEATURES)
    return AMCX_EINVAL;
  const int v = resolve_variant(frame_size, variant);
  if (v < 0) return v;
  if (n_frames == 0) return AMCX_OK;
  if (iq_host == nullptr || out_host == nullptr) return AMCX_EINVAL;
  amcx_ctx* c = nullptr;
  int rc = amcx_ctx_create(device, &c);
  if (rc != AMCX_OK) return rc;
  rc = ctx_run(c, iq_host, is_c128, n_frames, frame_size, row_stride_elems, out_host, out_row_stride, v);
  (void)amcx_ctx_destroy(c);
  return rc;
}

int amcx_features18_c64_host(const void* iq_host, int64_t n_frames, int32_t frame_size,
                             int64_t row_stride_elems, float* out_host, int64_t out_row_stride,
                             int32_t device, int32_t variant) {
  return one_shot(iq_host, false, n_frames, frame_size, row_stride_elems, out_host, out_row_stride, device, variant);
}

int amcx_features18_c128_host(const void* iq_host, int64_t n_frames, int32_t frame_size,
                              int64_t row_stride_elems, float* out_host, int64_t out_row_stride,
                              int32_t device, int32_t variant) {
  return one_shot(iq_host, true, n_frames, frame_size, row_stride_elems, out_host, out_row_stride, device, variant);
}

int amcx_kernel_name(int32_t frame_size, int32_t variant, char* buf, int32_t buf_len) {
  if (buf == nullptr || buf_len <= 0) return AMCX_EINVAL;
  const int v = resolve_variant(frame_size, variant);
  if (v < 0) return v;
  const char* name = (v == AMCX_VARIANT_WAVE) ? amcx::wave_kernel_name(frame_size)
                     : block_mode(frame_size) == amcx::kBlockPow2      ? "amcx_features18_block_kernel<1>"
                     : block_mode(frame_size) == amcx::kBlockBluestein ? "amcx_features18_block_kernel<2>"
                                                                       : "amcx_features18_block_kernel<0>";
  snprintf(buf, (size_t)buf_len, "%s", name);
  return AMCX_OK;
}

int amcx_probe_read_bw(const void* src_dev, int64_t n_bytes, float* partial_dev, void* hip_stream) {
  if (src_dev == nullptr || partial_dev == nullptr || n_bytes < 0 || (n_bytes & 15)) return AMCX_EINVAL;
  if (n_bytes == 0) return AMCX_OK;
  hipLaunchKernelGGL(amcx_probe_read_kernel, dim3(4096), dim3(256), 0,
                     static_cast<hipStream_t>(hip_stream), static_cast<const float4*>(src_dev),
                     (long long)(n_bytes / 16), partial_dev);
  AMCX_HIP(hipGetLastError());
  return AMCX_OK;
}

int amcx_group_stats_f32(const float* x_dev, int64_t n_groups, int64_t rows_per_group,
                         int64_t row_stride, int32_t n_cols, double* mean_dev, double* std_dev,
                         void* hip_stream) {
  if (n_groups < 0 || rows_per_group < 1 || n_cols < 1 || n_cols > amcx::kStatMaxCols ||
      row_stride < n_cols || n_groups > 0x7fffffffLL)
    return AMCX_EINVAL;
  if (n_groups == 0) return AMCX_OK;
  if (x_dev == nullptr || mean_dev == nullptr || std_dev == nullptr) return AMCX_EINVAL;
  hipLaunchKernelGGL(amcx::amcx_group_stats_kernel, dim3((unsigned)n_groups), dim3(amcx::kBlockThreads), 0,
                     static_cast<hipStream_t>(hip_stream), x_dev, (long long)rows_per_group,
                     (long long)row_stride, (int)n_cols, mean_dev, std_dev);
  AMCX_HIP(hipGetLastError());
  return AMCX_OK;
}

int amcx_select_scale_f32(const float* x_dev, int64_t n_rows, int64_t row_stride,
                          const int32_t* cols_dev, int32_t n_sel, const double* mean_dev,
                          const double* scale_dev, float* out_dev, int64_t out_stride,
                          void* hip_stream) {
  if (n_rows < 0 || n_sel < 1 || out_stride < n_sel || row_stride < 1) return AMCX_EINVAL;
  if (n_rows == 0) return AMCX_OK;
  if (!x_dev || !cols_dev || !mean_dev || !scale_dev || !out_dev) return AMCX_EINVAL;
  const int64_t total = n_rows * n_sel;
  int64_t grid = (total + amcx::kBlockThreads - 1) / amcx::kBlockThreads;
  const int64_t cap = (int64_t)cu_count() * 8;
  if (grid > cap) grid = cap;
  hipLaunchKernelGGL(amcx::amcx_select_scale_kernel, dim3((unsigned)grid), dim3(amcx::kBlockThreads), 0,
                     static_cast<hipStream_t>(hip_stream), x_dev, (long long)n_rows, (long long)row_stride,
                     cols_dev, (int)n_sel, mean_dev, scale_dev, out_dev, (long long)out_stride);
  AMCX_HIP(hipGetLastError());
  return AMCX_OK;
}

}  // extern "C"
