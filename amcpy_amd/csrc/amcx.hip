// libamcx.so -- C ABI (include/amcx.h) over the gfx950 feature kernels.
// Build: hipcc --offload-arch=gfx950 -O3 -fPIC -shared (see build.py).
#include "../../include/amcx.h"

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <new>

#include "amcx_block_kernel.h"
#include "amcx_stream_kernel.h"
#include "amcx_wave_kernel.h"
#include "amcx_quad_kernel.h"
#include "amcx_group_kernel.h"
#include "amcx_short_kernel.h"
#include "amcx_post_kernels.h"
#include "amcx_pack_kernel.h"
#include "amcx_upload.h"

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
  const bool block_ok = frame_size <= AMCX_MAX_BLOCK_FRAME_SIZE;      // every size in range since ABI 6
  switch (variant) {
    case AMCX_VARIANT_AUTO:
      return amcx::wave_supports(frame_size) ? AMCX_VARIANT_WAVE : block_ok ? AMCX_VARIANT_BLOCK : AMCX_ENOTSUP;
    case AMCX_VARIANT_BLOCK:
      return block_ok ? AMCX_VARIANT_BLOCK : AMCX_ENOTSUP;
    case AMCX_VARIANT_WAVE:
      return amcx::wave_supports(frame_size) ? AMCX_VARIANT_WAVE : AMCX_ENOTSUP;
    default:
      return AMCX_EINVAL;
  }
}

// CUs of the calling thread's current device (cached per device: a node may mix partitioned and whole GPUs)
int cu_count() {
  static int cus[64] = {};   // benign race: every writer of a slot stores the same value
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 256; }
  if (dev >= 0 && dev < 64 && cus[dev] > 0) return cus[dev];
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return 256;
  }
  if (dev >= 0 && dev < 64) cus[dev] = n;
  return n;
}

// include/amcx.h, DEVICE OWNERSHIP: a device pointer must live on the current device.  Pointers the
// runtime does not know (or host-visible ones) are let through -- the launch itself will say.
bool on_another_device(const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (a.type != hipMemoryTypeDevice) return false;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.device != dev;
}

// spectral-term variant of the block kernel for this frame size
int block_mode(int N) {
  if (is_pow2(N)) return amcx::kBlockPow2;
  if (N >= amcx::kBluesteinMinN && N <= amcx::kBluesteinMaxN) return amcx::kBlockBluestein;
  return N > amcx::kBluesteinMaxN ? amcx::kBlockBluesteinBig : amcx::kBlockDirect;
}

// 8192 < N <= 32768: one 1024-thread workgroup per frame, the frame read where it lies (amcx_stream_kernel.h).  With a
// workspace of at least one workgroup's share the spectral term is an FFT through it (Bluestein for the sizes that are
// not powers of two), otherwise the DFT by its definition.
struct StreamPlan {
  int M;            // transform length of the FFT form
  bool chirped;     // not a power of two: one more buffer of M for the chirp's spectrum
  int64_t grid;     // workgroups = frames in flight
};

StreamPlan stream_plan(int32_t N, int64_t n_frames) {
  StreamPlan p;
  p.M = amcx::stream::conv_length(N);
  p.chirped = !is_pow2(N);
  p.grid = cu_count();                                   // one resident workgroup per CU, grid-stride beyond
  if (p.grid > n_frames) p.grid = n_frames;
  if (p.grid < 1) p.grid = 1;
  return p;
}

int set_stream_lds_once(const void* kern, int which) {
  static bool attr_set[3][64] = {};                      // once per (kernel, device), to the most any frame size asks for
  int dev = 0;
  AMCX_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_set[which][dev]) {
    const size_t most = which == 0 ? amcx::stream::lds_bytes(amcx::stream::kMaxN)
                        : which == 1 ? amcx::stream::lds_bytes_fft(amcx::stream::kMaxN)
                                     : (size_t)amcx::stream::kTileBytes + amcx::stream::kFftTabBytes;
    AMCX_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)most));
    if (dev >= 0 && dev < 64) attr_set[which][dev] = true;   // benign race: idempotent
  }
  return AMCX_OK;
}

int launch_stream(const float2* iq, int64_t n_frames, int32_t N, int64_t row_stride, float* out,
                  int64_t out_stride, hipStream_t stream, void* ws, int64_t ws_bytes) {
  StreamPlan p = stream_plan(N, n_frames);
  const int64_t per = (int64_t)p.M * (int64_t)sizeof(float2);
  int64_t fit = ws != nullptr && ws_bytes > 0 ? ws_bytes / per - (p.chirped ? 1 : 0) : 0;   // workgroups the workspace has room for
  if (fit >= 1 && (reinterpret_cast<uintptr_t>(ws) & 7u) == 0) {
    if (p.grid > fit) p.grid = fit;
    float2* const base = static_cast<float2*>(ws);
    const float2* bspec = nullptr;
    float2* bufs = base;
    if (p.chirped) {
      auto chirp = amcx::stream::amcx_stream_chirp_kernel;
      const int rc = set_stream_lds_once(reinterpret_cast<const void*>(chirp), 2);
      if (rc != AMCX_OK) return rc;
      hipLaunchKernelGGL(chirp, dim3(1), dim3(amcx::stream::kThreads),
                         (size_t)amcx::stream::kTileBytes + amcx::stream::kFftTabBytes, stream, base, (int)N, p.M);
      AMCX_HIP(hipGetLastError());
      bspec = base;
      bufs = base + p.M;
    }
    auto kern = amcx::stream::amcx_features18_stream_kernel<true>;
    const int rc = set_stream_lds_once(reinterpret_cast<const void*>(kern), 1);
    if (rc != AMCX_OK) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)p.grid), dim3(amcx::stream::kThreads), amcx::stream::lds_bytes_fft(N), stream,
                       iq, (long long)n_frames, (int)N, (long long)row_stride, out, (long long)out_stride, bspec, bufs, p.M);
    AMCX_HIP(hipGetLastError());
    return AMCX_OK;
  }
  auto kern = amcx::stream::amcx_features18_stream_kernel<false>;
  const int rc = set_stream_lds_once(reinterpret_cast<const void*>(kern), 0);
  if (rc != AMCX_OK) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)p.grid), dim3(amcx::stream::kThreads), amcx::stream::lds_bytes(N), stream, iq,
                     (long long)n_frames, (int)N, (long long)row_stride, out, (long long)out_stride,
                     static_cast<const float2*>(nullptr), static_cast<float2*>(nullptr), 0);
  AMCX_HIP(hipGetLastError());
  return AMCX_OK;
}

int launch_block(const float2* iq, int64_t n_frames, int32_t N, int64_t row_stride, float* out,
                 int64_t out_stride, hipStream_t stream, void* ws = nullptr, int64_t ws_bytes = 0) {
  if (N > amcx::kBlockMaxN) return launch_stream(iq, n_frames, N, row_stride, out, out_stride, stream, ws, ws_bytes);
  const int mode = block_mode(N);
  const size_t lds = (mode == amcx::kBlockBluestein      ? (size_t)16 * amcx::bluestein_length(N)
                      : mode == amcx::kBlockBluesteinBig ? (size_t)8 * amcx::kBluesteinBigM
                                                         : (size_t)16 * N) +
                     amcx::kBlockScratchBytes + amcx::kBlockTwiddleBytes;
  auto kern = mode == amcx::kBlockPow2           ? amcx::amcx_features18_block_kernel<amcx::kBlockPow2>
              : mode == amcx::kBlockBluestein    ? amcx::amcx_features18_block_kernel<amcx::kBlockBluestein>
              : mode == amcx::kBlockBluesteinBig ? amcx::amcx_features18_block_kernel<amcx::kBlockBluesteinBig>
                                                 : amcx::amcx_features18_block_kernel<amcx::kBlockDirect>;
  // > 64 KiB of dynamic LDS needs the attribute.  It is set once per (kernel, device) to the most any
  // frame size can ask for, never per launch: two host threads launching different N would otherwise
  // race between one's attribute and the other's launch.
  {
    static bool attr_set[4][64] = {};
    constexpr int kMaxLds = 16 * amcx::kBlockMaxN + amcx::kBlockScratchBytes + amcx::kBlockTwiddleBytes;
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

// The instruction-issue ceiling under the board's power cap: 16 wavefronts per CU (4 per SIMD, the N = 2048 kernel's
// occupancy), each running `iters` trips of 32 independent v_fma_f32 (8 chains x 4) on registers -- no memory traffic.
// Lane 0 of every wave leaves its shader-clock cycles and its 100 MHz real-time ticks, from which the clock follows.
__global__ __launch_bounds__(1024) void amcx_probe_fma_kernel(int iters, float* sink, unsigned long long* ticks) {
  float a0 = (float)threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f,
        a6 = a0 + 6.f, a7 = a0 + 7.f;
  const float b0 = 1.0001f, b1 = 0.9999f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      asm volatile(
          "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
          "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  const float s = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  if (s == 12345.678f) sink[0] = s;                       // keeps the chains alive; never true in practice
  if ((threadIdx.x & 63) == 0) {
    const long long w = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    ticks[2 * w] = t1 - t0;
    ticks[2 * w + 1] = r1 - r0;
  }
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
    case AMCX_EIO: return "reading the container's file failed (see amcx_last_hip_error for the errno text)";
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

int64_t amcx_features18_workspace_bytes(int32_t frame_size, int64_t n_frames, int32_t variant) {
  if (n_frames < 0) return -1;
  const int v = resolve_variant(frame_size, variant);
  if (v < 0) return -1;
  if (v != AMCX_VARIANT_BLOCK || frame_size <= amcx::kBlockMaxN || n_frames == 0) return 0;
  const StreamPlan p = stream_plan(frame_size, n_frames);
  return (p.grid + (p.chirped ? 1 : 0)) * (int64_t)p.M * (int64_t)sizeof(float2);
}

int amcx_features18_c64_ws(const void* iq_dev, int64_t n_frames, int32_t frame_size,
                           int64_t row_stride_elems, float* out_dev, int64_t out_row_stride,
                           void* hip_stream, int32_t variant, void* workspace_dev, int64_t workspace_bytes) {
  if (n_frames < 0 || row_stride_elems < frame_size || out_row_stride < AMCX_NUM_FEATURES)
    return AMCX_EINVAL;
  const int v = resolve_variant(frame_size, variant);
  if (v < 0) return v;
  if (n_frames == 0) return AMCX_OK;
  if (iq_dev == nullptr || out_dev == nullptr) return AMCX_EINVAL;
  if ((reinterpret_cast<uintptr_t>(iq_dev) & 7u) || (reinterpret_cast<uintptr_t>(out_dev) & 3u))
    return AMCX_EINVAL;
  if (on_another_device(iq_dev) || on_another_device(out_dev)) return AMCX_EINVAL;
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  const float2* iq = static_cast<const float2*>(iq_dev);
  if (v == AMCX_VARIANT_WAVE) {
    // N = 8192: four waves per frame (amcx_quad_kernel.h); 16384 / 32768: eight / sixteen (amcx_group_kernel.h); 128 / 256 / 512:
    // four frames per wave (amcx_short_kernel.h); every other wave size: one wave per frame
    hipError_t e;
    if (frame_size == amcx::quad::kN)
      e = amcx::quad::launch_quad(iq, n_frames, row_stride_elems, out_dev, out_row_stride, stream, cu_count());
    else if (frame_size == amcx::group::G<8>::kN)
      e = amcx::group::launch_group<8>(iq, n_frames, row_stride_elems, out_dev, out_row_stride, stream, cu_count());
    else if (frame_size == amcx::group::G<16>::kN)
      e = amcx::group::launch_group<16>(iq, n_frames, row_stride_elems, out_dev, out_row_stride, stream, cu_count());
    else if (amcx::shortk::short_supports(frame_size))       // 128, 256, 512: four frames per wave (amcx_short_kernel.h)
      e = amcx::shortk::launch_short(iq, n_frames, frame_size, row_stride_elems, out_dev, out_row_stride, stream, cu_count());
    else
      e = amcx::launch_wave(iq, n_frames, frame_size, row_stride_elems, out_dev, out_row_stride, stream, cu_count());
    if (e != hipSuccess) return hip_fail(e, "wave kernel launch");
    // every throughput kernel (N = 128 ... 4096 one wave per frame, N = 8192 the quad, 16384 / 32768 the group) has re-run the frames outside its
    // fp32 sums' range itself -- one launch, rows final -- and finished frames with a phase step within an angle rounding
    // of +-pi in its finaliser.
    return AMCX_OK;
  }
  if (workspace_dev != nullptr && on_another_device(workspace_dev)) return AMCX_EINVAL;
  return launch_block(iq, n_frames, frame_size, row_stride_elems, out_dev, out_row_stride, stream, workspace_dev,
                      workspace_bytes < 0 ? 0 : workspace_bytes);
}

int amcx_features18_c64_ex(const void* iq_dev, int64_t n_frames, int32_t frame_size,
                           int64_t row_stride_elems, float* out_dev, int64_t out_row_stride,
                           void* hip_stream, int32_t variant) {
  // The any-size path above 8192 samples runs its FFT form through a workspace from the stream-ordered allocator
  // (as amcx_group_stats_f32 does) -- unless the stream is being captured into a graph, or the allocator has nothing:
  // then the DFT by its definition, which needs none (amcx_stream_kernel.h).  Everything else allocates nothing.
  const int64_t want = n_frames > 0 && iq_dev != nullptr && out_dev != nullptr && row_stride_elems >= frame_size &&
                               out_row_stride >= AMCX_NUM_FEATURES
                           ? amcx_features18_workspace_bytes(frame_size, n_frames, variant) : 0;
  if (want <= 0)
    return amcx_features18_c64_ws(iq_dev, n_frames, frame_size, row_stride_elems, out_dev, out_row_stride, hip_stream,
                                  variant, nullptr, 0);
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
  void* ws = nullptr;
  if (cap == hipStreamCaptureStatusNone && hipMallocAsync(&ws, (size_t)want, st) != hipSuccess) {
    (void)hipGetLastError();
    ws = nullptr;
  }
  // Which form ran is otherwise invisible (amcx_kernel_name says amcx_features18_stream_kernel either way, the results
  // agree to 8e-7): AMCX_VERBOSE=1 says it once per process on stderr when the O(N^2) form is taken.  A caller that
  // must not fall back hands in its own workspace (amcx_features18_c64_ws), as amcpy_amd.features.features18 does.
  if (ws == nullptr) {
    static const bool verbose = [] { const char* t = getenv("AMCX_VERBOSE"); return t != nullptr && t[0] != '\0' && t[0] != '0'; }();
    static std::atomic<bool> said{false};
    if (verbose && !said.exchange(true))
      fprintf(stderr, "amcx: frame_size %d runs the DFT by its definition (O(N^2)): %s; amcx_features18_c64_ws with %lld "
                      "bytes of workspace selects the FFT form\n", (int)frame_size,
              cap != hipStreamCaptureStatusNone ? "the stream is being captured" : "hipMallocAsync had no workspace", (long long)want);
  }
  const int rc = amcx_features18_c64_ws(iq_dev, n_frames, frame_size, row_stride_elems, out_dev, out_row_stride,
                                        hip_stream, variant, ws, ws != nullptr ? want : 0);
  if (ws != nullptr) {
    const hipError_t fe = hipFreeAsync(ws, st);
    if (rc == AMCX_OK) AMCX_HIP(fe);
  }
  return rc;
}

int amcx_features18_c64(const void* iq_dev, int64_t n_frames, int32_t frame_size,
                        int64_t row_stride_elems, float* out_dev, int64_t out_row_stride,
                        void* hip_stream) {
  return amcx_features18_c64_ex(iq_dev, n_frames, frame_size, row_stride_elems, out_dev,
                                out_row_stride, hip_stream, AMCX_VARIANT_AUTO);
}

// ---- host-buffer entry points over a reusable context --------------------------------------
// The context owns two streams, pinned staging slots and device scratch that only ever grow, so a
// loop of per-frame calls (the reference's usage pattern, features.py:214-232 called once per queue
// item) pays two small copies and the launches, not hipMalloc/hipFree/stream creation per call, and
// a whole container goes up through the staged, overlapped path (ctx_run_strided).
struct amcx_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  float* d_out = nullptr; size_t out_cap = 0;
  // strided containers (amcx_ctx_features18_strided_host): staging pool, three pinned slots, a second stream
  amcx::Pool pool;
  int threads = 0;                               // 0: not yet sized
  size_t slot_bytes = size_t(32) << 20;
  bool round_on_device = false;
  hipStream_t copy_stream = nullptr;
  char* pin = nullptr;     size_t pin_cap = 0;   // kPinSlots x slot
  void* d_slab = nullptr;  size_t slab_cap = 0;  // 2 x slot: uploaded chunks
  void* d_frames = nullptr; size_t frames_cap = 0;   // plane-major sources: the frame-major complex64 image
  void* d_ws = nullptr; size_t ws_cap = 0;           // the any-size path's FFT workspace (frame sizes above 8192, amcx_features18_c64_ws)
  hipEvent_t up_done[3] = {nullptr, nullptr, nullptr};
  hipEvent_t slab_free[2] = {nullptr, nullptr};
  float* out_pin = nullptr; size_t out_pin_cap = 0;   // the result lands in pinned memory first
  amcx_upload_stats stats = {};
  // host placement (amcx_upload.h, NumaPlace): the CPUs local to this device; staging threads, the calling thread for the
  // duration of a threaded upload, and with it the pinned slots it allocates, stay on them.  Empty: nothing is bound.
  char pci_bus_id[32] = {0};
  int numa_node = -1;
  std::vector<int> bind_cpus;
  // small row-major calls (a loop of per-frame calculate_features calls): the copy in, the launches and the copy
  // out as ONE instantiated graph per (frames, frame size, variant, element type, buffers), relaunched
  struct SmallGraph {
    hipGraphExec_t exec = nullptr;
    int64_t frames = 0; int32_t frame_size = 0, variant = 0; bool c128 = false, zero_copy = false;
    const void* pin = nullptr; const void* slab = nullptr; const void* out = nullptr; const void* out_pin = nullptr;
    const void* ws = nullptr;      // the workspace the captured kernel node points into
    size_t slot = 0;
  };
  SmallGraph graphs[4];
  int graph_next = 0;               // slot the next capture replaces
  int graph_hits = 0, graph_misses = 0;
  bool graphs_ok = true;            // false: capture failed once, or the calls vary too much for a cache of four
  // calls in flight on this context (a context serves one call at a time; the counter is there so that
  // amcx_ctx_bind_cpus can refuse to rebuild the staging pool's binding under a running upload)
  std::atomic<int> in_call{0};
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

// The context's own workspace for a call of `frames` frames (nothing for the frame sizes that need none).  Reserved
// BEFORE any capture begins: a captured kernel node points into it, and an allocation inside a capture is not allowed.
// A context that cannot have it runs the workspace-free form.
void ctx_reserve_ws(amcx_ctx* c, int32_t N, int64_t frames, int32_t variant) {
  const int64_t want = amcx_features18_workspace_bytes(N, frames, variant);
  if (want > 0 && ctx_reserve(&c->d_ws, &c->ws_cap, (size_t)want) != AMCX_OK) { c->d_ws = nullptr; c->ws_cap = 0; }
}

int ctx_features(amcx_ctx* c, const void* rows, int64_t frames, int32_t N, float* out, int32_t variant) {
  const int64_t want = amcx_features18_workspace_bytes(N, frames, variant);
  const bool have = want > 0 && c->d_ws != nullptr && c->ws_cap >= (size_t)want;
  return amcx_features18_c64_ws(rows, frames, N, N, out, AMCX_NUM_FEATURES, c->stream, variant, have ? c->d_ws : nullptr,
                                have ? want : 0);
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

// ---- strided host containers ----------------------------------------------------------------------
constexpr int kPinSlots = 3;

using amcx::classify_layout;      // amcx_upload.h: which axis is contiguous decides how a container goes up

double wall_now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int strided_prepare(amcx_ctx* c, size_t slot, size_t dslot, size_t frames_bytes, size_t out_bytes, bool threaded) {
  if (c->threads == 0) {
    unsigned hw = std::thread::hardware_concurrency();
    if (!c->bind_cpus.empty()) hw = (unsigned)amcx::allowed_subset(c->bind_cpus).size();   // this device's share of the host
    c->threads = (int)(hw == 0 ? 4 : hw > 8 ? 8 : hw);
  }
  if (threaded) c->pool.resize(c->threads);      // the staging threads start with the first call that has work for them
  if (c->copy_stream == nullptr) AMCX_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  for (auto& ev : c->up_done) if (ev == nullptr) AMCX_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  for (auto& ev : c->slab_free) if (ev == nullptr) AMCX_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  if (c->pin_cap < kPinSlots * slot) {
    if (c->pin) { (void)hipHostFree(c->pin); c->pin = nullptr; c->pin_cap = 0; }
    if (hipHostMalloc(reinterpret_cast<void**>(&c->pin), kPinSlots * slot, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      c->pin = nullptr;
      return AMCX_ENOMEM;
    }
    c->pin_cap = kPinSlots * slot;
  }
  if (c->out_pin_cap < out_bytes) {
    if (c->out_pin) { (void)hipHostFree(c->out_pin); c->out_pin = nullptr; c->out_pin_cap = 0; }
    const size_t want = out_bytes + out_bytes / 4 + 4096;
    if (hipHostMalloc(reinterpret_cast<void**>(&c->out_pin), want, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      c->out_pin = nullptr;
      return AMCX_ENOMEM;
    }
    c->out_pin_cap = want;
  }
  int rc = ctx_reserve(&c->d_slab, &c->slab_cap, 2 * dslot);
  if (rc == AMCX_OK && frames_bytes) rc = ctx_reserve(&c->d_frames, &c->frames_cap, frames_bytes);
  if (rc == AMCX_OK) rc = ctx_reserve(reinterpret_cast<void**>(&c->d_out), &c->out_cap, out_bytes);
  return rc;
}

// src: memory (re / im) or a file (fd, byte offsets); src.kind is checked here, src.io_error is set here
int ctx_run_strided(amcx_ctx* c, amcx::Source src, int64_t S, int64_t K,
                    int32_t N, int64_t ss, int64_t sk, int64_t sn, float* out_host, int64_t out_row_stride,
                    int32_t variant) {
  if (c == nullptr) return AMCX_EINVAL;
  const int32_t kind = src.kind;
  if (S < 0 || K < 0 || ss < 0 || sk < 0 || sn < 0 || out_row_stride < AMCX_NUM_FEATURES ||
      kind < AMCX_SRC_C64 || kind > AMCX_SRC_F64_SPLIT)
    return AMCX_EINVAL;
  const int v = resolve_variant(N, variant);
  if (v < 0) return v;
  if (S == 0 || K == 0) return AMCX_OK;
  if (S > (int64_t(1) << 40) / K) return AMCX_EINVAL;
  if ((src.fd < 0 && src.re == nullptr) || (src.fd >= 0 && src.re_off < 0) || out_host == nullptr) return AMCX_EINVAL;
  struct InCall {
    std::atomic<int>& n;
    explicit InCall(std::atomic<int>& c) : n(c) { n.fetch_add(1, std::memory_order_acq_rel); }
    ~InCall() { n.fetch_sub(1, std::memory_order_acq_rel); }
  } in_call(c->in_call);
  if (kind < AMCX_SRC_F32_SPLIT) { src.im = nullptr; src.im_off = -1; }
  std::atomic<int> io_error{0};
  src.io_error = &io_error;
  const int64_t F = S * K;
  bool rows = false, inner_snr = false;
  amcx::RunMap map;
  if (!classify_layout(S, K, N, ss, sk, sn, &rows, &inner_snr, &map)) return AMCX_ENOTSUP;
  if (!rows && S > 0x7fffffffLL) return AMCX_EINVAL;             // the transposition kernel indexes the snr axis with an int
  const bool as_c128 = c->round_on_device && kind == AMCX_SRC_C128;
  const size_t esz = as_c128 ? 16 : 8;
  const size_t src_esz = kind == AMCX_SRC_C64 ? 8 : kind == AMCX_SRC_C128 ? 16 : kind == AMCX_SRC_F32_SPLIT ? 4 : 8;
  const int64_t unit = rows ? N : F;                      // staged elements per chunk unit (a frame / a plane)
  const int64_t n_units = rows ? F : N;
  size_t slot = c->slot_bytes;
  const size_t total_staged = (size_t)unit * esz * (size_t)n_units;
  if (slot > total_staged) slot = total_staged;                   // a per-frame call pins kilobytes, not 3 x 32 MiB
  if (slot < (size_t)unit * esz) slot = (size_t)unit * esz;       // a slot holds at least one frame / one plane
  if (slot > (size_t(4) << 30)) return AMCX_ENOMEM;               // > 4 GiB per plane: split the call by snr
  slot = (slot + 4095) & ~size_t(4095);

  DeviceGuard guard;
  AMCX_HIP(guard.enter(c->device));
  const double t_start = wall_now();
  // rows of complex128 rounded on the device: each device slot is followed by room for its rounded rows
  const size_t dslot = (rows && as_c128) ? slot + slot / 2 : slot;
  const bool threaded = total_staged >= (size_t(1) << 20);        // below 1 MiB a condition-variable wake costs more than the copy
  // an upload worth its staging threads runs on the device's own socket, this thread included: it stages, and the pinned
  // slots strided_prepare may allocate are placed where it runs (a per-frame call is not worth two affinity system calls)
  static const std::vector<int> kNoCpus;
  amcx::AffinityGuard on_local_cpus(threaded ? c->bind_cpus : kNoCpus);
  int rc = strided_prepare(c, slot, dslot, rows ? 0 : (size_t)F * N * 8, sizeof(float) * AMCX_NUM_FEATURES * (size_t)F,
                           threaded);
  if (rc != AMCX_OK) return rc;
  ctx_reserve_ws(c, N, F, v);                                      // (frame sizes above 8192 only) before any capture below
  amcx::Pool inline_pool;                                          // size 1: stage_runs runs on the caller
  amcx_upload_stats st = {};
  amcx::Pool& pool = threaded ? c->pool : inline_pool;
  st.frames = F; st.threads = pool.size(); st.plane_major = rows ? 0 : 1; st.from_file = src.fd >= 0 ? 1 : 0;
  st.source_bytes = F * (int64_t)N * (int64_t)src_esz * ((kind >= AMCX_SRC_F32_SPLIT && src.has_im()) ? 2 : 1);

  hipError_t e = hipSuccess;
  const double t_loop = wall_now();
  // ---- small row-major calls: one graph launch ------------------------------------------------------------------
  // A per-frame loop (the reference's calculate_features per queue item, features.py:214-232) is launch-bound: copy in,
  // one or two conversions / kernels, copy out, a synchronisation -- seven runtime calls around 10 us of
  // GPU work.  Captured once per shape into a graph on the compute stream, a call is: stage into the pinned slot,
  // hipGraphLaunch, hipStreamSynchronize.  Anything that does not fit (several chunks, planes, staging threads) and any
  // failure to capture takes the general path below.
  if (rows && !threaded && total_staged <= slot && c->graphs_ok && getenv("AMCX_NO_GRAPH") == nullptr) {   // one chunk
    char* pinned = c->pin;
    char* dev = static_cast<char*>(c->d_slab);
    const size_t bytes = (size_t)n_units * (size_t)unit * esz;
    const size_t out_bytes = sizeof(float) * AMCX_NUM_FEATURES * (size_t)F;
    double t0 = wall_now();
    amcx::stage_runs(inline_pool, pinned, src, map, 0, n_units, as_c128);
    st.seconds_staging += wall_now() - t0;
    if (io_error.load() != 0) {
      snprintf(g_hip_err, sizeof g_hip_err, "reading the container's file: %s", strerror(io_error.load()));
      return AMCX_EIO;
    }
    // a few frames of complex64: the kernels read the pinned slot and write the pinned result themselves (host memory
    // from hipHostMalloc is mapped into the device's address space) -- two copy nodes fewer in the graph
    const bool zero_copy = !as_c128 && bytes <= (size_t(64) << 10) && getenv("AMCX_NO_ZERO_COPY") == nullptr;
    amcx_ctx::SmallGraph* g = nullptr;
    for (auto& cand : c->graphs)
      if (cand.exec && cand.frames == F && cand.frame_size == N && cand.variant == v && cand.c128 == as_c128 &&
          cand.zero_copy == zero_copy &&
          cand.pin == pinned && cand.slab == dev && cand.out == c->d_out && cand.out_pin == c->out_pin && cand.slot == slot &&
          cand.ws == c->d_ws)
        g = &cand;
    if (g != nullptr) {
      ++c->graph_hits;
    } else {
      ++c->graph_misses;
      if (c->graph_misses > 64 && c->graph_misses > 4 * c->graph_hits) c->graphs_ok = false;   // shapes keep changing
      amcx_ctx::SmallGraph& slot_g = c->graphs[c->graph_next];
      hipGraph_t graph = nullptr;
      hipGraphExec_t exec = nullptr;
      int crc = AMCX_OK;
      hipError_t ce = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
      if (ce == hipSuccess && zero_copy) {
        crc = ctx_features(c, pinned, F, N, c->out_pin, v);
        const hipError_t ee = hipStreamEndCapture(c->stream, &graph);
        ce = ee;
      } else if (ce == hipSuccess) {
        ce = hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, c->stream);
        const void* d_rows = dev;
        if (ce == hipSuccess && as_c128) {
          float2* rounded = reinterpret_cast<float2*>(dev + slot);
          hipLaunchKernelGGL(amcx::amcx_c128_to_c64_kernel, dim3(2048), dim3(256), 0, c->stream,
                             reinterpret_cast<const double2*>(dev), (long long)F, (int)N, (long long)N, rounded);
          ce = hipGetLastError();
          d_rows = rounded;
        }
        if (ce == hipSuccess) crc = ctx_features(c, d_rows, F, N, c->d_out, v);
        if (ce == hipSuccess && crc == AMCX_OK)
          ce = hipMemcpyAsync(c->out_pin, c->d_out, out_bytes, hipMemcpyDeviceToHost, c->stream);
        const hipError_t ee = hipStreamEndCapture(c->stream, &graph);      // always ends the capture
        if (ce == hipSuccess) ce = ee;
      }
      if (ce == hipSuccess && crc == AMCX_OK && graph != nullptr) ce = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
      if (graph != nullptr) (void)hipGraphDestroy(graph);
      if (ce != hipSuccess || crc != AMCX_OK || exec == nullptr) {
        (void)hipGetLastError();
        c->graphs_ok = false;                 // the general path takes this call and every later one
      } else {
        if (slot_g.exec) (void)hipGraphExecDestroy(slot_g.exec);
        slot_g.exec = exec; slot_g.frames = F; slot_g.frame_size = N; slot_g.variant = v; slot_g.c128 = as_c128;
        slot_g.zero_copy = zero_copy;
        slot_g.pin = pinned; slot_g.slab = dev; slot_g.out = c->d_out; slot_g.out_pin = c->out_pin; slot_g.slot = slot;
        slot_g.ws = c->d_ws;
        c->graph_next = (c->graph_next + 1) % 4;
        g = &slot_g;
      }
    }
    if (g != nullptr) {
      st.seconds_prepare = t_loop - t_start;
      const double t_tail = wall_now();
      e = hipGraphLaunch(g->exec, c->stream);
      const hipError_t e2 = hipStreamSynchronize(c->stream);
      if (e != hipSuccess || e2 != hipSuccess) return hip_fail(e != hipSuccess ? e : e2, "amcx_ctx_features18 (graph launch)");
      if (out_row_stride == AMCX_NUM_FEATURES) {
        memcpy(out_host, c->out_pin, out_bytes);
      } else {
        for (int64_t gi = 0; gi < F; ++gi)
          memcpy(out_host + (size_t)gi * (size_t)out_row_stride, c->out_pin + (size_t)gi * AMCX_NUM_FEATURES,
                 sizeof(float) * AMCX_NUM_FEATURES);
      }
      st.pcie_bytes = (int64_t)bytes; st.chunks = 1;
      st.seconds_tail = wall_now() - t_tail;
      st.seconds = wall_now() - t_start;
      c->stats = st;
      return AMCX_OK;
    }
  }
  const int64_t units_per_slot = (int64_t)(slot / ((size_t)unit * esz));
  int64_t u = 0;
  for (int ch = 0; u < n_units && rc == AMCX_OK; ++ch) {
    // the first chunks are small so that the link starts early and staging overlaps it from the start:
    // 2, 2, 4, 4, 8, 8 ... MiB staged, up to whole slots (a 16 MB modulation of BASELINE configs[0] is five chunks)
    int64_t take = units_per_slot;
    if (ch < 12) {
      const int64_t ramp = (int64_t)((size_t(2) << 20 << (ch / 2)) / ((size_t)unit * esz));
      if (ramp < take) take = ramp;
    }
    if (take < 1) take = 1;
    if (take > n_units - u) take = n_units - u;
    const int ps = ch % kPinSlots, ds = ch & 1;
    char* pinned = c->pin + (size_t)ps * slot;
    char* dev = static_cast<char*>(c->d_slab) + (size_t)ds * dslot;
    const size_t bytes = (size_t)take * (size_t)unit * esz;
    if (ch >= kPinSlots) {                                  // the upload that last read this pinned slot is done
      const double t0 = wall_now();
      e = hipEventSynchronize(c->up_done[ps]);
      st.seconds_waiting += wall_now() - t0;
      if (e != hipSuccess) break;
    }
    {
      const double t0 = wall_now();
      // rows: run = frame g; planes: a plane is map.cnt_b runs
      const int64_t per_unit = rows ? 1 : map.cnt_b;
      amcx::stage_runs(pool, pinned, src, map, u * per_unit, (u + take) * per_unit, as_c128);
      st.seconds_staging += wall_now() - t0;
      if (io_error.load() != 0) {                          // nothing of this chunk is queued; what is in flight is drained below
        snprintf(g_hip_err, sizeof g_hip_err, "reading the container's file: %s", strerror(io_error.load()));
        rc = AMCX_EIO;
        break;
      }
    }
    if (ch >= 2) { e = hipStreamWaitEvent(c->copy_stream, c->slab_free[ds], 0); if (e != hipSuccess) break; }
    e = hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, c->copy_stream);
    if (e != hipSuccess) break;
    e = hipEventRecord(c->up_done[ps], c->copy_stream);
    if (e != hipSuccess) break;
    e = hipStreamWaitEvent(c->stream, c->up_done[ps], 0);
    if (e != hipSuccess) break;
    st.pcie_bytes += (int64_t)bytes;
    if (rows) {
      const void* d_rows = dev;
      if (as_c128) {
        float2* rounded = reinterpret_cast<float2*>(dev + slot);
        hipLaunchKernelGGL(amcx::amcx_c128_to_c64_kernel, dim3(2048), dim3(256), 0, c->stream,
                           reinterpret_cast<const double2*>(dev), (long long)take, (int)N, (long long)N, rounded);
        e = hipGetLastError();
        if (e != hipSuccess) break;
        d_rows = rounded;
      }
      rc = ctx_features(c, d_rows, take, N, c->d_out + (size_t)u * AMCX_NUM_FEATURES, v);
    } else {
      float2* frames = static_cast<float2*>(c->d_frames);
      e = as_c128 ? amcx::launch_pack_planes(reinterpret_cast<const double2*>(dev), (int)take, (long long)F, (long long)F,
                                             (int)S, (long long)K, inner_snr ? 1 : 0, frames, (long long)N, (int)u, c->stream)
                  : amcx::launch_pack_planes(reinterpret_cast<const float2*>(dev), (int)take, (long long)F, (long long)F,
                                             (int)S, (long long)K, inner_snr ? 1 : 0, frames, (long long)N, (int)u, c->stream);
      if (e != hipSuccess) break;
    }
    if (rc != AMCX_OK) break;
    e = hipEventRecord(c->slab_free[ds], c->stream);
    if (e != hipSuccess) break;
    u += take;
    st.chunks = ch + 1;
  }
  if (rc == AMCX_OK && e == hipSuccess && !rows)
    rc = ctx_features(c, c->d_frames, F, N, c->d_out, v);
  const double t_tail = wall_now();
  st.seconds_prepare = t_loop - t_start;
  // the result comes back into pinned memory (a copy into the caller's pageable rows would be staged by the
  // runtime, ~100 us for 72 KB) and is spread over the caller's row stride by the host
  if (rc == AMCX_OK && e == hipSuccess)
    e = hipMemcpyAsync(c->out_pin, c->d_out, sizeof(float) * AMCX_NUM_FEATURES * (size_t)F, hipMemcpyDeviceToHost, c->stream);
  if (rc == AMCX_OK && e != hipSuccess) rc = hip_fail(e, "amcx_ctx_features18_strided_host");
  // success or not, nothing of this call is in flight when it returns (every upload is ordered before the
  // compute stream's last kernel by an event, so on success that stream alone says so)
  hipError_t e2 = hipStreamSynchronize(c->stream);
  hipError_t e1 = (rc == AMCX_OK && e2 == hipSuccess) ? hipSuccess : hipStreamSynchronize(c->copy_stream);
  if (rc == AMCX_OK && (e1 != hipSuccess || e2 != hipSuccess))
    rc = hip_fail(e1 != hipSuccess ? e1 : e2, "amcx_ctx_features18_strided_host (sync)");
  if (rc == AMCX_OK) {
    if (out_row_stride == AMCX_NUM_FEATURES) {
      memcpy(out_host, c->out_pin, sizeof(float) * AMCX_NUM_FEATURES * (size_t)F);
    } else {
      for (int64_t g = 0; g < F; ++g)
        memcpy(out_host + (size_t)g * (size_t)out_row_stride, c->out_pin + (size_t)g * AMCX_NUM_FEATURES,
               sizeof(float) * AMCX_NUM_FEATURES);
    }
  }
  st.seconds_tail = wall_now() - t_tail;
  st.seconds = wall_now() - t_start;
  c->stats = st;
  return rc;
}

// the row-major host entries (amcx_ctx_features18_c64_host / _c128_host and their one-shot forms): a single-snr
// container whose frames are row_stride_elems apart -- the row path of the strided engine
int ctx_run(amcx_ctx* c, const void* iq_host, bool is_c128, int64_t n_frames, int32_t frame_size,
            int64_t row_stride_elems, float* out_host, int64_t out_row_stride, int32_t variant) {
  if (c == nullptr) return AMCX_EINVAL;
  if (n_frames < 0 || row_stride_elems < frame_size || out_row_stride < AMCX_NUM_FEATURES) return AMCX_EINVAL;
  amcx::Source src;
  src.re = static_cast<const char*>(iq_host);
  src.kind = is_c128 ? AMCX_SRC_C128 : AMCX_SRC_C64;
  return ctx_run_strided(c, src, 1, n_frames, frame_size, 0, row_stride_elems, 1, out_host, out_row_stride, variant);
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
  // Warm the runtime HERE, under the creating thread's own affinity mask: the first pinned allocation and the first
  // copy on a stream may start HIP / HSA helper threads, and a thread inherits its creator's mask for good.  Left to the
  // first upload they would be started inside its AffinityGuard window (ctx_run_strided narrows the calling thread to
  // the device's socket for the duration of a threaded upload) and stay on those CPUs.  Best effort: a failure here
  // surfaces where the real allocation is made.
  {
    void* warm_pin = nullptr;
    void* warm_dev = nullptr;
    if (hipHostMalloc(&warm_pin, 4096, hipHostMallocDefault) == hipSuccess && hipMalloc(&warm_dev, 4096) == hipSuccess) {
      memset(warm_pin, 0, 4096);
      if (hipMemcpyAsync(warm_dev, warm_pin, 4096, hipMemcpyHostToDevice, c->stream) == hipSuccess)
        (void)hipStreamSynchronize(c->stream);
    }
    if (warm_dev != nullptr) (void)hipFree(warm_dev);
    if (warm_pin != nullptr) (void)hipHostFree(warm_pin);
    (void)hipGetLastError();
  }
  // which CPUs are local to this device: from the kernel's PCI tree, unless AMCX_NUMA=0 (AMCX_SYSFS_ROOT: another tree)
  if (hipDeviceGetPCIBusId(c->pci_bus_id, (int)sizeof c->pci_bus_id, device) != hipSuccess) {
    (void)hipGetLastError();
    c->pci_bus_id[0] = '\0';
  }
  const char* numa_env = getenv("AMCX_NUMA");
  if (c->pci_bus_id[0] != '\0' && !(numa_env != nullptr && numa_env[0] == '0')) {
    const char* root = getenv("AMCX_SYSFS_ROOT");
    const amcx::NumaPlace place = amcx::numa_place_of(root != nullptr && root[0] != '\0' ? root : "/sys", c->pci_bus_id);
    if (!place.empty()) {
      c->numa_node = place.node;
      c->bind_cpus = place.cpus;
      c->pool.set_cpus(c->bind_cpus);
    }
  }
  *ctx_out = c;
  return AMCX_OK;
}

int amcx_ctx_bind_cpus(amcx_ctx* ctx, const int32_t* cpus, int32_t n_cpus) {
  if (ctx == nullptr || n_cpus < 0 || (n_cpus > 0 && cpus == nullptr)) return AMCX_EINVAL;
  std::vector<int> v;
  for (int32_t i = 0; i < n_cpus; ++i) {
    if (cpus[i] < 0 || cpus[i] >= CPU_SETSIZE) return AMCX_EINVAL;
    v.push_back((int)cpus[i]);
  }
  // not while an upload runs on this context: its staging threads are reading the list this call replaces
  if (ctx->in_call.load(std::memory_order_acquire) != 0) return AMCX_EINVAL;
  ctx->bind_cpus = v;
  if (v.empty()) ctx->numa_node = -1;
  ctx->pool.set_cpus(ctx->bind_cpus);
  return AMCX_OK;
}

int amcx_ctx_placement(const amcx_ctx* ctx, amcx_placement* out) {
  if (ctx == nullptr || out == nullptr) return AMCX_EINVAL;
  memset(out, 0, sizeof *out);
  out->device = ctx->device;
  out->numa_node = ctx->numa_node;
  out->n_cpus = (int32_t)ctx->bind_cpus.size();
  out->n_cpus_allowed = (int32_t)amcx::allowed_subset(ctx->bind_cpus).size();
  out->first_cpu = ctx->bind_cpus.empty() ? -1 : ctx->bind_cpus.front();
  out->last_cpu = ctx->bind_cpus.empty() ? -1 : ctx->bind_cpus.back();
  snprintf(out->pci_bus_id, sizeof out->pci_bus_id, "%s", ctx->pci_bus_id);
  return AMCX_OK;
}

int amcx_device_pci_bus_id(int32_t device, char* buf, int32_t buf_len) {
  if (buf == nullptr || buf_len < 16) return AMCX_EINVAL;
  buf[0] = '\0';
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    (void)hipGetLastError();
    return AMCX_ENODEV;
  }
  AMCX_HIP(hipDeviceGetPCIBusId(buf, buf_len, device));
  for (char* p = buf; *p; ++p) *p = (char)tolower((unsigned char)*p);
  return AMCX_OK;
}

int amcx_numa_place(const char* sysfs_root, const char* pci_bus_id, int32_t* node_out, int32_t* cpus_out,
                    int32_t cpus_cap, int32_t* n_cpus_out) {
  if (pci_bus_id == nullptr || node_out == nullptr || n_cpus_out == nullptr || cpus_cap < 0 ||
      (cpus_cap > 0 && cpus_out == nullptr))
    return AMCX_EINVAL;
  const amcx::NumaPlace place = amcx::numa_place_of(sysfs_root != nullptr && sysfs_root[0] != '\0' ? sysfs_root : "/sys", pci_bus_id);
  *node_out = place.empty() ? -1 : place.node;
  *n_cpus_out = place.empty() ? 0 : (int32_t)place.cpus.size();
  for (int32_t i = 0; i < *n_cpus_out && i < cpus_cap; ++i) cpus_out[i] = place.cpus[(size_t)i];
  return AMCX_OK;
}

int amcx_ctx_destroy(amcx_ctx* c) {
  if (c == nullptr) return AMCX_OK;
  DeviceGuard guard;
  (void)guard.enter(c->device);
  if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
  if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
  for (auto ev : c->up_done) if (ev) (void)hipEventDestroy(ev);
  for (auto ev : c->slab_free) if (ev) (void)hipEventDestroy(ev);
  if (c->d_out) (void)hipFree(c->d_out);
  if (c->d_slab) (void)hipFree(c->d_slab);
  if (c->d_frames) (void)hipFree(c->d_frames);
  if (c->d_ws) (void)hipFree(c->d_ws);
  if (c->pin) (void)hipHostFree(c->pin);
  if (c->out_pin) (void)hipHostFree(c->out_pin);
  for (auto& g : c->graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
  delete c;                                     // joins the staging threads
  return AMCX_OK;
}

int amcx_ctx_features18_strided_host(amcx_ctx* ctx, const void* re, const void* im, int32_t kind,
                                     int64_t n_snr, int64_t n_frames, int32_t frame_size,
                                     int64_t stride_snr, int64_t stride_frame, int64_t stride_sample,
                                     float* out_host, int64_t out_row_stride, int32_t variant) {
  amcx::Source src;
  src.re = static_cast<const char*>(re);
  src.im = static_cast<const char*>(im);
  src.kind = kind;
  return ctx_run_strided(ctx, src, n_snr, n_frames, frame_size, stride_snr, stride_frame, stride_sample,
                         out_host, out_row_stride, variant);
}

int amcx_ctx_features18_strided_file(amcx_ctx* ctx, int32_t fd, int64_t re_offset, int64_t im_offset, int32_t kind,
                                     int64_t n_snr, int64_t n_frames, int32_t frame_size,
                                     int64_t stride_snr, int64_t stride_frame, int64_t stride_sample,
                                     float* out_host, int64_t out_row_stride, int32_t variant) {
  if (fd < 0 || re_offset < 0) return AMCX_EINVAL;
  amcx::Source src;
  src.fd = fd;
  src.re_off = re_offset;
  src.im_off = im_offset;
  src.kind = kind;
  return ctx_run_strided(ctx, src, n_snr, n_frames, frame_size, stride_snr, stride_frame, stride_sample,
                         out_host, out_row_stride, variant);
}

static int stage_any(amcx::Source src, int64_t n_snr, int64_t n_frames,
                     int32_t frame_size, int64_t stride_snr, int64_t stride_frame, int64_t stride_sample,
                     int64_t first_unit, int64_t n_units, void* dst, int64_t dst_bytes, int32_t threads,
                     int32_t* plane_major, int32_t* inner_snr_out) {
  const int32_t kind = src.kind;
  if (n_snr < 0 || n_frames < 0 || stride_snr < 0 || stride_frame < 0 || stride_sample < 0 || first_unit < 0 ||
      n_units < 0 || threads < 0 || threads > 256 || kind < AMCX_SRC_C64 || kind > AMCX_SRC_F64_SPLIT ||
      frame_size < AMCX_MIN_FRAME_SIZE || frame_size > AMCX_MAX_FRAME_SIZE)
    return AMCX_EINVAL;
  if (n_frames > 0 && n_snr > (int64_t(1) << 40) / n_frames) return AMCX_EINVAL;
  bool rows = false, inner_snr = false;
  amcx::RunMap map;
  if (!classify_layout(n_snr, n_frames, frame_size, stride_snr, stride_frame, stride_sample, &rows, &inner_snr, &map))
    return AMCX_ENOTSUP;
  if (plane_major) *plane_major = rows ? 0 : 1;
  if (inner_snr_out) *inner_snr_out = inner_snr ? 1 : 0;
  const int64_t F = n_snr * n_frames, unit = rows ? frame_size : F, total_units = rows ? F : frame_size;
  if (first_unit + n_units > total_units) return AMCX_EINVAL;
  if (n_units == 0 || unit == 0) return AMCX_OK;
  if ((src.fd < 0 && src.re == nullptr) || dst == nullptr || dst_bytes < n_units * unit * 8) return AMCX_EINVAL;
  if (kind < AMCX_SRC_F32_SPLIT) { src.im = nullptr; src.im_off = -1; }
  std::atomic<int> io_error{0};
  src.io_error = &io_error;
  amcx::Pool pool;
  pool.resize(threads < 1 ? 1 : threads);
  const int64_t per_unit = rows ? 1 : map.cnt_b;
  amcx::stage_runs(pool, static_cast<char*>(dst), src, map, first_unit * per_unit, (first_unit + n_units) * per_unit, false);
  if (io_error.load() != 0) {
    snprintf(g_hip_err, sizeof g_hip_err, "reading the container's file: %s", strerror(io_error.load()));
    return AMCX_EIO;
  }
  return AMCX_OK;
}

int amcx_stage_host(const void* re, const void* im, int32_t kind, int64_t n_snr, int64_t n_frames,
                    int32_t frame_size, int64_t stride_snr, int64_t stride_frame, int64_t stride_sample,
                    int64_t first_unit, int64_t n_units, void* dst, int64_t dst_bytes, int32_t threads,
                    int32_t* plane_major, int32_t* inner_snr_out) {
  amcx::Source src;
  src.re = static_cast<const char*>(re);
  src.im = static_cast<const char*>(im);
  src.kind = kind;
  return stage_any(src, n_snr, n_frames, frame_size, stride_snr, stride_frame, stride_sample, first_unit, n_units, dst,
                   dst_bytes, threads, plane_major, inner_snr_out);
}

int amcx_stage_file(int32_t fd, int64_t re_offset, int64_t im_offset, int32_t kind, int64_t n_snr, int64_t n_frames,
                    int32_t frame_size, int64_t stride_snr, int64_t stride_frame, int64_t stride_sample,
                    int64_t first_unit, int64_t n_units, void* dst, int64_t dst_bytes, int32_t threads,
                    int32_t* plane_major, int32_t* inner_snr_out) {
  if (fd < 0 || re_offset < 0) return AMCX_EINVAL;
  amcx::Source src;
  src.fd = fd;
  src.re_off = re_offset;
  src.im_off = im_offset;
  src.kind = kind;
  return stage_any(src, n_snr, n_frames, frame_size, stride_snr, stride_frame, stride_sample, first_unit, n_units, dst,
                   dst_bytes, threads, plane_major, inner_snr_out);
}

int amcx_ctx_configure(amcx_ctx* ctx, int32_t threads, int64_t slot_bytes, int32_t round_on_device) {
  if (ctx == nullptr || threads < 0 || threads > 256 || slot_bytes < 0) return AMCX_EINVAL;
  if (threads > 0) ctx->threads = threads;
  if (slot_bytes > 0) ctx->slot_bytes = (size_t)slot_bytes < 4096 ? 4096 : (size_t)slot_bytes;
  if (round_on_device >= 0) ctx->round_on_device = round_on_device != 0;
  return AMCX_OK;
}

int amcx_ctx_upload_stats(const amcx_ctx* ctx, amcx_upload_stats* out) {
  if (ctx == nullptr || out == nullptr) return AMCX_EINVAL;
  *out = ctx->stats;
  return AMCX_OK;
}

int amcx_pack_planes_c64(const void* slab_dev, int32_t src_kind, int32_t n_planes, int64_t plane_stride,
                         int64_t n_snr, int64_t n_frames, int32_t inner_snr, void* frames_dev,
                         int64_t row_stride_elems, int32_t n0, void* hip_stream) {
  if (n_planes < 0 || n_snr < 0 || n_frames < 0 || n0 < 0 || (src_kind != AMCX_SRC_C64 && src_kind != AMCX_SRC_C128))
    return AMCX_EINVAL;
  if (n_snr > 0x7fffffffLL || (n_frames > 0 && n_snr > (int64_t(1) << 40) / n_frames)) return AMCX_EINVAL;
  const int64_t P = n_snr * n_frames;
  if (plane_stride < P || row_stride_elems < (int64_t)n0 + n_planes) return AMCX_EINVAL;
  if (n_planes == 0 || P == 0) return AMCX_OK;
  if (slab_dev == nullptr || frames_dev == nullptr) return AMCX_EINVAL;
  if (on_another_device(slab_dev) || on_another_device(frames_dev)) return AMCX_EINVAL;
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  float2* dst = static_cast<float2*>(frames_dev);
  hipError_t e = src_kind == AMCX_SRC_C128
      ? amcx::launch_pack_planes(static_cast<const double2*>(slab_dev), n_planes, P, plane_stride, (int)n_snr, n_frames,
                                 inner_snr, dst, row_stride_elems, n0, stream)
      : amcx::launch_pack_planes(static_cast<const float2*>(slab_dev), n_planes, P, plane_stride, (int)n_snr, n_frames,
                                 inner_snr, dst, row_stride_elems, n0, stream);
  if (e != hipSuccess) return hip_fail(e, "amcx_pack_planes_c64");
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
                     : frame_size > amcx::kBlockMaxN                   ? "amcx_features18_stream_kernel"     /* <true> with a workspace, <false> without */
                     : block_mode(frame_size) == amcx::kBlockPow2      ? "amcx_features18_block_kernel<1>"
                     : block_mode(frame_size) == amcx::kBlockBluestein ? "amcx_features18_block_kernel<2>"
                     : block_mode(frame_size) == amcx::kBlockBluesteinBig ? "amcx_features18_block_kernel<3>"
                                                                       : "amcx_features18_block_kernel<0>";
  snprintf(buf, (size_t)buf_len, "%s", name);
  return AMCX_OK;
}

int amcx_probe_fma_rate(double seconds, void* hip_stream, double* wave_instr_per_s, double* clock_ghz) {
  if (!(seconds > 0.0) || seconds > 60.0 || wave_instr_per_s == nullptr) return AMCX_EINVAL;
  *wave_instr_per_s = 0.0;
  if (clock_ghz) *clock_ghz = 0.0;
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  const int grid = cu_count();
  if (grid <= 0) return AMCX_ENODEV;
  const long long n_waves = (long long)grid * 16;
  constexpr int kIters = 32768;                            // x 32 instructions x 4096 waves: ~5 ms a launch
  float* sink = nullptr;
  unsigned long long* ticks = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
  int rc = AMCX_OK;
  auto fail = [&](hipError_t e, const char* what) { rc = hip_fail(e, what); };
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&sink), 4);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ticks), (size_t)n_waves * 16);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  if (e == hipSuccess) e = hipEventCreate(&e2);
  if (e != hipSuccess) {
    fail(e, "fma probe setup");
  } else {
    auto launch = [&]() { hipLaunchKernelGGL(amcx_probe_fma_kernel, dim3((unsigned)grid), dim3(1024), 0, stream, kIters, sink, ticks); };
    // one launch to learn its length, then `seconds` of back-to-back launches: the first half lets the board's power
    // management settle the clock, the second half is timed
    (void)hipEventRecord(e0, stream);
    launch();
    (void)hipEventRecord(e1, stream);
    e = hipEventSynchronize(e1);
    float one_ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&one_ms, e0, e1);
    if (e != hipSuccess) {
      fail(e, "fma probe launch");
    } else {
      if (!(one_ms > 0.01f)) one_ms = 0.01f;
      long long n = (long long)(seconds * 1e3 / 2.0 / one_ms);
      if (n < 1) n = 1;
      if (n > 100000) n = 100000;
      for (long long i = 0; i < n; ++i) launch();
      (void)hipEventRecord(e1, stream);
      for (long long i = 0; i < n; ++i) launch();
      (void)hipEventRecord(e2, stream);
      e = hipEventSynchronize(e2);
      float ms = 0.f;
      if (e == hipSuccess) e = hipEventElapsedTime(&ms, e1, e2);
      if (e == hipSuccess) e = hipGetLastError();
      if (e != hipSuccess || !(ms > 0.f)) {
        fail(e, "fma probe timing");
      } else {
        *wave_instr_per_s = (double)n * (double)n_waves * (double)kIters * 32.0 / ((double)ms * 1e-3);
        if (clock_ghz) {
          std::vector<unsigned long long> h((size_t)n_waves * 2);
          e = hipMemcpy(h.data(), ticks, h.size() * 8, hipMemcpyDeviceToHost);
          if (e == hipSuccess) {
            double cyc = 0.0, real = 0.0;
            for (long long w = 0; w < n_waves; ++w) { cyc += (double)h[(size_t)(2 * w)]; real += (double)h[(size_t)(2 * w + 1)]; }
            if (real > 0.0) *clock_ghz = cyc / (real * 10.0);       // s_memrealtime ticks at 100 MHz
          } else {
            fail(e, "fma probe read-back");
          }
        }
      }
    }
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (e2) (void)hipEventDestroy(e2);
  if (sink) (void)hipFree(sink);
  if (ticks) (void)hipFree(ticks);
  return rc;
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

// how a statistics call is cut: chunks per group and rows per chunk (whole tiles)
static void stats_plan(int64_t n_groups, int64_t rows_per_group, int n_cols, int64_t* chunks, int64_t* rows_per_chunk) {
  const int64_t tile = amcx::stat_tile_rows(n_cols);
  const int64_t tiles = (rows_per_group + tile - 1) / tile;
  const int64_t target = 1024;                               // workgroups wanted in all (4 per CU, one resident round); the same on any device
  int64_t want = (target + n_groups - 1) / n_groups;
  if (want > tiles) want = tiles;
  if (want < 1) want = 1;
  const int64_t tiles_per_chunk = (tiles + want - 1) / want;
  *rows_per_chunk = tiles_per_chunk * tile;
  *chunks = (tiles + tiles_per_chunk - 1) / tiles_per_chunk;
}

static bool stats_args_ok(int64_t n_groups, int64_t rows_per_group, int64_t row_stride, int32_t n_cols) {
  return n_groups >= 0 && rows_per_group >= 1 && n_cols >= 1 && n_cols <= amcx::kStatMaxCols &&
         row_stride >= n_cols && row_stride <= (1 << 20) && n_groups <= 0x7fffffffLL;
}

int64_t amcx_group_stats_workspace_bytes(int64_t n_groups, int64_t rows_per_group, int32_t n_cols) {
  if (!stats_args_ok(n_groups, rows_per_group, n_cols, n_cols)) return -1;
  if (n_groups == 0) return 0;
  int64_t chunks, rpc;
  stats_plan(n_groups, rows_per_group, n_cols, &chunks, &rpc);
  return n_groups * chunks * n_cols * 3 * (int64_t)sizeof(double);
}

int amcx_group_stats_ws_f32(const float* x_dev, int64_t n_groups, int64_t rows_per_group,
                            int64_t row_stride, int32_t n_cols, double* mean_dev, double* std_dev,
                            void* workspace_dev, int64_t workspace_bytes, void* hip_stream) {
  if (!stats_args_ok(n_groups, rows_per_group, row_stride, n_cols)) return AMCX_EINVAL;
  if (n_groups == 0) return AMCX_OK;
  if (x_dev == nullptr || mean_dev == nullptr || std_dev == nullptr || workspace_dev == nullptr) return AMCX_EINVAL;
  if (on_another_device(x_dev) || on_another_device(workspace_dev)) return AMCX_EINVAL;
  int64_t chunks, rpc;
  stats_plan(n_groups, rows_per_group, n_cols, &chunks, &rpc);
  if (workspace_bytes < n_groups * chunks * n_cols * 3 * (int64_t)sizeof(double)) return AMCX_EINVAL;
  if (n_groups * chunks > 0x7fffffffLL) return AMCX_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  double* part = static_cast<double*>(workspace_dev);
  hipLaunchKernelGGL(amcx::amcx_stats_part_kernel, dim3((unsigned)(n_groups * chunks)),
                     dim3(amcx::kStatThreads), 0, st, x_dev, (long long)rows_per_group,
                     (long long)row_stride, (int)n_cols, (long long)rpc, (int)chunks, part);
  // the pooling launch: one small workgroup per (column, group); groups beyond the grid's y limit go in slices
  const unsigned pool_threads = chunks > 128 ? 256u : chunks > 64 ? 128u : 64u;
  for (int64_t g0 = 0; g0 < n_groups; g0 += 65535) {
    const int64_t ng = n_groups - g0 < 65535 ? n_groups - g0 : 65535;
    hipLaunchKernelGGL(amcx::amcx_stats_combine_kernel, dim3((unsigned)n_cols, (unsigned)ng), dim3(pool_threads), 0, st,
                       part + g0 * n_cols * 3 * chunks, (int)chunks, (int)n_cols, mean_dev + g0 * n_cols,
                       std_dev + g0 * n_cols);
  }
  AMCX_HIP(hipGetLastError());
  return AMCX_OK;
}

int amcx_group_stats_f32(const float* x_dev, int64_t n_groups, int64_t rows_per_group,
                         int64_t row_stride, int32_t n_cols, double* mean_dev, double* std_dev,
                         void* hip_stream) {
  if (!stats_args_ok(n_groups, rows_per_group, row_stride, n_cols)) return AMCX_EINVAL;
  if (n_groups == 0) return AMCX_OK;
  if (x_dev == nullptr || mean_dev == nullptr || std_dev == nullptr) return AMCX_EINVAL;
  const int64_t bytes = amcx_group_stats_workspace_bytes(n_groups, rows_per_group, n_cols);
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  void* ws = nullptr;
  AMCX_HIP(hipMallocAsync(&ws, (size_t)bytes, st));           // stream-ordered: no synchronisation
  const int rc = amcx_group_stats_ws_f32(x_dev, n_groups, rows_per_group, row_stride, n_cols, mean_dev,
                                         std_dev, ws, bytes, hip_stream);
  const hipError_t fe = hipFreeAsync(ws, st);
  if (rc != AMCX_OK) return rc;
  AMCX_HIP(fe);
  return AMCX_OK;
}

int64_t amcx_standardize_workspace_bytes(int64_t n_rows, int32_t n_cols) {
  if (n_rows < 1) return n_rows == 0 ? 0 : -1;
  const int64_t part = amcx_group_stats_workspace_bytes(1, n_rows, n_cols);
  return part < 0 ? -1 : part + 2 * (int64_t)amcx::kStatMaxCols * (int64_t)sizeof(double);
}

int amcx_standardize_fit_transform_f32(const float* x_dev, int64_t n_rows, int64_t row_stride,
                                       int32_t n_cols, const int32_t* cols_host, int32_t n_sel,
                                       float* out_dev, int64_t out_stride, double* mean_dev,
                                       double* scale_dev, void* workspace_dev, int64_t workspace_bytes,
                                       void* hip_stream) {
  if (n_rows < 0 || n_sel < 1 || n_sel > amcx::kStatMaxCols || out_stride < n_sel || cols_host == nullptr ||
      n_cols < 1 || n_cols > amcx::kStatMaxCols || row_stride < n_cols)
    return AMCX_EINVAL;
  amcx::SelectCols sel;
  sel.n = n_sel;
  for (int j = 0; j < amcx::kStatMaxCols; ++j) sel.c[j] = 0;
  for (int j = 0; j < n_sel; ++j) {
    if (cols_host[j] < 0 || cols_host[j] >= n_cols) return AMCX_EINVAL;
    sel.c[j] = cols_host[j];
  }
  if (n_rows == 0) return AMCX_OK;
  if (!x_dev || !out_dev || !mean_dev || !scale_dev || !workspace_dev) return AMCX_EINVAL;
  if (on_another_device(x_dev) || on_another_device(out_dev)) return AMCX_EINVAL;
  const int64_t need = amcx_standardize_workspace_bytes(n_rows, n_cols);
  if (need < 0 || workspace_bytes < need) return AMCX_EINVAL;
  double* all_mean = static_cast<double*>(workspace_dev);
  double* all_std = all_mean + amcx::kStatMaxCols;
  void* part = all_std + amcx::kStatMaxCols;
  const int rc = amcx_group_stats_ws_f32(x_dev, 1, n_rows, row_stride, n_cols, all_mean, all_std, part,
                                         need - 2 * (int64_t)amcx::kStatMaxCols * (int64_t)sizeof(double), hip_stream);
  if (rc != AMCX_OK) return rc;
  const int64_t grid = (n_rows + amcx::kSelectRows - 1) / amcx::kSelectRows;
  if (grid > 0x7fffffffLL) return AMCX_EINVAL;
  hipLaunchKernelGGL(amcx::amcx_select_fit_scale_kernel, dim3((unsigned)grid), dim3(amcx::kBlockThreads), 0,
                     static_cast<hipStream_t>(hip_stream), x_dev, (long long)n_rows, (long long)row_stride, sel,
                     all_mean, all_std, out_dev, (long long)out_stride, mean_dev, scale_dev);
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
