// Microbenchmark: the 16-point twisted decimation-in-time pass of the register FFT (32 butterflies) in its scalar
// form (6 v_fma_f32 per butterfly) and its packed form (3 v_pk_fma_f32), data and factors in registers, nothing else
// in the loop but a rescale.  Cycles per pass per SIMD (s_memtime) and chip-wide passes/s at 1 / 3 / 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-math-errno -fno-slp-vectorize tools/ubench_fft.hip -o tools/ubench_fft
#include "../amcpy_amd/csrc/amcx_wave_kernel.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
using namespace amcx::wave;
constexpr int ITERS = 4000;

template <int KIND>
__global__ void k(const float2* tw_in, float* out, unsigned long long* cyc) {
  float2 tw[15];
  for (int i = 0; i < 15; ++i) tw[i] = tw_in[i * 64 + (threadIdx.x & 63)];
  unsigned long long t0 = 0;
  float acc = 0.f;
  if constexpr (KIND == 0) {
    float re[16], im[16];
    for (int i = 0; i < 16; ++i) { re[i] = 0.01f * (threadIdx.x + i); im[i] = 0.02f * (threadIdx.x - i); }
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
      twisted_dit<16>(re, im, [&](auto ii) { return tw[decltype(ii)::value]; });
      static_for<16>([&](auto ii) { re[decltype(ii)::value] *= 0.0625f; im[decltype(ii)::value] *= 0.0625f; });
    }
    for (int i = 0; i < 16; ++i) acc += re[i] + im[i];
  } else {
    v2 x[16];
    for (int i = 0; i < 16; ++i) x[i] = (v2){0.01f * (threadIdx.x + i), 0.02f * (threadIdx.x - i)};
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
      twisted_dit_pk<16>(x, [&](auto ii) { return (v2){tw[decltype(ii)::value].x, tw[decltype(ii)::value].y}; });
      static_for<16>([&](auto ii) { x[decltype(ii)::value] *= (v2){0.0625f, 0.0625f}; });
    }
    for (int i = 0; i < 16; ++i) acc += x[i].x + x[i].y;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
void run(const char* label, int waves_per_simd, const float2* d_tw, float* d_out, unsigned long long* d_cyc) {
  const int threads = 64 * 4 * waves_per_simd, grid = 256;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int r = 0; r < 40; ++r) hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(threads), 0, 0, d_tw, d_out, d_cyc);   // ~1.5 s: let the clock settle
  CHECK(hipEventRecord(e0));
  const int reps = 10;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(threads), 0, 0, d_tw, d_out, d_cyc);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  std::vector<unsigned long long> h(grid * threads / 64);
  std::vector<float> o(grid * threads);
  CHECK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(o.data(), d_out, o.size() * 4, hipMemcpyDeviceToHost));
  double c = 0; for (auto v : h) c += (double)v; c /= h.size();
  double sum = 0; for (auto v : o) sum += v;
  const double passes = (double)grid * (threads / 64) * ITERS;
  printf("%-34s waves/SIMD=%d  cycles per pass per SIMD = %7.1f (per wave %7.1f)  chip %.2f G passes/s  (%.2f ms)  checksum %.6g\n", label,
         waves_per_simd, c / ITERS / waves_per_simd * 1.0, c / ITERS, passes / (ms * 1e-3) / 1e9, ms, sum);
}

int main() {
  float2* d_tw; float* d_out; unsigned long long* d_cyc;
  std::vector<float2> tw(15 * 64);
  for (int i = 0; i < 15 * 64; ++i) { const float a = 0.37f * i; tw[i] = make_float2(cosf(a), -sinf(a)); }
  CHECK(hipMalloc(&d_tw, tw.size() * 8)); CHECK(hipMemcpy(d_tw, tw.data(), tw.size() * 8, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&d_out, 256 * 1024 * 4)); CHECK(hipMalloc(&d_cyc, 256 * 16 * 8));
  for (int w : {1, 3, 4}) {
    run<0>("scalar: 32 x 6 v_fma_f32 + 32 mul", w, d_tw, d_out, d_cyc);
    run<1>("packed: 32 x 3 v_pk_fma_f32 + 16 mul", w, d_tw, d_out, d_cyc);
  }
  return 0;
}
