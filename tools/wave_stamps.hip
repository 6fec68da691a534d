// Diagnostic: where does a wave of amcx_features18_wave_kernel spend its cycles?
// Builds the kernel with in-kernel s_memtime stamps (shares of each section; the
// stamped build's run time itself is not a benchmark).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize tools/wave_stamps.hip -o tools/wave_stamps
#define AMCX_WAVE_STAMPS 1
#include "../amcpy_amd/csrc/amcx_wave_kernel.h"
#include <stdio.h>
#include <vector>
#include <random>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <int N> void run(const float2* d_iq, long long F, float* d_out, unsigned long long* d_st, int grid) {
  using namespace amcx::wave;
  auto kern = amcx_features18_wave_kernel<N>;
  constexpr int kLdsBytes = Cfg<N>::kLdsBytes;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(Cfg<N>::kThreads), kLdsBytes, 0, d_iq, F, (long long)N, d_out, 18LL, d_st);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  }
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  constexpr int kWavesPerWG = Cfg<N>::kWavesPerWG;
  int nw = grid * kWavesPerWG;
  std::vector<unsigned long long> h(nw * kStampSections);
  CHECK(hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost));
  double s[kStampSections] = {0};
  for (int w = 0; w < nw; ++w) for (int k = 0; k < kStampSections; ++k) s[k] += (double)h[w * kStampSections + k];
  // stamp k closes the interval that ENDS at AMCX_STAMP(k) in amcx_wave_kernel.h: 7 = top of a frame (so it holds the
  // previous frame's FFT and the loop / load issue), 0 = end of the statistics sweep (incl. the wait for the frame's loads),
  // 3 = end of the envelope sweep, 1 = end of the wave reduction, 4 = after the batch's last frame (its FFT), 5 = finalisation
  const char* names[kStampSections] = {"load wait + statistics sweep", "wave reduction of the sums", "-", "envelope second sweep",
                                       "FFT of the batch's last frame", "fp64 finalisation + store", "-",
                                       "FFT of the other frames + loop + load issue"};
  double real_us = s[6] / nw / 100.0; s[6] = 0;
  { double mn = 1e30, mx = 0; for (int w = 0; w < nw; ++w) { double v = (double)h[w * kStampSections + 6] / 100.0; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
    printf("   wave lifetime us: min %.1f  mean %.1f  max %.1f   (kernel %.1f)\n", mn, real_us, mx, ms * 1e3);
    double xs[8] = {0}; int xn[8] = {0};
    for (int w = 0; w < nw; ++w) { int wg = w / kWavesPerWG; xs[wg % 8] += (double)h[w * kStampSections + 6] / 100.0; xn[wg % 8]++; }
    printf("   mean lifetime by blockIdx %% 8:"); for (int x = 0; x < 8; ++x) printf(" %.0f", xs[x] / xn[x]); printf("\n");
    double lo = 1e30, hi = 0; for (int wg = 0; wg < nw / kWavesPerWG; ++wg) { double m = 0; for (int k = 0; k < kWavesPerWG; ++k) m += (double)h[(wg * kWavesPerWG + k) * kStampSections + 6] / 100.0; m /= kWavesPerWG; lo = m < lo ? m : lo; hi = m > hi ? m : hi; }
    printf("   per-workgroup mean lifetime: min %.0f max %.0f\n", lo, hi); }
  double tot = 0; for (double v : s) tot += v;
  double frames_per_wave = (double)F / nw;
  printf("N=%d  kernel %.3f ms  (%.1f M frames/s)  mean wave lifetime %.1f us -> s_memtime clock %.2f GHz; cycles per frame per wave: %.0f\n", N, ms, F / ms / 1e3, real_us, tot / nw / real_us / 1e3, tot / nw / frames_per_wave);
  for (int k = 0; k < kStampSections; ++k) if (s[k] > 0) printf("   %-44s %6.1f %%   %8.0f cycles/frame\n", names[k], 100.0 * s[k] / tot, s[k] / nw / frames_per_wave);
}
int main() {
  const long long F = 6 * 26 * 4096 / 2;   // 319488 frames, 5.2 GB
  std::vector<float2> h(F * 2048);
  std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
  for (long long i = 0; i < 2048LL * 4096; ++i) h[i] = make_float2(nd(rng), nd(rng));
  for (long long i = 2048LL * 4096; i < F * 2048; ++i) h[i] = h[i % (2048LL * 4096)];
  float2* d_iq; float* d_out; unsigned long long* d_st;
  CHECK(hipMalloc(&d_iq, F * 2048 * 8)); CHECK(hipMalloc(&d_out, 2 * F * 18 * 4)); CHECK(hipMalloc(&d_st, 256 * 16 * 8 * 8));
  CHECK(hipMemcpy(d_iq, h.data(), F * 2048 * 8, hipMemcpyHostToDevice));
  run<2048>(d_iq, F, d_out, d_st, 256);
  run<2048>(d_iq, F, d_out, d_st, 128);   // half the CUs: does the clock rise (power-bound)?
  run<2048>(d_iq, F, d_out, d_st, 64);
  run<1024>(d_iq, F * 2, d_out, d_st, 256);
  run<4096>(d_iq, F / 2, d_out, d_st, 256);
  return 0;
}
