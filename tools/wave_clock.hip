// Diagnostic: the in-kernel clock of amcx_features18_wave_kernel<N>, measured the way
// MI355X_MICROARCH.md ("DVFS give-back", item 6) prescribes: the product instruction stream
// with ONE s_memtime / s_memrealtime pair around the whole frame loop (AMCX_WAVE_STAMPS=2),
// >= 2 s of back-to-back launches on random data, median over waves of the last launch.
// Also: the same on all-zero data (is the chip holding its clock down under load?), on half
// and a quarter of the CUs, and -- built with -DAMCX_ABL_NOFFT / -DAMCX_ABL_NOSTATS -- with
// one half of the frame's work removed (where do cycles and clock go?).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-math-errno -fno-slp-vectorize \
//         [-DAMCX_ABL_NOFFT | -DAMCX_ABL_NOSTATS] tools/wave_clock.hip -o tools/wave_clock
//   tools/wave_clock [frame_size=2048] [valu_instr_per_frame=0] [only_config=-1] [seconds=2.5]
#define AMCX_WAVE_STAMPS 2
#include "../amcpy_amd/csrc/amcx_wave_kernel.h"
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <chrono>
#include <random>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

static double g_valu_per_frame = 0;

template <int N>
void run(const char* label, const float2* d_iq, long long F, float* d_out, unsigned long long* d_st, int grid,
         double seconds, long long row_stride = N) {
  using namespace amcx::wave;
  auto kern = amcx_features18_wave_kernel<N>;
  constexpr int kLdsBytes = Cfg<N>::kLdsBytes;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
  auto launch = [&]() {
    hipLaunchKernelGGL(kern, dim3(grid), dim3(Cfg<N>::kThreads), kLdsBytes, 0, d_iq, F, row_stride, d_out, 18LL, d_st);
  };
  // >= `seconds` of back-to-back launches
  const auto t0 = std::chrono::steady_clock::now();
  long launches = 0;
  for (;;) {
    for (int k = 0; k < 16; ++k) launch();
    launches += 16;
    CHECK(hipDeviceSynchronize());
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() >= seconds) break;
  }
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int reps = 20;
  CHECK(hipEventRecord(e0));
  for (int k = 0; k < reps; ++k) launch();
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  constexpr int kWavesPerWG = Cfg<N>::kWavesPerWG;
  const int nw = grid * kWavesPerWG;
  std::vector<unsigned long long> h((size_t)nw * kStampSections);
  CHECK(hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> clk(nw), life(nw), cyc(nw);
  for (int w = 0; w < nw; ++w) {
    cyc[w] = (double)h[(size_t)w * kStampSections + 0];
    life[w] = (double)h[(size_t)w * kStampSections + 6] / 100.0;           // us
    clk[w] = cyc[w] / (life[w] * 1e3);                                     // GHz
  }
  auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  auto mn = [](const std::vector<double>& v) { return *std::min_element(v.begin(), v.end()); };
  auto mx = [](const std::vector<double>& v) { return *std::max_element(v.begin(), v.end()); };
  const double clock = med(clk);
  const double frames_per_simd = (double)F / (grid * 4.0);
  const double cyc_per_frame_simd = med(cyc) / frames_per_simd;     // SIMD cycles available per frame
  printf("%-28s N=%d grid=%d  %ld launches warm  kernel %.3f ms  %.1f M frames/s | in-kernel clock %.3f GHz "
         "(min %.3f max %.3f) | wave lifetime us med %.0f min %.0f max %.0f | SIMD cycles/frame %.0f",
         label, N, grid, launches, ms, F / ms / 1e3, clock, mn(clk), mx(clk), med(life), mn(life), mx(life),
         cyc_per_frame_simd);
  {  // FNV-1a over the output rows (sums + features): two builds that claim bit-identical arithmetic print the same word
    std::vector<unsigned> o((size_t)F * 18);
    CHECK(hipMemcpy(o.data(), d_out, o.size() * 4, hipMemcpyDeviceToHost));
    unsigned long long hsh = 1469598103934665603ull;
    for (unsigned w : o) { hsh ^= w; hsh *= 1099511628211ull; }
    printf(" | out %016llx", hsh);
  }
  {  // when does each WORKGROUP finish (its slowest wave)?  The kernel lasts as long as the slowest workgroup.  The eight XCDs
     // run equal slices up to 8 % apart; a pool of frames shared by all workgroups at the end of the launch levels them to
     // +-3 us and gains 0.7 %, because the power cap hands a finished XCD's budget to the others anyway
     // (profiles/r4_xcd_pool_ab.txt, tools/experiments/r4_shared_pool.patch)
    std::vector<double> wg(grid);
    for (int g = 0; g < grid; ++g) {
      double m = 0;
      for (int w = 0; w < kWavesPerWG; ++w) m = std::max(m, life[(size_t)g * kWavesPerWG + w]);
      wg[g] = m;
    }
    std::vector<double> srt = wg;
    std::sort(srt.begin(), srt.end());
    double mean = 0;
    for (double v : wg) mean += v;
    mean /= grid;
    printf(" | workgroup finish us min %.0f p10 %.0f med %.0f mean %.0f p90 %.0f max %.0f", srt[0], srt[grid / 10], srt[grid / 2], mean,
           srt[grid * 9 / 10], srt[grid - 1]);
    // by XCD (workgroup g runs on XCD g % 8 when the grid is dispatched round-robin)
    printf(" | mean by g%%8:");
    for (int x = 0; x < 8; ++x) {
      double a = 0; int n = 0;
      for (int g = x; g < grid; g += 8) { a += wg[g]; ++n; }
      printf(" %.0f", n ? a / n : 0.0);
    }
  }
  if (g_valu_per_frame > 0)
    printf(" | VALU issue slots used %.1f %% (%.0f instr x 2 cyc)", 100.0 * g_valu_per_frame * 2.0 / cyc_per_frame_simd,
           g_valu_per_frame);
  printf("\n");
  fflush(stdout);
}

static int g_only = -1;        // run only this configuration (index below), all if < 0
static double g_seconds = 2.5;

template <int N>
void all(long long F) {
  const size_t n_samp = (size_t)F * N;
  float2 *d_iq, *d_zero; float* d_out; unsigned long long* d_st;
  CHECK(hipMalloc(&d_iq, n_samp * 8)); CHECK(hipMalloc(&d_zero, n_samp * 8));
  CHECK(hipMalloc(&d_out, (size_t)F * 18 * 4)); CHECK(hipMalloc(&d_st, 256 * 16 * amcx::wave::kStampSections * 8));
  {  // 8 Mi random samples, tiled
    const size_t tile = (size_t)1 << 23;
    std::vector<float2> h(tile);
    std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 0.7071f);
    for (auto& v : h) v = make_float2(nd(rng), nd(rng));
    for (size_t o = 0; o < n_samp; o += tile)
      CHECK(hipMemcpy(d_iq + o, h.data(), std::min(tile, n_samp - o) * 8, hipMemcpyHostToDevice));
  }
  CHECK(hipMemset(d_zero, 0, n_samp * 8));
  const double T = g_seconds;
  auto want = [](int k) { return g_only < 0 || g_only == k; };
  if (want(0)) run<N>("random, all CUs", d_iq, F, d_out, d_st, 256, T);
  if (want(1)) run<N>("zeros,  all CUs", d_zero, F, d_out, d_st, 256, T);
  if (want(2)) run<N>("random, all CUs (again)", d_iq, F, d_out, d_st, 256, T);
  // row stride 0: every frame is frame 0 (L1/L2-resident): the same arithmetic on the same random bits
  // with nothing streamed from HBM
  if (want(3)) run<N>("random, frame 0 only (L2)", d_iq, F, d_out, d_st, 256, T, 0);
  if (want(4)) run<N>("random, 224 workgroups", d_iq, F, d_out, d_st, 224, T);
  if (want(5)) run<N>("random, 192 workgroups", d_iq, F, d_out, d_st, 192, T);
  if (want(6)) run<N>("random, 128 workgroups", d_iq, F, d_out, d_st, 128, T);
  if (want(7)) run<N>("random, 64 workgroups", d_iq, F, d_out, d_st, 64, T);
  // the same random frames in allocations of other kinds (hipExtMallocWithFlags): does the memory type of the caller's
  // buffer -- what L2 and the memory-side cache may keep of a stream nobody reads twice -- move the power-capped rate?
  if (want(8) && g_only == 8) {
    const struct { const char* name; unsigned flags; } kinds[] = {
        {"random, uncached allocation", hipDeviceMallocUncached}, {"random, fine-grained allocation", hipDeviceMallocFinegrained}};
    for (const auto& kind : kinds) {
      float2* d_alt = nullptr;
      CHECK(hipExtMallocWithFlags(reinterpret_cast<void**>(&d_alt), n_samp * 8, kind.flags));
      CHECK(hipMemcpy(d_alt, d_iq, n_samp * 8, hipMemcpyDeviceToDevice));
      run<N>("random, all CUs (hipMalloc)", d_iq, F, d_out, d_st, 256, T);
      run<N>(kind.name, d_alt, F, d_out, d_st, 256, T);
      run<N>("random, all CUs (hipMalloc)", d_iq, F, d_out, d_st, 256, T);
      run<N>(kind.name, d_alt, F, d_out, d_st, 256, T);
      CHECK(hipFree(d_alt));
    }
  }
  CHECK(hipFree(d_iq)); CHECK(hipFree(d_zero)); CHECK(hipFree(d_out)); CHECK(hipFree(d_st));
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 2048;
  g_valu_per_frame = argc > 2 ? atof(argv[2]) : 0;
  g_only = argc > 3 ? atoi(argv[3]) : -1;
  g_seconds = argc > 4 ? atof(argv[4]) : 2.5;
#if defined(AMCX_ABL_NOFFT)
  printf("# build: FFT removed (AMCX_ABL_NOFFT)\n");
#elif defined(AMCX_ABL_NOSTATS)
  printf("# build: statistics sweep removed (AMCX_ABL_NOSTATS)\n");
#elif defined(AMCX_ABL_NORCP) && defined(AMCX_ABL_NOSQRT)
  printf("# build: both per-sample transcendentals removed (AMCX_ABL_NORCP + AMCX_ABL_NOSQRT)\n");
#elif defined(AMCX_ABL_NORCP)
  printf("# build: v_rcp_f32 of the half-angle quotient removed (AMCX_ABL_NORCP)\n");
#elif defined(AMCX_ABL_NOSQRT)
  printf("# build: v_sqrt_f32 of the envelope removed (AMCX_ABL_NOSQRT)\n");
#elif defined(AMCX_ABL_FFT_TAIL_MFMA)
  printf("# build: FFT passes 2-3 replaced by the instruction mix of the coarse-spectrum scheme: fp32 twiddles, fp16 conversions, 8 + 16 MFMAs, candidate scan, 2 exact candidates (AMCX_ABL_FFT_TAIL_MFMA; results wrong on purpose)\n");
#elif defined(AMCX_ABL_FFT_TAIL)
  printf("# build: FFT passes 2-3 and both exchanges removed, pass 1 kept (AMCX_ABL_FFT_TAIL; results wrong on purpose)\n");
#elif defined(AMCX_EXP_PK_FFT)
  printf("# build: packed-fp32 FFT butterflies, all passes (AMCX_EXP_PK_FFT)\n");
#elif defined(AMCX_EXP_PK_PASS1_ONLY)
  printf("# build: packed-fp32 butterflies in FFT pass 1 only (AMCX_EXP_PK_PASS1_ONLY)\n");
#elif defined(AMCX_EXP_PK_TAIL_ONLY)
  printf("# build: packed-fp32 butterflies in FFT passes 2-3 only (AMCX_EXP_PK_TAIL_ONLY)\n");
#else
  printf("# build: product instruction stream\n");
#endif
  const long long F = 6LL * 26 * 4096 * 2048 / N;   // the bench shard's bytes at every N
  switch (N) {
    case 1024: all<1024>(F); break;
    case 2048: all<2048>(F); break;
    case 4096: all<4096>(F); break;
    default: printf("frame size 1024, 2048 or 4096\n"); return 2;
  }
  return 0;
}
