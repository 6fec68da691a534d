// Issue-rate microbenchmarks for the VALU instructions the feature kernel
// leans on (gfx950).  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench
// Prints, per instruction kind and waves/SIMD, cycles per wave-instruction per SIMD
// (from s_memtime inside the kernel) and the chip-wide rate from HIP events.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITERS = 40000;
constexpr int UNROLL = 32;   // instructions per loop body

template <int KIND>
__global__ void k(float* out, unsigned long long* cyc) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b0 = 1.0001f, b1 = 0.9999f;
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3, e0 = 1.0000001, e1 = 1e-9;
  typedef float v2 __attribute__((ext_vector_type(2)));
  v2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, q0 = {b0, b1}, q1 = {b1, b0};
  unsigned long long msk = 0x5555aaaa5555aaaaull ^ (unsigned long long)blockIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < ITERS; ++i) {
    if constexpr (KIND == 0) {        // v_fma_f32, 8 independent chains
#pragma unroll
      for (int u = 0; u < UNROLL / 8; ++u)
        asm volatile(
            "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
            "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
    } else if constexpr (KIND == 1) { // v_pk_fma_f32, 4 independent chains
#pragma unroll
      for (int u = 0; u < UNROLL / 4; ++u)
        asm volatile(
            "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
            : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q0), "v"(q1));
    } else if constexpr (KIND == 2) { // v_fma_f64
#pragma unroll
      for (int u = 0; u < UNROLL / 4; ++u)
        asm volatile(
            "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
            : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(e0), "v"(e1));
    } else if constexpr (KIND == 3) { // v_rcp_f32
#pragma unroll
      for (int u = 0; u < UNROLL / 8; ++u)
        asm volatile(
            "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
            "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if constexpr (KIND == 4) { // v_sqrt_f32
#pragma unroll
      for (int u = 0; u < UNROLL / 8; ++u)
        asm volatile(
            "v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
            "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if constexpr (KIND == 5) { // v_add_f32 with DPP row_shr:1
#pragma unroll
      for (int u = 0; u < UNROLL / 8; ++u)
        asm volatile(
            "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
            "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
            "v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
            "v_add_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if constexpr (KIND == 6) { // v_cndmask_b32 (vcc)
#pragma unroll
      for (int u = 0; u < UNROLL / 8; ++u)
        asm volatile(
            "v_cndmask_b32_e64 %0, %0, %8, %9\n v_cndmask_b32_e64 %1, %1, %8, %9\n v_cndmask_b32_e64 %2, %2, %8, %9\n v_cndmask_b32_e64 %3, %3, %8, %9\n"
            "v_cndmask_b32_e64 %4, %4, %8, %9\n v_cndmask_b32_e64 %5, %5, %8, %9\n v_cndmask_b32_e64 %6, %6, %8, %9\n v_cndmask_b32_e64 %7, %7, %8, %9\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "s"(msk));
    } else if constexpr (KIND == 7) { // v_cvt_f64_f32 + v_add_f64 pair
#pragma unroll
      for (int u = 0; u < UNROLL / 8; ++u)
        asm volatile(
            "v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7\n"
            "v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
            : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(e0));
    } else if constexpr (KIND == 8) { // v_pk_mul_f32 + v_pk_add_f32
#pragma unroll
      for (int u = 0; u < UNROLL / 4; ++u)
        asm volatile(
            "v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %5\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5\n"
            : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q0), "v"(q1));
    }
#define OP8(str, ...) \
    _Pragma("unroll") for (int u = 0; u < UNROLL / 8; ++u) asm volatile(str : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1) __VA_ARGS__);
    else if constexpr (KIND == 10) { OP8("v_mul_f32_e32 %0, %8, %0\n v_mul_f32_e32 %1, %8, %1\n v_mul_f32_e32 %2, %8, %2\n v_mul_f32_e32 %3, %8, %3\n v_mul_f32_e32 %4, %8, %4\n v_mul_f32_e32 %5, %8, %5\n v_mul_f32_e32 %6, %8, %6\n v_mul_f32_e32 %7, %8, %7\n") }
    else if constexpr (KIND == 11) { OP8("v_add_f32_e32 %0, %8, %0\n v_add_f32_e32 %1, %8, %1\n v_add_f32_e32 %2, %8, %2\n v_add_f32_e32 %3, %8, %3\n v_add_f32_e32 %4, %8, %4\n v_add_f32_e32 %5, %8, %5\n v_add_f32_e32 %6, %8, %6\n v_add_f32_e32 %7, %8, %7\n") }
    else if constexpr (KIND == 12) { OP8("v_fmac_f32_e32 %0, %8, %9\n v_fmac_f32_e32 %1, %8, %9\n v_fmac_f32_e32 %2, %8, %9\n v_fmac_f32_e32 %3, %8, %9\n v_fmac_f32_e32 %4, %8, %9\n v_fmac_f32_e32 %5, %8, %9\n v_fmac_f32_e32 %6, %8, %9\n v_fmac_f32_e32 %7, %8, %9\n") }
    else if constexpr (KIND == 13) { OP8("v_mov_b32_e32 %0, %1\n v_mov_b32_e32 %1, %2\n v_mov_b32_e32 %2, %3\n v_mov_b32_e32 %3, %4\n v_mov_b32_e32 %4, %5\n v_mov_b32_e32 %5, %6\n v_mov_b32_e32 %6, %7\n v_mov_b32_e32 %7, %8\n") }
    else if constexpr (KIND == 14) { OP8("v_max_f32_e32 %0, %8, %0\n v_max_f32_e32 %1, %8, %1\n v_max_f32_e32 %2, %8, %2\n v_max_f32_e32 %3, %8, %3\n v_max_f32_e32 %4, %8, %4\n v_max_f32_e32 %5, %8, %5\n v_max_f32_e32 %6, %8, %6\n v_max_f32_e32 %7, %8, %7\n") }
    else if constexpr (KIND == 15) { OP8("v_fmaak_f32 %0, %8, %0, 0x3f800001\n v_fmaak_f32 %1, %8, %1, 0x3f800001\n v_fmaak_f32 %2, %8, %2, 0x3f800001\n v_fmaak_f32 %3, %8, %3, 0x3f800001\n v_fmaak_f32 %4, %8, %4, 0x3f800001\n v_fmaak_f32 %5, %8, %5, 0x3f800001\n v_fmaak_f32 %6, %8, %6, 0x3f800001\n v_fmaak_f32 %7, %8, %7, 0x3f800001\n") }
    else if constexpr (KIND == 16) { OP8("v_cmp_gt_f32_e32 vcc, %8, %0\n v_cndmask_b32_e32 %1, %1, %9, vcc\n v_cmp_gt_f32_e32 vcc, %8, %2\n v_cndmask_b32_e32 %3, %3, %9, vcc\n v_cmp_gt_f32_e32 vcc, %8, %4\n v_cndmask_b32_e32 %5, %5, %9, vcc\n v_cmp_gt_f32_e32 vcc, %8, %6\n v_cndmask_b32_e32 %7, %7, %9, vcc\n", : "vcc") }
    else if constexpr (KIND == 17) { OP8("v_mul_f32_e32 %0, %8, %0\n v_fma_f32 %1, %1, %8, %9\n v_add_f32_e32 %2, %8, %2\n v_fma_f32 %3, %3, %8, %9\n v_mul_f32_e32 %4, %8, %4\n v_fma_f32 %5, %5, %8, %9\n v_sub_f32_e32 %6, %8, %6\n v_fma_f32 %7, %7, %8, %9\n") }
    else if constexpr (KIND == 18) { OP8("v_fma_f32 %0, %0, %8, %1\n v_fma_f32 %1, %1, %8, %2\n v_fma_f32 %2, %2, %8, %3\n v_fma_f32 %3, %3, %8, %4\n v_fma_f32 %4, %4, %8, %5\n v_fma_f32 %5, %5, %8, %6\n v_fma_f32 %6, %6, %8, %7\n v_fma_f32 %7, %7, %8, %0\n") }
    else if constexpr (KIND == 19) { OP8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n") }
    else if constexpr (KIND == 20) { OP8("v_bfi_b32 %0, %8, %0, %9\n v_and_b32_e32 %1, %8, %1\n v_bfi_b32 %2, %8, %2, %9\n v_and_b32_e32 %3, %8, %3\n v_rndne_f32_e32 %4, %4\n v_min_f32_e32 %5, %8, %5\n v_rndne_f32_e32 %6, %6\n v_min_f32_e32 %7, %8, %7\n") }

  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + (float)(d0 + d1 + d2 + d3);
  if (s == 12345.678f) out[0] = s;
  if ((threadIdx.x & 63) == 0) { cyc[2 * ((blockIdx.x * blockDim.x + threadIdx.x) >> 6)] = t1 - t0; cyc[2 * ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) + 1] = r1 - r0; }
}

template <int KIND>
void run(const char* name, int waves_per_simd) {
  int cus = 256;
  int threads = 64 * 4 * waves_per_simd;   // one workgroup per CU, 4 SIMDs (3/SIMD -> 768 threads)
  if (threads > 1024) { threads = 1024; }
  int blocks_per_cu = (64 * 4 * waves_per_simd) / threads;
  int grid = cus * blocks_per_cu;
  float* out; unsigned long long* cyc;
  int nw = grid * threads / 64;
  CHECK(hipMalloc(&out, 4)); CHECK(hipMalloc(&cyc, nw * 16));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(threads), 0, 0, out, cyc);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(threads), 0, 0, out, cyc);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(2 * nw);
  CHECK(hipMemcpy(h.data(), cyc, nw * 16, hipMemcpyDeviceToHost));
  double avg = 0, avr = 0; for (int i = 0; i < nw; ++i) { avg += (double)h[2 * i]; avr += (double)h[2 * i + 1]; } avg /= nw; avr /= nw;
  double ghz = avg / (avr * 10.0);  // s_memrealtime ticks at 100 MHz
  double instr_per_wave = (double)ITERS * UNROLL;
  // cycles per wave-instruction per SIMD = wave cycles / (instr per wave * waves on the SIMD)
  printf("%-28s waves/SIMD=%d  cycles per wave-instr per SIMD = %.3f   clock %.2f GHz   chip rate = %.1f G wave-instr/s  (%.2f ms)\n",
         name, waves_per_simd, avg / (instr_per_wave * waves_per_simd), ghz, nw * instr_per_wave / (ms * 1e6), ms);
  CHECK(hipFree(out)); CHECK(hipFree(cyc));
}

int main() {
  for (int w : {1, 3, 4}) {
    run<0>("v_fma_f32", w);
    run<1>("v_pk_fma_f32", w);
    run<8>("v_pk_mul/add_f32", w);
    run<2>("v_fma_f64", w);
    run<3>("v_rcp_f32", w);
    run<4>("v_sqrt_f32", w);
    run<5>("v_add_f32_dpp", w);
    run<6>("v_cndmask_b32", w);
    run<7>("v_cvt_f64_f32+v_add_f64", w);
    run<10>("v_mul_f32_e32", w);
    run<11>("v_add_f32_e32", w);
    run<12>("v_fmac_f32_e32", w);
    run<13>("v_mov_b32_e32", w);
    run<14>("v_max_f32_e32", w);
    run<15>("v_fmaak_f32 (literal)", w);
    run<16>("v_cmp+v_cndmask vcc", w);
    run<17>("mix mul/fma/add/fma", w);
    run<18>("v_fma_f32 cross-dependent", w);
    run<19>("v_fma_f32 single chain", w);
    run<20>("bfi/and/rndne/min mix", w);
  }
  return 0;
}
