// AMCX_VARIANT_WAVE: one wavefront (64 lanes) per frame, frame held in registers.
//
// Why not "one 256-thread workgroup per frame": this path is VALU-bound, not
// HBM-bound (~110 fp32 lane-ops per sample against ~100 available at the HBM
// roofline), so the design minimises instructions per sample:
//   * 32 samples per lane (N = 2048) amortise every cross-lane reduction
//     over 32x more work than a 256-thread block would (8 samples per lane);
//   * waves never synchronise with each other: no s_barrier in the frame
//     loop, the LDS exchange buffer is private to the wave;
//   * the FFT is three register passes (radix 16, 16, 8 with constant
//     twiddles) joined by two conflict-free LDS transposes, instead of 11
//     shared-memory radix-2 stages (176 B/sample of LDS traffic -> 32 B);
//   * centred statistics are one-pass shifted sums (shift = mean of the first
//     64 samples' value), the envelope alone needs a second sweep over |x|
//     kept in registers, because f4 needs mean|a - mu| with the exact mu;
//   * the fp64 scalar algebra that turns 27 sums into 18 features runs once
//     per batch of kFramesPerWave frames with one frame per lane.
//
// Index maps (verified against np.fft.fft with the LDS bank rules in
// tools/wave_fft_model.py), N = 2048 = R1*R2*R3 = 16*16*8:
//   load     lane l, register (i, b)  <- x[128 i + 2 l + b]   (global_load_dwordx4, coalesced)
//   pass 1   16-point DFT over i  -> k1 ; twiddle W_2048^((2l+b) k1)
//   xchg 1   two phases g = k1>>3:  LDS[kk*136 + b*68 + l] (kk = k1&7), reader lane l'
//            (kk = l'>>3, n3 = l'&7) takes n2 = 0..15 at [kk*136 + (n3&1)*68 + 4 n2 + (n3>>1)]
//   pass 2   16-point DFT over n2 -> k2 ; twiddle W_128^(n3 k2)
//   xchg 2   LDS[k2*65 + 8 kk + n3], reader lane l'' (kk = l''>>3, k2 = (l''&7) + 8 j)
//   pass 3   8-point DFT over n3  -> X[k1 + 16 k2 + 256 k3], only max |X|^2 is kept
// Every ds_write_b64 / ds_read_b64 / ds_read_b128 of the exchanges and twiddle
// tables is bank-conflict free and addressed as lane_base + immediate.
//
// LDS per 768-thread workgroup: twiddles 15 KiB + 960 B, 12 x 8704 B exchange,
// 12 x 1056 B sums stash = 133.4 KB -> one workgroup (12 waves, exactly 3 per
// SIMD) per CU.  (Two 6-wave workgroups do NOT co-reside: their waves land
// 2,2,1,1 on the SIMDs and a SIMD holds at most 3 waves of 160 VGPRs --
// measured as half the waves' lifetime in SQ_WAVE_CYCLES, profiles/r1.)
// Algorithmic HBM bytes per frame: 8*N read + 72 written.
#pragma once

#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "amcx_math.h"

namespace amcx {
namespace wave {

constexpr int kWavesPerWG = 12;              // 3 per SIMD: one workgroup fills a CU
constexpr int kThreads = 64 * kWavesPerWG;
constexpr int kFramesPerWave = 8;               // frames per wave per batch (finalised together)
constexpr int kFramesPerBatch = kWavesPerWG * kFramesPerWave;
constexpr int kNumSums = 27;
constexpr int kStashStride = 33;                // floats; odd -> conflict-free column reads

constexpr int kEx1StrideKK = 136, kEx1StrideB = 68;   // complex units (tools/wave_fft_model.py)
constexpr int kEx2StrideK2 = 65;
constexpr int kExchangeBytes = 8 * kEx1StrideKK * 8;   // 8704 >= 16*65*8 = 8320

constexpr int kT1Bytes = 15 * 64 * 16;          // [k1-1][lane][b] complex
constexpr int kT2Bytes = 15 * 8 * 8;            // [k2-1][n3] complex
constexpr int kLdsBytes = kT1Bytes + kT2Bytes + kWavesPerWG * (kExchangeBytes + kFramesPerWave * kStashStride * 4);

// ---- compile-time loop -------------------------------------------------------
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

constexpr int bitrev(int v, int bits) {
  int r = 0;
  for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1) << (bits - 1 - i);
  return r;
}

// cos / sin of 2*pi*j/16
constexpr float kC16[16] = {1.f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f,
                            0.f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f,
                            -1.f, -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f,
                            0.f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f};
constexpr float kS16[16] = {0.f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f,
                            1.f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f,
                            0.f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f,
                            -1.f, -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f};

// (r, i) *= W_16^J = cos - i sin, with the trivial cases folded at compile time
template <int J>
__device__ __forceinline__ void mul_w16(float& r, float& i) {
  constexpr float h = 0.70710678118654752f;
  if constexpr (J == 0) {
  } else if constexpr (J == 4) {            // -i
    const float t = r; r = i; i = -t;
  } else if constexpr (J == 2) {            // (1 - i)/sqrt2
    const float t = (r + i) * h; i = (i - r) * h; r = t;
  } else if constexpr (J == 6) {            // (-1 - i)/sqrt2
    const float t = (i - r) * h; i = (-r - i) * h; r = t;
  } else {
    constexpr float c = kC16[J], s = kS16[J];
    const float t = __builtin_fmaf(r, c, i * s);
    i = __builtin_fmaf(i, c, -(r * s));
    r = t;
  }
}

// In-place radix-2 decimation-in-frequency DFT of LEN points at [OFF, OFF+LEN);
// result for frequency k sits at OFF + bitrev(k).
template <int LEN, int OFF, int R>
__device__ __forceinline__ void dif(float (&re)[R], float (&im)[R]) {
  if constexpr (LEN >= 2) {
    constexpr int H = LEN / 2;
    static_for<H>([&](auto jj) {
      constexpr int j = decltype(jj)::value;
      const float ar = re[OFF + j], ai = im[OFF + j], br = re[OFF + j + H], bi = im[OFF + j + H];
      re[OFF + j] = ar + br;
      im[OFF + j] = ai + bi;
      float dr = ar - br, di = ai - bi;
      mul_w16<j * (16 / LEN)>(dr, di);
      re[OFF + j + H] = dr;
      im[OFF + j + H] = di;
    });
    dif<H, OFF, R>(re, im);
    dif<H, OFF + H, R>(re, im);
  }
}

// ---- DPP helpers -------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf, bool BOUND = true>
__device__ __forceinline__ float dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL,
                                                                ROW_MASK, BANK_MASK, BOUND));
}
constexpr int kQuadXor1 = 0xB1;   // quad_perm [1,0,3,2]
constexpr int kQuadXor2 = 0x4E;   // quad_perm [2,3,0,1]
constexpr int kRowHalfMirror = 0x141, kRowMirror = 0x140, kRowBcast15 = 0x142, kRowBcast31 = 0x143;
constexpr int kWaveRol1 = 0x134;  // lane l reads lane (l+1) mod 64

// wave-wide sum, valid in lane 63
__device__ __forceinline__ float wave_sum_l63(float v) {
  v += dpp<kQuadXor1>(v);
  v += dpp<kQuadXor2>(v);
  v += dpp<kRowHalfMirror>(v);
  v += dpp<kRowMirror>(v);
  v += dpp<kRowBcast15, 0xa, 0xf, false>(v);
  v += dpp<kRowBcast31, 0xc, 0xf, false>(v);
  return v;
}
__device__ __forceinline__ float wave_max_l63(float v) {
  v = __builtin_fmaxf(v, dpp<kQuadXor1>(v));
  v = __builtin_fmaxf(v, dpp<kQuadXor2>(v));
  v = __builtin_fmaxf(v, dpp<kRowHalfMirror>(v));
  v = __builtin_fmaxf(v, dpp<kRowMirror>(v));
  // bound_ctrl=false keeps the lane's own value where the source row does not exist
  v = __builtin_fmaxf(v, dpp<kRowBcast15, 0xa, 0xf, false>(v));
  v = __builtin_fmaxf(v, dpp<kRowBcast31, 0xc, 0xf, false>(v));
  return v;
}
__device__ __forceinline__ float bcast_l63(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ void lds_wave_fence() {
  // exchanges are wave-private: LDS instructions of one wave execute in order,
  // so only the compiler has to be kept from reordering across this point
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

struct c2 { float re, im; };

// Diagnostic build only (tools/wave_stamps.hip defines AMCX_WAVE_STAMPS): per-section
// s_memtime deltas summed per wave into a buffer nothing else reads.  The product
// build compiles none of this.
#ifdef AMCX_WAVE_STAMPS
#define AMCX_STAMP_ARG , unsigned long long* __restrict__ stamp_out
#define AMCX_STAMP(sec)                                                               \
  do {                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                \
    unsigned long long t_;                                                            \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
    __builtin_amdgcn_sched_barrier(0);                                                \
    stamp_acc[sec] += t_ - stamp_last;                                                \
    stamp_last = t_;                                                                  \
  } while (0)
#else
#define AMCX_STAMP_ARG
#define AMCX_STAMP(sec) do { } while (0)
#endif
constexpr int kStampSections = 8;

// ---------------------------------------------------------------------------
template <int N, bool PREFETCH>
__global__ __launch_bounds__(kThreads, 3) void amcx_features18_wave_kernel(
    const float2* __restrict__ iq, long long n_frames, long long row_stride,
    float* __restrict__ out, long long out_stride AMCX_STAMP_ARG) {
  static_assert(N == 2048, "register-FFT kernel is instantiated for N = 2048");
  extern __shared__ float4 amcx_wave_smem[];
  char* smem = reinterpret_cast<char*>(amcx_wave_smem);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* t1 = smem;                                        // [15][64][2] complex
  char* t2 = smem + kT1Bytes;                             // [15][8] complex
  char* ex = smem + kT1Bytes + kT2Bytes + wave * kExchangeBytes;
  float* stash = reinterpret_cast<float*>(smem + kT1Bytes + kT2Bytes + kWavesPerWG * kExchangeBytes) +
                 wave * (kFramesPerWave * kStashStride);

  // ---- twiddle tables, once per workgroup -----------------------------------
  for (int e = tid; e < 15 * 128; e += kThreads) {         // T1[k1-1][l][b] = W_2048^((2l+b) k1)
    const int k1 = e / 128 + 1, lb = e % 128;
    float sn, cs;
    sincospif((float)(lb * k1) * (1.0f / 1024.0f), &sn, &cs);
    reinterpret_cast<float2*>(t1)[e] = make_float2(cs, -sn);
  }
  for (int e = tid; e < 15 * 8; e += kThreads) {           // T2[k2-1][n3] = W_128^(n3 k2)
    const int k2 = e / 8 + 1, n3 = e % 8;
    float sn, cs;
    sincospif((float)(n3 * k2) * (1.0f / 64.0f), &sn, &cs);
    reinterpret_cast<float2*>(t2)[e] = make_float2(cs, -sn);
  }
  __syncthreads();

  // lane-constant LDS byte addresses
  const int kkL = lane >> 3, n3L = lane & 7;
  char* const t1_lane = t1 + lane * 16;
  char* const t2_lane = t2 + n3L * 8;
  char* const ex1_w = ex + lane * 8;
  char* const ex1_r = ex + (kkL * kEx1StrideKK + (n3L & 1) * kEx1StrideB + (n3L >> 1)) * 8;
  char* const ex2_w = ex + lane * 8;
  char* const ex2_r = ex + (n3L * kEx2StrideK2 + kkL * 8) * 8;   // (lane&7) is k2lo for the reader

  const long long n_batches = (n_frames + kFramesPerBatch - 1) / kFramesPerBatch;
#ifdef AMCX_WAVE_STAMPS
  unsigned long long stamp_acc[kStampSections] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
  const unsigned long long real0 = __builtin_amdgcn_s_memrealtime();
#endif
  float4 xv[16];                 // lane l, vector i = samples 128 i + 2 l + {0, 1}
  bool have_next = false;        // xv already holds (or is receiving) the frame about to be processed
  auto load_frame = [&](long long f) {
    const float2* src = iq + f * row_stride + 2 * lane;
    static_for<16>([&](auto ii) {
      constexpr int i = decltype(ii)::value;
      xv[i] = *reinterpret_cast<const float4*>(src + 128 * i);
    });
  };
  for (long long batch = blockIdx.x; batch < n_batches; batch += gridDim.x) {
    const long long f0 = batch * kFramesPerBatch + (long long)wave * kFramesPerWave;
    long long left = n_frames - f0;
    const int n_here = left <= 0 ? 0 : (left < kFramesPerWave ? (int)left : kFramesPerWave);

    // the frame's 16 x 16-byte vectors; the NEXT frame's are requested as soon as
    // pass 1 has parked this frame in LDS, so HBM latency hides under passes 2-3
    if (PREFETCH && !have_next && n_here > 0) load_frame(f0);
    for (int g = 0; g < n_here; ++g) {
      asm volatile("; MARK load");
      AMCX_STAMP(7);
      float xr[32], xi[32];
      if constexpr (PREFETCH) {
        static_for<16>([&](auto ii) {
          constexpr int i = decltype(ii)::value;
          xr[2 * i] = xv[i].x; xi[2 * i] = xv[i].y; xr[2 * i + 1] = xv[i].z; xi[2 * i + 1] = xv[i].w;
        });
      } else {
        const float2* src = iq + (f0 + g) * row_stride + 2 * lane;
        static_for<16>([&](auto ii) {
          constexpr int i = decltype(ii)::value;
          const float4 v = *reinterpret_cast<const float4*>(src + 128 * i);
          xr[2 * i] = v.x; xi[2 * i] = v.y; xr[2 * i + 1] = v.z; xi[2 * i + 1] = v.w;
        });
      }
      // frame this wave handles after (f0 + g): next of the batch, else first of its next batch
      long long f_next = f0 + g + 1;
      if (g + 1 >= n_here) {
        f_next = (batch + gridDim.x) * kFramesPerBatch + (long long)wave * kFramesPerWave;
      }
      have_next = f_next < n_frames && (g + 1 < n_here || batch + gridDim.x < n_batches);

      // =====================================================================
      // statistics sweep
      // =====================================================================
      float sA = 0, sBh = 0, sP = 0, sAA = 0, sBB = 0, sAB = 0, sAP = 0, sBP = 0;
      float sAAA = 0, sABB = 0, sAAB = 0, sBBB = 0, sAAP = 0, sBBP = 0, sABP = 0;
      float sa = 0, st1 = 0, st2 = 0, sabst = 0, sw1 = 0, sw2 = 0, sw3 = 0, sw4 = 0;
      float Kt = 0, Kw = 0;
      float* const a_lds = reinterpret_cast<float*>(ex) + lane;   // |x| parked in the wave's LDS: [e][lane]
      float th_b1_prev = 0.f;     // angle of sample (i-1, b=1), waiting for its right neighbour
      float rot_prev = 0.f;       // wave_rol1(angle(i-1, b=0))

      static_for<16>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        float th[2];
        static_for<2>([&](auto bb) {
          constexpr int b = decltype(bb)::value;
          constexpr int e = 2 * i + b;
          const float re = xr[e], im = xi[e];
          const float q = __builtin_fmaf(im, im, kTinyPower);   // zero guard for fast_angle, free in the fma
          const float P = __builtin_fmaf(re, re, q);
          const float A = __builtin_fmaf(re, re, -q);
          const float Bh = re * im;
          const float AA = A * A, BB = Bh * Bh, AP = A * P;
          sA += A; sBh += Bh; sP += P; sAA += AA; sBB += BB; sAP += AP;
          sAB = __builtin_fmaf(A, Bh, sAB);
          sBP = __builtin_fmaf(Bh, P, sBP);
          sAAA = __builtin_fmaf(AA, A, sAAA);
          sABB = __builtin_fmaf(A, BB, sABB);
          sAAB = __builtin_fmaf(AA, Bh, sAAB);
          sBBB = __builtin_fmaf(BB, Bh, sBBB);
          sAAP = __builtin_fmaf(AA, P, sAAP);
          sBBP = __builtin_fmaf(BB, P, sBBP);
          sABP = __builtin_fmaf(AP, Bh, sABP);
          const float av = __builtin_amdgcn_sqrtf(P);
          a_lds[e * 64] = av;
          sa += av;
          th[b] = fast_angle(re, im, av);
        });
        if constexpr (i == 0) {
          // shifts: mean over the wave of the first angle / first step
          const float w00 = wrapped_step(th[1], th[0]);
          Kt = bcast_l63(wave_sum_l63(th[0])) * (1.0f / 64.0f);
          Kw = bcast_l63(wave_sum_l63(w00)) * (1.0f / 64.0f);
        }
        static_for<2>([&](auto bb) {
          constexpr int b = decltype(bb)::value;
          const float d = th[b] - Kt;
          st1 += d;
          st2 = __builtin_fmaf(d, d, st2);
          sabst += __builtin_fabsf(th[b]);
        });
        auto add_step = [&](float w) {
          const float d = w - Kw, d2 = d * d;
          sw1 += d; sw2 += d2;
          sw3 = __builtin_fmaf(d2, d, sw3);
          sw4 = __builtin_fmaf(d2, d2, sw4);
        };
        add_step(wrapped_step(th[1], th[0]));                 // sample (i,0) -> (i,1), same lane
        const float rot = dpp<kWaveRol1>(th[0]);              // lane l: angle(i,0) of lane l+1 (63 -> lane 0)
        if constexpr (i > 0) {
          // right neighbour of (i-1, b=1) is (i-1, b=0) of lane l+1, or (i, b=0) of lane 0 for lane 63
          const float nxt = (lane == 63) ? rot : rot_prev;
          add_step(wrapped_step(nxt, th_b1_prev));
        }
        if constexpr (i == 15) {
          // (15, b=1): lane 63 holds the frame's last sample, which has no step
          const float w = wrapped_step(rot, th[1]);
          add_step(lane == 63 ? Kw : w);
        }
        th_b1_prev = th[1];
        rot_prev = rot;
      });

      asm volatile("; MARK envelope");
      AMCX_STAMP(0);
      __builtin_amdgcn_sched_barrier(0);
      // envelope second sweep about the exact mean
      const float mu = bcast_l63(wave_sum_l63(sa)) * (1.0f / (float)N);
      float sad1 = 0, sad2 = 0, sad4 = 0;
      static_for<32>([&](auto ee) {
        constexpr int e = decltype(ee)::value;
        const float d = a_lds[e * 64] - mu, d2 = d * d;
        sad1 += __builtin_fabsf(d);
        sad2 += d2;
        sad4 = __builtin_fmaf(d2, d2, sad4);
      });

      // =====================================================================
      // spectral peak: 16 x 16 x 8 register FFT
      // =====================================================================
      asm volatile("; MARK fft1");
      AMCX_STAMP(1);
      __builtin_amdgcn_sched_barrier(0);
      // pass 1 (both b groups), twiddle T1, exchange 1
      float v0r[16], v0i[16], v1r[16], v1i[16];
      static_for<16>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        v0r[i] = xr[2 * i]; v0i[i] = xi[2 * i]; v1r[i] = xr[2 * i + 1]; v1i[i] = xi[2 * i + 1];
      });
      dif<16, 0>(v0r, v0i);
      dif<16, 0>(v1r, v1i);
      static_for<15>([&](auto kk1) {
        constexpr int k1 = decltype(kk1)::value + 1;
        constexpr int p = bitrev(k1, 4);
        const float4 t = *reinterpret_cast<const float4*>(t1_lane + (k1 - 1) * 1024);
        float r = v0r[p], im = v0i[p];
        v0r[p] = __builtin_fmaf(r, t.x, -(im * t.y));
        v0i[p] = __builtin_fmaf(r, t.y, im * t.x);
        r = v1r[p]; im = v1i[p];
        v1r[p] = __builtin_fmaf(r, t.z, -(im * t.w));
        v1i[p] = __builtin_fmaf(r, t.w, im * t.z);
      });
      float zr[2][16], zi[2][16];
      static_for<2>([&](auto gg) {
        constexpr int gph = decltype(gg)::value;
        lds_wave_fence();
        static_for<8>([&](auto kk_) {
          constexpr int kk = decltype(kk_)::value;
          constexpr int p = bitrev(8 * gph + kk, 4);
          *reinterpret_cast<float2*>(ex1_w + (kk * kEx1StrideKK) * 8) = make_float2(v0r[p], v0i[p]);
          *reinterpret_cast<float2*>(ex1_w + (kk * kEx1StrideKK + kEx1StrideB) * 8) = make_float2(v1r[p], v1i[p]);
        });
        lds_wave_fence();
        static_for<16>([&](auto nn) {
          constexpr int n2 = decltype(nn)::value;
          const float2 v = *reinterpret_cast<const float2*>(ex1_r + n2 * 32);
          zr[gph][n2] = v.x; zi[gph][n2] = v.y;
        });
      });

      // x is dead: prefetch behind passes 2 and 3 (branch-free: with nothing left, the
      // current frame is simply requested again and never used)
      if (PREFETCH) load_frame(have_next ? f_next : f0 + g);
      asm volatile("; MARK fft2");
      AMCX_STAMP(2);
      __builtin_amdgcn_sched_barrier(0);
      // pass 2, twiddle T2, exchange 2, pass 3
      dif<16, 0>(zr[0], zi[0]);
      dif<16, 0>(zr[1], zi[1]);
      static_for<15>([&](auto kk2) {
        constexpr int k2 = decltype(kk2)::value + 1;
        constexpr int p = bitrev(k2, 4);
        const float2 t = *reinterpret_cast<const float2*>(t2_lane + (k2 - 1) * 64);
        static_for<2>([&](auto gg) {
          constexpr int gph = decltype(gg)::value;
          const float r = zr[gph][p], im = zi[gph][p];
          zr[gph][p] = __builtin_fmaf(r, t.x, -(im * t.y));
          zi[gph][p] = __builtin_fmaf(r, t.y, im * t.x);
        });
      });
      float peak = 0.f;
      static_for<2>([&](auto gg) {
        constexpr int gph = decltype(gg)::value;
        lds_wave_fence();
        static_for<16>([&](auto kk2) {
          constexpr int k2 = decltype(kk2)::value;
          constexpr int p = bitrev(k2, 4);
          *reinterpret_cast<float2*>(ex2_w + (k2 * kEx2StrideK2) * 8) = make_float2(zr[gph][p], zi[gph][p]);
        });
        lds_wave_fence();
        static_for<2>([&](auto jj) {
          constexpr int j = decltype(jj)::value;
          float ur[8], ui[8];
          static_for<8>([&](auto nn) {
            constexpr int n3 = decltype(nn)::value;
            const float2 v = *reinterpret_cast<const float2*>(ex2_r + (j * 8 * kEx2StrideK2 + n3) * 8);
            ur[n3] = v.x; ui[n3] = v.y;
          });
          dif<8, 0>(ur, ui);
          static_for<8>([&](auto pp) {
            constexpr int p = decltype(pp)::value;
            peak = __builtin_fmaxf(peak, __builtin_fmaf(ur[p], ur[p], ui[p] * ui[p]));
          });
        });
      });
      lds_wave_fence();

      // =====================================================================
      // wave reduction of the 27 partial results -> stash row g (lane 63 writes)
      // =====================================================================
      asm volatile("; MARK reduce");
      AMCX_STAMP(3);
      __builtin_amdgcn_sched_barrier(0);
      // 26 sums: two swap levels (lane bits 5, 4) halve the live values each time --
      // v_permlane32_swap / v_permlane16_swap exchange half a register pair in one
      // instruction -- then four DPP steps inside the 16-lane rows.  70 VALU ops
      // against 156 for 26 independent 6-step butterflies.
      float r28[28] = {sA, sBh, sP, sAA, sBB, sAB, sAP, sBP, sAAA, sABB, sAAB, sBBB, sAAP, sBBP,
                       sABP, sa, sad1, sad2, sad4, st1, st2, sabst, sw1, sw2, sw3, sw4, 0.f, 0.f};
      // (inline asm: hipcc 7.2 folds the two results of __builtin_amdgcn_permlane*_swap
      //  into one register here -- "v_add v3, v142, v142" -- so the swaps are spelled
      //  out; one statement per level, opening with the two wait states a VALU
      //  write -> v_permlane* read needs, which hipcc does not add inside asm.)
      asm volatile(
          "s_nop 1\n\t"
          "v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\t"
          "v_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\t"
          "v_permlane32_swap_b32 %8, %9\n\tv_permlane32_swap_b32 %10, %11\n\t"
          "v_permlane32_swap_b32 %12, %13\n\tv_permlane32_swap_b32 %14, %15\n\t"
          "v_permlane32_swap_b32 %16, %17\n\tv_permlane32_swap_b32 %18, %19\n\t"
          "v_permlane32_swap_b32 %20, %21\n\tv_permlane32_swap_b32 %22, %23\n\t"
          "v_permlane32_swap_b32 %24, %25\n\tv_permlane32_swap_b32 %26, %27"
          : "+v"(r28[0]), "+v"(r28[1]), "+v"(r28[2]), "+v"(r28[3]), "+v"(r28[4]), "+v"(r28[5]),
            "+v"(r28[6]), "+v"(r28[7]), "+v"(r28[8]), "+v"(r28[9]), "+v"(r28[10]), "+v"(r28[11]),
            "+v"(r28[12]), "+v"(r28[13]), "+v"(r28[14]), "+v"(r28[15]), "+v"(r28[16]), "+v"(r28[17]),
            "+v"(r28[18]), "+v"(r28[19]), "+v"(r28[20]), "+v"(r28[21]), "+v"(r28[22]), "+v"(r28[23]),
            "+v"(r28[24]), "+v"(r28[25]), "+v"(r28[26]), "+v"(r28[27]));
      // lanes 0-31: value 2j summed over lane bit 5; lanes 32-63: value 2j+1
      float r14[14];
      static_for<14>([&](auto jj) {
        constexpr int j = decltype(jj)::value;
        r14[j] = r28[2 * j] + r28[2 * j + 1];
      });
      asm volatile(
          "s_nop 1\n\t"
          "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\t"
          "v_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\t"
          "v_permlane16_swap_b32 %8, %9\n\tv_permlane16_swap_b32 %10, %11\n\t"
          "v_permlane16_swap_b32 %12, %13"
          : "+v"(r14[0]), "+v"(r14[1]), "+v"(r14[2]), "+v"(r14[3]), "+v"(r14[4]), "+v"(r14[5]),
            "+v"(r14[6]), "+v"(r14[7]), "+v"(r14[8]), "+v"(r14[9]), "+v"(r14[10]), "+v"(r14[11]),
            "+v"(r14[12]), "+v"(r14[13]));
      // rows 0..3 now hold values 4j+0, 4j+2, 4j+1, 4j+3 summed over lane bits 5 and 4
      float r7[7];
      static_for<7>([&](auto jj) {
        constexpr int j = decltype(jj)::value;
        float v = r14[2 * j] + r14[2 * j + 1];
        v += dpp<kQuadXor1>(v);
        v += dpp<kQuadXor2>(v);
        v += dpp<kRowHalfMirror>(v);
        v += dpp<kRowMirror>(v);
        r7[j] = v;
      });
      const float pk = wave_max_l63(peak);
      {
        float* row = stash + g * kStashStride;
        if ((lane & 15) == 0) {                    // one lane per row: rows hold 4j + {0, 2, 1, 3}
          const int rsel = lane >> 4;
          float* dst = row + (((rsel & 1) << 1) | (rsel >> 1));
          static_for<7>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            dst[4 * j] = r7[j];
          });
        }
        if (lane == 63) {
          row[kNumSums - 1] = pk;                  // overwrites the zero pad slot 26
          row[kNumSums] = Kt;
          row[kNumSums + 1] = Kw;
        }
      }
    }

    AMCX_STAMP(4);
    asm volatile("; MARK finalize");
    // ---- batch finalisation: lane g turns frame g's sums into 18 features ----
    lds_wave_fence();
    if (lane < n_here) {
      const float* row = stash + lane * kStashStride;
      FrameSums S;
      S.sA = row[0]; S.sBh = row[1]; S.sP = row[2]; S.sAA = row[3]; S.sBB = row[4]; S.sAB = row[5];
      S.sAP = row[6]; S.sBP = row[7]; S.sAAA = row[8]; S.sABB = row[9]; S.sAAB = row[10];
      S.sBBB = row[11]; S.sAAP = row[12]; S.sBBP = row[13]; S.sABP = row[14];
      S.sa = row[15]; S.sad1 = row[16]; S.sad2 = row[17]; S.sad4 = row[18];
      S.std1 = row[19]; S.std2 = row[20]; S.sabst = row[21];
      S.swd1 = row[22]; S.swd2 = row[23]; S.swd3 = row[24]; S.swd4 = row[25];
      S.gmax_raw = row[26]; S.Kt = row[27]; S.Kw = row[28];
      float feat[18];
      finalize_features(S, N, feat);
      float* dst = out + (f0 + lane) * out_stride;
#pragma unroll
      for (int j = 0; j < 18; ++j) dst[j] = feat[j];
    }
    lds_wave_fence();
    AMCX_STAMP(5);
  }
#ifdef AMCX_WAVE_STAMPS
  if (lane == 0) {
    const long long w = (long long)blockIdx.x * kWavesPerWG + wave;
    stamp_acc[6] = __builtin_amdgcn_s_memrealtime() - real0;   // wave lifetime, 100 MHz ticks
    for (int k = 0; k < kStampSections; ++k) stamp_out[w * kStampSections + k] = stamp_acc[k];
  }
#endif
}

}  // namespace wave

inline bool wave_supports(int frame_size) { return frame_size == 2048; }

inline const char* wave_kernel_name(int frame_size) {
  return frame_size == 2048 ? "amcx_features18_wave_kernel<2048>" : "";
}

#ifndef AMCX_WAVE_STAMPS
inline hipError_t launch_wave(const float2* iq, int64_t n_frames, int32_t frame_size,
                              int64_t row_stride, float* out, int64_t out_stride,
                              hipStream_t stream, int cus) {
  if (frame_size != 2048) return hipErrorNotSupported;
  static const bool prefetch = getenv("AMCX_WAVE_PREFETCH") != nullptr;   // experiment switch
  auto kern = prefetch ? wave::amcx_features18_wave_kernel<2048, true> : wave::amcx_features18_wave_kernel<2048, false>;
  static_assert(wave::kLdsBytes <= 160 * 1024, "one workgroup per CU must fit in 160 KiB of LDS");
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, wave::kLdsBytes);
  if (e != hipSuccess) return e;
  const int64_t n_batches = (n_frames + wave::kFramesPerBatch - 1) / wave::kFramesPerBatch;
  int64_t grid = (int64_t)cus;                        // persistent: one resident workgroup per CU
  if (grid > n_batches) grid = n_batches;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(wave::kThreads), wave::kLdsBytes, stream, iq,
                     (long long)n_frames, (long long)row_stride, out, (long long)out_stride);
  return hipGetLastError();
}
#endif

}  // namespace amcx
