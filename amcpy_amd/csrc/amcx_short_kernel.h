// N = 128, 256 and 512 (RadioML-2016-style short frames): FOUR frames per wavefront, sixteen lanes per frame.
//
// The one-wave-per-frame kernel (amcx_wave_kernel.h) spreads a 128-sample frame over 64 lanes, two samples each: every
// frame then pays a full 64-lane reduction of its 27 sums, another for the envelope's mean, and shares a 1024-point-shaped
// FFT back half with seven other frames -- 327 VALU instructions per frame where the per-sample work is ~210, and at FULL
// clock (not the board's power cap) a quarter of its wave-cycles in s_waitcnt (profiles/r5_short_kernel_ab.txt).  Here a
// frame lives in ONE 16-lane DPP row (247 instructions per frame at N = 128):
//   * lane l of a row holds samples 32 j + 2 l + b, j < 4, b < 2 (four global_load_dwordx4, 256 contiguous bytes per row
//     and load); the statistics sweep is the wave kernel's, with the neighbour's angle from a row rotate (row_ror:15) and
//     the shifts from the row's first 16 samples;
//   * every reduction is a row reduction: four DPP steps (quad xor 1, xor 2, half-mirror, mirror) leave the total in all
//     sixteen lanes -- the mean envelope costs 4 instructions instead of a wave reduction, the 27 sums 108 for FOUR frames;
//   * every frame is multiplied by 2^-ex first (ex the even-rounded exponent of its largest component, a row maximum), as
//     the block kernel does, and finalised through the features' scaling laws (finalize_features<true>): no range check, no
//     re-run pass, any float32 input;
//   * the 128-point FFT, X[kj + 4 (kc + 8 ka)]: pass 1 radix 4 over j in registers (twiddle W_128^(m kj), m = 2 l + b),
//     a transpose through LDS so that lane (kj, a) holds y[kj][4 c + a], c < 8; pass 2 radix 8 over c in registers (twiddle
//     W_32^(a kc)); a second transpose so that lane (kj, h) holds z[kj][kc = 2 h + e][a], a < 4; pass 3 radix 4 over a.
//     All three passes are the wave kernel's compile-time dif<>; the four frames of a wave go through together;
//   * after eight passes (32 frames; four passes at N = 256 / 512) lanes 0-31 turn a stash row each into 18 features in fp64; frames with a phase step
//     within an angle rounding of +-pi get f5 / f9 from the exact fp64 sweep (wave_exact_frequency).
// N = 256 is the same machine with eight rows per lane: pass 1 is a radix 8 over j, and a lane takes TWO of the eight
// 32-point transforms of passes 2 and 3 (kj = l / 4 and l / 4 + 4), one after the other.  N = 512 has sixteen rows per lane:
// pass 1 is a radix 16 over j, its sixteen 32-point transforms go through the exchange block in two batches of eight, only
// four of the sixteen rows are requested a pass ahead (the rest at the head of the pass: with eight ahead three of them were
// stored to scratch straight from their loads), and |x| is parked in the exchange block for the envelope's second sweep.
// N = 128: 16 waves per CU (116 VGPRs); N = 256 / 512: 12 (143 / 162 VGPRs; a 9.5 KB exchange block per wave).  No barrier in the frame loop,
// nothing shared between waves but the twiddle tables.  Algorithmic HBM bytes per frame: 8 N + 72.
#pragma once

#include "amcx_wave_kernel.h"

namespace amcx {
namespace shortk {

using namespace wave;

constexpr int kRowRor15 = 0x12F;                             // DPP row_ror:15: lane l reads lane (l + 1) mod 16 of its row
constexpr int kQuad = 4;                                     // frames per wave per pass
constexpr int kKjStride = 36;                                // complex elements: 32 per (frame, kj) block + 4 (bank spread)
constexpr int kRow = 36;                                     // floats per stash row: 0-26 sums, 27 peak, 28-30 shifts, 31 tie, 32 ex
constexpr int kTw2Bytes = 4 * 8 * 8;                         // pass 2: [a][kc] complex

template <int N>
struct SCfg {
  static_assert(N == 128 || N == 256 || N == 512, "sixteen lanes per frame: 8, 16 or 32 samples a lane");
  static constexpr int kRows = N / 32;                        // rows of 32 samples (16 lanes x 2) per frame
  static constexpr int kLogRows = N == 128 ? 2 : N == 256 ? 3 : 4;
  // passes 2 and 3 take the kRows 32-point transforms of a frame in kHalves batches of kBlocks through the exchange block
  // (N = 512: two batches of eight, so that the block stays at 9.5 KB), a lane kRounds of them per batch
  static constexpr int kHalves = N == 512 ? 2 : 1, kBlocks = kRows / kHalves, kRounds = kBlocks / 4;
  static constexpr int kWavesPerWG = N == 128 ? 16 : 12, kThreads = 64 * kWavesPerWG;
  static constexpr int kBatchPasses = N == 128 ? 8 : 4, kBatch = kQuad * kBatchPasses;   // frames finalised together, one per lane
  // rows requested a pass ahead (the rest at the head of the pass itself: N = 512 has 64 registers of samples)
  static constexpr int kHead = N == 512 ? 4 : kRows;
  // |x| of the sweep for the envelope's second sweep: in registers, or (N = 512) parked in the exchange block
  static constexpr bool kParkA = N == 512;
  static constexpr int kFrameStride = kBlocks * kKjStride + (N == 128 ? 0 : 16);   // the four frames of a pass 128 bytes apart in the banks
  static constexpr int kExBytes = kQuad * kFrameStride * 8;   // 4 608 / 9 728 / 9 728
  static_assert(!kParkA || 2 * kRows * 64 * 4 <= kExBytes, "|x| parking fits the wave's exchange area");
  static constexpr int kStashBytes = kBatch * kRow * 4;       // 4 608 / 2 304
  static constexpr int kTw1Bytes = 16 * (kRows - 1) * 16;     // pass 1: [l][kj - 1] (c0, s0, c1, s1)
  static constexpr int kOffTw1 = kWavesPerWG * (kExBytes + kStashBytes), kOffTw2 = kOffTw1 + kTw1Bytes;
  static constexpr int kLdsBytes = kOffTw2 + kTw2Bytes;       // 148 480 / 146 432
  static_assert(kLdsBytes <= 163840, "one workgroup per CU");
};

// sum / maximum over the 16 lanes of a DPP row, the result in every lane of the row
__device__ __forceinline__ float row_sum(float v) {
  v += dpp<kQuadXor1>(v);
  v += dpp<kQuadXor2>(v);
  v += dpp<kRowHalfMirror>(v);
  v += dpp<kRowMirror>(v);
  return v;
}
__device__ __forceinline__ float row_max(float v) {
  v = __builtin_fmaxf(v, dpp<kQuadXor1>(v));
  v = __builtin_fmaxf(v, dpp<kQuadXor2>(v));
  v = __builtin_fmaxf(v, dpp<kRowHalfMirror>(v));
  v = __builtin_fmaxf(v, dpp<kRowMirror>(v));
  return v;
}

// The statistics sweep of amcx_wave_kernel.h (StatsT) for a frame that lives in one 16-lane row: the right neighbour of
// (j, b = 1) is (j, b = 0) of lane l + 1, or (j + 1, b = 0) of lane 0 for lane 15; the shifts are the means of the row's
// first 16 values.
struct RowLanes {
  static constexpr float kInvLanes = 1.0f / 16.0f;
  static __device__ __forceinline__ float sum_all(float v) { return row_sum(v); }
  static __device__ __forceinline__ float next_lane(float v) { return dpp<kRowRor15>(v); }
  static __device__ __forceinline__ bool is_last(int lane) { return (lane & 15) == 15; }
};
using RowStats = StatsT<RowLanes>;

typedef float v4f __attribute__((ext_vector_type(4)));

template <int N>
__global__ __launch_bounds__(SCfg<N>::kThreads, SCfg<N>::kWavesPerWG / 4) void amcx_features18_short_kernel(
    const float2* __restrict__ iq, long long n_frames, long long row_stride,
    float* __restrict__ out, long long out_stride) {
  using C = SCfg<N>;
  constexpr int kN = N, kRows = C::kRows, kRounds = C::kRounds, kWavesPerWG = C::kWavesPerWG, kExBytes = C::kExBytes;
  constexpr int kStashBytes = C::kStashBytes, kOffTw1 = C::kOffTw1, kOffTw2 = C::kOffTw2, kFrameStride = C::kFrameStride;
  constexpr int kBatchPasses = C::kBatchPasses, kHead = C::kHead, kBlocks = C::kBlocks;
  extern __shared__ float4 amcx_short_smem[];
  char* const smem = reinterpret_cast<char*>(amcx_short_smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fq = lane >> 4, l = lane & 15;                   // frame of the pass, lane of the row
  float2* const ex = reinterpret_cast<float2*>(smem + wave * kExBytes) + fq * kFrameStride;   // this frame's exchange block
  float* const stash = reinterpret_cast<float*>(smem + kWavesPerWG * kExBytes + wave * kStashBytes);

  // ---- the twiddles of passes 1 and 2, tabulated once per workgroup (in registers they cost 26 VGPRs and the next pass's
  // prefetched rows went to scratch straight from their loads) ----
  // pass 1: W_N^(m kj), m = 2 l + b, kj = 1 .. kRows - 1;  pass 2: W_32^(a kc), a < 4, kc < 8
  {
    float4* const tw1 = reinterpret_cast<float4*>(smem + kOffTw1);
    float2* const tw2 = reinterpret_cast<float2*>(smem + kOffTw2);
    constexpr int kT1 = 16 * (kRows - 1);
    if (tid < kT1) {
      const int tl = tid / (kRows - 1), kj = tid % (kRows - 1) + 1;
      float s0, c0, s1, c1;
      sincospif((float)((2 * tl) * kj) * (2.0f / (float)kN), &s0, &c0);
      sincospif((float)((2 * tl + 1) * kj) * (2.0f / (float)kN), &s1, &c1);
      tw1[tid] = make_float4(c0, -s0, c1, -s1);
    } else if (tid < kT1 + 32) {
      const int e = tid - kT1;                               // a * 8 + kc
      float sn, cs;
      sincospif((float)((e >> 3) * (e & 7)) * (2.0f / 32.0f), &sn, &cs);
      tw2[e] = make_float2(cs, -sn);
    }
    __syncthreads();
  }
  const float4* const tw1_l = reinterpret_cast<const float4*>(smem + kOffTw1) + (kRows - 1) * l;
  const float2* const tw2_l = reinterpret_cast<const float2*>(smem + kOffTw2) + 8 * (l & 3);
  // exchange addresses (complex elements within the frame's block)
  const int kjL = l >> 2, aL = l & 3;                          // the (kj [+ 4 r], a) / (kj, h) this lane becomes after a transpose
  float2* const ex1_w = ex + 2 * l;                            // + kj * kKjStride: (y[kj][2 l], y[kj][2 l + 1])
  const float2* const ex1_r = ex + kjL * kKjStride + aL;       // + 4 c
  float2* const ex2_w = ex + kjL * kKjStride + aL;             // + 4 kc
  const float2* const ex2_r = ex + kjL * kKjStride + 8 * aL;   // kc = 2 h + e, a: + 4 e + a

  // ---- work: passes of four consecutive frames, kBatchPasses of them a batch; wave g of G owns batches g, g + G, g + 2 G, ...
  // INTERLEAVED, not one contiguous run per wave (round 5): a container holds its frames sorted by modulation and SNR,
  // the slow paths (exact f5 / f9 of +-pi ties, fp64 moment sums of cancelling cumulants) are taken by 8 % of the frames
  // of one cell and by none of another, and a wave that owned one cell set the launch's length: +60 % at N = 128 on the
  // benchmark's data at a flag rate of 0.7 % (profiles/r6_short_interleave_ab.txt).  v: index into the wave's own passes.
  const long long n_pass = (n_frames + kQuad - 1) / kQuad;
  const long long n_waves = (long long)gridDim.x * kWavesPerWG;
  const long long gw = (long long)blockIdx.x * kWavesPerWG + wave;
  const long long n_batches = (n_pass + kBatchPasses - 1) / kBatchPasses;
  const long long my_batches = n_batches > gw ? (n_batches - gw - 1) / n_waves + 1 : 0;
  long long my_passes = my_batches * kBatchPasses;
  if (my_batches > 0 && (my_batches - 1) * n_waves + gw == n_batches - 1) my_passes -= n_batches * kBatchPasses - n_pass;   // the last, short batch
  auto pass_of = [&](long long v) -> long long {
    const long long k = v / kBatchPasses;
    return (k * n_waves + gw) * kBatchPasses + (v - k * kBatchPasses);
  };

  // rows [FIRST, FIRST + COUNT) of this lane's frame of pass p
  auto load_rows = [&](auto first, auto& v, long long p) {
    constexpr int FIRST = decltype(first)::value;
    long long f = p * kQuad + fq;
    if (f >= n_frames) f = n_frames - 1;                      // a pass past the end reads the last frame again (not stored)
    const float2* src = iq + f * row_stride + 2 * l;
    static_for<sizeof(v) / sizeof(v[0])>([&](auto jj) {
      constexpr int j = decltype(jj)::value;
      v[j] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(src + 32 * (FIRST + j)));
    });
  };
  using HeadRows = std::integral_constant<int, 0>;
  using TailRows = std::integral_constant<int, kHead>;

  // ---- batch finalisation: lane g turns stash row g into 18 features (fp64) ----
  auto finalise = [&](long long f0, int count) {
    lds_wave_fence();
    float feat[18];
    float sc = 1.0f, kw_shift = 0.f;
    int ex_half = 0;
    bool tie = false, cancel = false;
    if (lane < count) {
      const float* row = stash + lane * kRow;
      {                                                       // fp32, on the stash values, ahead of the fp64 algebra (amcx_math.h)
        float s15[15];
#pragma unroll
        for (int k = 0; k < 15; ++k) s15[k] = row[k];
        cancel = cancellation_suspect(s15, (float)kN, (float)cancel_kappa(kN));
      }
      FrameSums F;
      F.sA = row[0]; F.sBh = row[1]; F.sP = row[2]; F.sAA = row[3]; F.sX4 = row[4]; F.sAB = row[5];
      F.sAP = row[6]; F.sBP = row[7]; F.sAAA = row[8]; F.sABB = row[9]; F.sAAB = row[10];
      F.sBBB = row[11]; F.sAAP = row[12]; F.sX4P = row[13]; F.sABP = row[14];
      F.sa = row[15]; F.sad1 = row[16]; F.sad2 = row[17]; F.sad4 = row[18];
      F.std1 = row[19]; F.std2 = row[20]; F.sab1 = row[21]; F.sab2 = row[22];
      F.swd1 = row[23]; F.swd2 = row[24]; F.swd3 = row[25]; F.swd4 = row[26];
      F.gmax_raw = row[27]; F.Kt = row[28]; F.Kw = row[29]; F.Ka = row[30];
      F.pi_tie = row[31] != 0.0f;
      kw_shift = row[29];
      const int ex_f = (int)row[32];
      cancel = finalize_features<true>(F, kN, feat, ex_f) && cancel;
      sc = __builtin_bit_cast(float, (127 - ex_f) << 23);     // the 2^-ex the frame was multiplied by
      ex_half = ex_f / 2;
      // flagged by the sweep (f5 came back negated) and not NaN
      tie = __builtin_signbitf(feat[4]) && feat[4] == feat[4] && feat[4] != -__builtin_inff();
    }
    unsigned long long ties = __builtin_amdgcn_ballot_w64(tie);
    while (ties != 0) {                                       // phase steps within an fp32 rounding of +-pi: exact f5 / f9
      const int idx = __builtin_ctzll(ties);
      ties &= ties - 1;
      const float sct = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sc), idx));
      const float kwt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, kw_shift), idx));
      float f5x, f9x;
      wave_exact_frequency<kN>(iq + (f0 + idx) * row_stride, sct, kwt, lane, f5x, f9x);
      if (lane == idx) { feat[4] = f5x; feat[8] = f9x; }
    }
    if (lane < count) {
      float* dst = out + (f0 + lane) * out_stride;
#pragma unroll
      for (int j = 0; j < 18; ++j) dst[j] = feat[j];
    }
    unsigned long long cz = __builtin_amdgcn_ballot_w64(cancel);
    while (cz != 0) {                                         // a cumulant that cancels below what fp32 sums resolve: ids 10-18 from fp64 sums, over the stored row
      const int idx = __builtin_ctzll(cz);
      cz &= cz - 1;
      const float sct = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sc), idx));
      const int hx = __builtin_amdgcn_readlane(ex_half, idx);
      wave_exact_cumulants<kN>(iq + (f0 + idx) * row_stride, sct, hx, lane, lane == idx, out + (f0 + idx) * out_stride);
    }
    lds_wave_fence();
  };

  v4f nxt[kHead];
  if (my_passes > 0) load_rows(HeadRows{}, nxt, pass_of(0));
  int in_batch = 0;                                           // passes whose rows are in the stash

  for (long long v = 0; v < my_passes; ++v) {
    const long long p = pass_of(v);
    float xr[2 * kRows], xi[2 * kRows];
    if constexpr (kHead < kRows) {                            // the rows that were not requested a pass ahead
      v4f late[kRows - kHead];
      load_rows(TailRows{}, late, p);
      static_for<kRows - kHead>([&](auto jj) {
        constexpr int j = kHead + decltype(jj)::value;
        xr[2 * j] = late[j - kHead].x; xi[2 * j] = late[j - kHead].y; xr[2 * j + 1] = late[j - kHead].z; xi[2 * j + 1] = late[j - kHead].w;
      });
    }
    static_for<kHead>([&](auto jj) {
      constexpr int j = decltype(jj)::value;
      xr[2 * j] = nxt[j].x; xi[2 * j] = nxt[j].y; xr[2 * j + 1] = nxt[j].z; xi[2 * j + 1] = nxt[j].w;
    });
    if (v + 1 < my_passes) load_rows(HeadRows{}, nxt, pass_of(v + 1));   // lands behind this pass
    __builtin_amdgcn_s_setprio(1);
    // ---- the frame times 2^-ex (exact), ex the even-rounded exponent of its largest component: NaNs drop out of the
    // maximum (the sums carry them), an infinite or all-zero frame keeps 0 ----
    int ex_f = 0;
    {
      float m = 0.f;
      static_for<2 * kRows>([&](auto ee) {
        constexpr int e = decltype(ee)::value;
        m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(xr[e])), __builtin_fabsf(xi[e]));
      });
      m = row_max(m);
      if (m >= 0x1p-125f && m <= 3.4028235e38f) ex_f = (((__builtin_bit_cast(int, m) >> 23) & 0xff) - 127) & ~1;
      const float sc = __builtin_bit_cast(float, (127 - ex_f) << 23);
      static_for<2 * kRows>([&](auto ee) {
        constexpr int e = decltype(ee)::value;
        xr[e] *= sc; xi[e] *= sc;
      });
    }
    // ---- statistics sweep ----
    RowStats S;
    float a[C::kParkA ? 2 : 2 * kRows];
    float2* const park = reinterpret_cast<float2*>(smem + wave * kExBytes) + lane;   // (N = 512) |x| of row j at park[64 j]
    static_for<kRows>([&](auto jj) {
      constexpr int j = decltype(jj)::value;
      constexpr int k = C::kParkA ? 0 : 2 * j;
      S.template row<j == 0, j == kRows - 1>(xr[2 * j], xi[2 * j], xr[2 * j + 1], xi[2 * j + 1], lane, a[k], a[k + 1]);
      if constexpr (C::kParkA) park[64 * j] = make_float2(a[0], a[1]);
    });
    // ---- envelope about the exact mean ----
    {
      const float mu = row_sum(S.sa) * (1.0f / (float)kN);
      if constexpr (C::kParkA) {
        lds_wave_fence();
        static_for<kRows>([&](auto jj) {
          const float2 aa = park[64 * decltype(jj)::value];
          S.envelope(aa.x, mu);
          S.envelope(aa.y, mu);
        });
        lds_wave_fence();                                     // the area is the exchange block again
      } else {
        static_for<2 * kRows>([&](auto ee) { S.envelope(a[decltype(ee)::value], mu); });
      }
    }
    // ---- the row's sums -> stash row (pass, frame): every lane of a row ends with the totals, lane 0 stores them ----
    float* const row = stash + (in_batch * kQuad + fq) * kRow;
    {
      float s[28] = {S.sA, S.sBh, S.sP, S.sAA, S.sX4, S.sAB, S.sAP, S.sBP, S.sAAA, S.sABB,
                     S.sAAB, S.sBBB, S.sAAP, S.sX4P, S.sABP, S.sa, S.sad1, S.sad2, S.sad4,
                     S.st1, S.st2, S.sab1, S.sab2, S.sw1, S.sw2, S.sw3, S.sw4, 0.f};
      static_for<27>([&](auto kk) { s[decltype(kk)::value] = row_sum(s[decltype(kk)::value]); });
      const bool tie = row_max(S.wmax) > kPi - kTieBand;
      if (l == 0) {
        static_for<7>([&](auto qq) {
          constexpr int q = decltype(qq)::value;
          *reinterpret_cast<float4*>(row + 4 * q) = make_float4(s[4 * q], s[4 * q + 1], s[4 * q + 2], s[4 * q + 3]);
        });
        *reinterpret_cast<float4*>(row + 28) = make_float4(S.Kt, S.Kw, S.Ka, tie ? 1.0f : 0.0f);
        row[32] = (float)ex_f;
      }
    }
    __builtin_amdgcn_s_setprio(0);
    // ---- FFT pass 1: radix kRows over j for b = 0, 1 -> y[kj][m], m = 2 l + b, times W_N^(m kj) ----
    {
      float yr[2][kRows], yi[2][kRows];
      static_for<2>([&](auto bb) {
        constexpr int b = decltype(bb)::value;
        static_for<kRows>([&](auto jj) {
          constexpr int j = decltype(jj)::value;
          yr[b][j] = xr[2 * j + b]; yi[b][j] = xi[2 * j + b];
        });
        dif<kRows, 0, kRows>(yr[b], yi[b]);                   // frequency kj at position bitrev(kj)
      });
      float pk = 0.f;
      static_for<C::kHalves>([&](auto hh) {
        constexpr int half = decltype(hh)::value;
        // y[kj][2 l], y[kj][2 l + 1] of this batch's kBlocks values of kj -> the exchange block
        static_for<kBlocks>([&](auto kk) {
          constexpr int jb = decltype(kk)::value, kj = half * kBlocks + jb;
          constexpr int pos = bitrev(kj, C::kLogRows);
          float r0 = yr[0][pos], i0 = yi[0][pos], r1 = yr[1][pos], i1 = yi[1][pos];
          if constexpr (kj != 0) {
            const float4 t = tw1_l[kj - 1];
            const float c0 = t.x, s0 = t.y, c1 = t.z, s1 = t.w;
            const float tr0 = __builtin_fmaf(r0, c0, -(i0 * s0)); i0 = __builtin_fmaf(r0, s0, i0 * c0); r0 = tr0;
            const float tr1 = __builtin_fmaf(r1, c1, -(i1 * s1)); i1 = __builtin_fmaf(r1, s1, i1 * c1); r1 = tr1;
          }
          *reinterpret_cast<float4*>(ex1_w + jb * kKjStride) = make_float4(r0, i0, r1, i1);
        });
        lds_wave_fence();
        // ---- pass 2: lane (kj, a) takes y[kj][4 c + a], c < 8: radix 8 over c, times W_32^(a kc) -> z[kj][kc][a] ----
        // (a lane reads and writes the same 8 places of its block: positions = a mod 4 of block kj)
        static_for<kRounds>([&](auto rr) {
          constexpr int blk = 4 * decltype(rr)::value * kKjStride;   // round r: block l / 4 + 4 r of the batch
          float zr[8], zi[8];
          static_for<8>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            const float2 v = ex1_r[blk + 4 * c];
            zr[c] = v.x; zi[c] = v.y;
          });
          dif<8, 0, 8>(zr, zi);                                 // frequency kc at position bitrev(kc)
          lds_wave_fence();
          static_for<8>([&](auto kk) {
            constexpr int kc = decltype(kk)::value;
            constexpr int pos = bitrev(kc, 3);
            float r = zr[pos], i = zi[pos];
            if constexpr (kc != 0) {
              const float2 t = tw2_l[kc];
              const float c = t.x, s2 = t.y;
              const float tr = __builtin_fmaf(r, c, -(i * s2)); i = __builtin_fmaf(r, s2, i * c); r = tr;
            }
            ex2_w[blk + 4 * kc] = make_float2(r, i);
          });
        });
        lds_wave_fence();
        // ---- pass 3: lane (kj, h) takes z[kj][2 h + e][a], a < 4: radix 4 over a; the frame's peak ----
        static_for<2 * kRounds>([&](auto ee) {
          constexpr int e = decltype(ee)::value & 1, blk = 4 * (decltype(ee)::value >> 1) * kKjStride;
          const float4 v01 = *reinterpret_cast<const float4*>(ex2_r + blk + 4 * e);
          const float4 v23 = *reinterpret_cast<const float4*>(ex2_r + blk + 4 * e + 2);
          float wr[4] = {v01.x, v01.z, v23.x, v23.z}, wi[4] = {v01.y, v01.w, v23.y, v23.w};
          dif<4, 0, 4>(wr, wi);
          static_for<4>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            pk = __builtin_fmaxf(pk, __builtin_fmaf(wr[k], wr[k], wi[k] * wi[k]));
          });
        });
        lds_wave_fence();                                       // the block is free for the next batch / pass
      });
      pk = row_max(pk);
      if (l == 0) row[27] = pk;
    }
    lds_wave_fence();                                         // the exchange block is free for the next pass
    ++in_batch;
    if (in_batch == kBatchPasses || v + 1 == my_passes) {
      const long long batch_f0 = (p - (in_batch - 1)) * kQuad;   // a batch's passes are consecutive
      long long left = n_frames - batch_f0;
      const int count = left < (long long)(in_batch * kQuad) ? (int)left : in_batch * kQuad;
      finalise(batch_f0, count);
      in_batch = 0;
    }
  }
}

template <int N>
inline hipError_t launch_short_n(const float2* iq, int64_t n_frames, int64_t row_stride, float* out,
                                 int64_t out_stride, hipStream_t stream, int cus) {
  using C = SCfg<N>;
  auto kern = amcx_features18_short_kernel<N>;
  static bool lds_attr_set[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = -1; }
  if (dev < 0 || dev >= 64 || !lds_attr_set[dev]) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             C::kLdsBytes);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) lds_attr_set[dev] = true;       // benign race: idempotent
  }
  const int64_t n_pass = (n_frames + kQuad - 1) / kQuad;
  int64_t grid = cus;                                         // persistent: one resident workgroup per CU
  const int64_t need = (n_pass + C::kWavesPerWG - 1) / C::kWavesPerWG;
  if (grid > need) grid = need;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(C::kThreads), C::kLdsBytes, stream, iq, (long long)n_frames,
                     (long long)row_stride, out, (long long)out_stride);
  return hipGetLastError();
}

inline bool short_supports(int frame_size) { return frame_size == 128 || frame_size == 256 || frame_size == 512; }

inline hipError_t launch_short(const float2* iq, int64_t n_frames, int32_t frame_size, int64_t row_stride, float* out,
                               int64_t out_stride, hipStream_t stream, int cus) {
  return frame_size == 128   ? launch_short_n<128>(iq, n_frames, row_stride, out, out_stride, stream, cus)
         : frame_size == 256 ? launch_short_n<256>(iq, n_frames, row_stride, out, out_stride, stream, cus)
                             : launch_short_n<512>(iq, n_frames, row_stride, out, out_stride, stream, cus);
}

}  // namespace shortk
}  // namespace amcx
