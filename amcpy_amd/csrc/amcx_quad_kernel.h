// N = 8192: FOUR wavefronts per frame, each holding one quarter of it in registers.
//
// A 64 KiB frame does not fit one wave's registers.  Rounds 1-2 gave the frame to ONE wave anyway
// (amcx_features18_wave_kernel<8192>): it streamed the frame three times -- statistics sweep, even
// bins, odd bins -- and because 2048 waves x 64 KiB is 8x what the L2s hold, the second and third
// visits came over the fabric: 3.44x the algorithmic bytes at the memory-side counters, 75 spilled
// VGPRs, 28 M frames/s (profiles/r2_n8192_*).  Here every byte is read from HBM once:
//
//   * a workgroup is a QUAD of four waves, one per SIMD, two workgroups per CU; wave q loads quarter q of the
//     frame (samples [2048 q, 2048 q + 2048), the N = 2048 kernel's register layout) and runs the N = 2048
//     statistics sweep on it -- its own shifts (re-centred in fp64 by the finaliser), plus the one phase step
//     that crosses into the next quarter;
//   * the quad exchanges its quarters through LDS for a radix-4 decimation-in-frequency stage:
//       X[4k + r] = FFT_2048( y_r ),   y_r[n] = W_8192^(r n) * sum_q x[n + 2048 q] (-i)^(r q)
//     wave r forms y_r in place of its quarter (r is a compile-time constant of the code path a wave takes, so
//     the (-i)^(rq) are sign flips and swaps and the row twiddles W_64^(r i) are immediates) and runs the
//     N = 2048 register FFT on it (fft_peak<16>, amcx_wave_kernel.h); the frame's spectral peak is the maximum
//     over the four waves.  The exchange goes in two rounds of eight rows (32 KiB each: one ds_write_b128 /
//     three ds_read_b128 per row and wave, linear and conflict-free), so that the area -- which is also the
//     four waves' FFT scratch -- stays at 34 KiB and TWO independent workgroups fit a CU: a first version with
//     one 64 KiB round kept both quads of an 8-wave workgroup in step behind shared barriers and left the waves
//     parked for 32 % of their cycles (9 % in the barrier-free N = 4096 kernel);
//   * the mean envelope crosses the quad through four floats of LDS (the envelope's second sweep is about the
//     exact mean, features.py:82-85).  Four s_barrier per frame, each among four waves;
//   * the next frame's quarter is requested before the current frame's FFT and lands during it;
//   * after four frames the wave holding quarter 0 turns the 4 x 4 stash rows into features in fp64 (one frame
//     per lane) while the others go on: the stash is double-buffered.
//
// LDS per workgroup: 16.3 KB of FFT tables + 34 KiB exchange / scratch + 8.4 KB stash = 59.5 KB; 256 VGPRs:
// two 256-thread workgroups per CU, 2 waves per SIMD, as the N = 4096 kernel.  No global workspace, no
// inter-workgroup communication.
#pragma once

#include "amcx_wave_kernel.h"

namespace amcx {
namespace quad {

using namespace wave;

constexpr int kN = 8192, kQuarter = 2048, kRowsQ = 16;
constexpr int kWavesPerQuad = 4, kThreads = 64 * kWavesPerQuad, kWGsPerCU = 2;
constexpr int kBatch = 4;                                   // frames a quad finalises together
using C2 = Cfg<2048>;                                        // the register FFT every wave runs
constexpr int kTabBytes = C2::kT2Bytes + C2::kT3Bytes;
constexpr int kRoundRows = kRowsQ / 2;                       // rows of a quarter exchanged per round
constexpr int kRegionBytes = 8704;                           // a wave's part of the area: 8 rows of 1 KiB in a round, its FFT exchange buffer afterwards
constexpr int kFrameBytes = kWavesPerQuad * kRegionBytes;
constexpr int kStashRow = kStashStride;                      // floats per (frame, quarter)
constexpr int kStashFloats = kBatch * kWavesPerQuad * kStashRow;       // one buffer of one quad
constexpr int kOffFrames = kTabBytes;
constexpr int kOffStash = kOffFrames + kFrameBytes;
constexpr int kOffMu = kOffStash + 2 * kStashFloats * 4;
constexpr int kOffMx = kOffMu + kWavesPerQuad * 4;          // [wave] largest |component| of its quarter (re-run of an out-of-range frame)
constexpr int kOffFlag = kOffMx + kWavesPerQuad * 4;         // != 0: this workgroup has a frame in its re-run mask
// Re-run mask: bit k = frame k of the workgroup's own contiguous run is outside the fp32 sums' range.  Set by the wave
// with quarter 0 (the only finaliser), read by all four waves behind a barrier: the list is the same in every wave by
// construction, and nothing is read back from the caller's result matrix.  2 KiB cover 16 384 frames per workgroup =
// 8.4 M frames (550 GB) per launch of 512 workgroups; launch_quad cuts longer inputs into several launches.
constexpr int kMaskWords = 512;
constexpr long long kMaskFrames = 32LL * kMaskWords;
constexpr int kOffMask = kOffFlag + 4;
// Cancellation list (round 6): bit k = frame k of the workgroup's own order has a cumulant that cancels below what fp32 sums
// resolve (amcx_math.h: cancellation_suspect).  Same shape and same writer as the re-run mask; the frames on it get their
// fp64 moment sums from ALL FOUR waves in a pass at the end of the launch -- each wave its quarter, the partial sums
// through kOffPart -- instead of from the finalising wave alone while the other three wait at the next barrier
// (that form: -5.8 % against round 5 on the benchmark's data, profiles/r6_quad_cancel_ab.txt).
constexpr int kOffCFlag = kOffMask + kMaskWords * 4;
constexpr int kOffCMask = kOffCFlag + 4;
constexpr int kOffPart = (kOffCMask + kMaskWords * 4 + 7) & ~7;     // [wave][16] doubles
constexpr int kLdsBytes = kOffPart + kWavesPerQuad * 16 * 8;
static_assert(kMaskFrames % kBatch == 0 && 32 % kBatch == 0, "a batch's bits never straddle a mask word");
static_assert(kWGsPerCU * kLdsBytes <= 163840, "two workgroups per CU");
static_assert(kExchangeBytes <= kRegionBytes && kRoundRows * 1024 <= kRegionBytes, "a wave's region holds a round's rows and its FFT exchange buffer");

// (r, i) *= W_64^J, J = 0..63
template <int J>
__device__ __forceinline__ void mul_w64_any(float& r, float& i) {
  if constexpr (J < 32) {
    mul_w64<J>(r, i);
  } else {
    mul_w64<J - 32>(r, i);
    r = -r; i = -i;
  }
}

// acc += v * (-i)^K, K = 0..3
template <int K>
__device__ __forceinline__ void add_rot(float& ar, float& ai, float vr, float vi) {
  if constexpr (K == 0) { ar += vr; ai += vi; }
  else if constexpr (K == 1) { ar += vi; ai -= vr; }        // * (-i): (re, im) -> (im, -re)
  else if constexpr (K == 2) { ar -= vr; ai -= vi; }
  else { ar -= vi; ai += vr; }                              // * (+i)
}

// the quarter in xr / xi (this wave's own: quarter RQ) -> y_RQ in place, the other three quarters from `fb`
template <int RQ, int ROW0>
__device__ __forceinline__ void radix4_stage(float (&xr)[2 * kRowsQ], float (&xi)[2 * kRowsQ], const char* fb, int lane,
                                             const float4 lw) {
  static_for<kRoundRows>([&](auto ii) {
    constexpr int i = ROW0 + decltype(ii)::value;
    float t0r = xr[2 * i], t0i = xi[2 * i], t1r = xr[2 * i + 1], t1i = xi[2 * i + 1];      // own term: (-i)^(RQ*RQ) applied below
    float a0r = 0.f, a0i = 0.f, a1r = 0.f, a1i = 0.f;
    static_for<4>([&](auto qq) {
      constexpr int q = decltype(qq)::value;
      constexpr int K = (RQ * q) & 3;
      if constexpr (q == RQ) {
        add_rot<K>(a0r, a0i, t0r, t0i);
        add_rot<K>(a1r, a1i, t1r, t1i);
      } else {
        const float4 v = *reinterpret_cast<const float4*>(fb + q * kRegionBytes + (i - ROW0) * 1024 + lane * 16);
        add_rot<K>(a0r, a0i, v.x, v.y);
        add_rot<K>(a1r, a1i, v.z, v.w);
      }
    });
    if constexpr (RQ != 0) {
      // W_8192^(RQ n), n = 128 i + 2 l + b:  W_64^(RQ i) (an immediate) times W_8192^(RQ (2l + b)) (this lane's lw)
      mul_w64_any<(RQ * i) & 63>(a0r, a0i);
      mul_w64_any<(RQ * i) & 63>(a1r, a1i);
      xr[2 * i] = __builtin_fmaf(a0r, lw.x, -(a0i * lw.y));
      xi[2 * i] = __builtin_fmaf(a0r, lw.y, a0i * lw.x);
      xr[2 * i + 1] = __builtin_fmaf(a1r, lw.z, -(a1i * lw.w));
      xi[2 * i + 1] = __builtin_fmaf(a1r, lw.w, a1i * lw.z);
    } else {
      xr[2 * i] = a0r; xi[2 * i] = a0i; xr[2 * i + 1] = a1r; xi[2 * i + 1] = a1i;
    }
  });
}

// wave reduction of a frame-quarter's 27 per-lane sums into one stash row: two swap levels, then four
// DPP steps inside the 16-lane rows (the form the wave kernel uses, amcx_wave_kernel.h)
__device__ __forceinline__ void reduce_store(float (&r28)[28], float* row, int lane) {
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
  if ((lane & 15) == 0) {                    // one lane per 16-lane row: rows hold sums 4j + {0, 2, 1, 3}
    const int rsel = lane >> 4;
    float* dst = row + (((rsel & 1) << 1) | (rsel >> 1));
    static_for<7>([&](auto jj) {
      constexpr int j = decltype(jj)::value;
      dst[4 * j] = r7[j];
    });
  }
}

// sum over the quarters of sum (v - K0)^k, k = 1..4, from each quarter's sums about its own shift K_q
// (n_q values each): (v - K0) = (v - K_q) + d,  d = K_q - K0
struct Recentred {
  double s1 = 0, s2 = 0, s3 = 0, s4 = 0;
  __device__ __forceinline__ void add(double n, double d, double q1, double q2, double q3 = 0, double q4 = 0) {
    const double d2 = d * d;
    s1 += q1 + n * d;
    s2 += q2 + 2.0 * d * q1 + n * d2;
    s3 += q3 + 3.0 * d * q2 + 3.0 * d2 * q1 + n * d2 * d;
    s4 += q4 + 4.0 * d * q3 + 6.0 * d2 * q2 + 4.0 * d2 * d * q1 + n * d2 * d2;
  }
};

// Wave priority by section (bit 0 phase A, 1 envelope + reduction, 2 radix-4 rounds, 3 pass 1 of the register FFT; its
// passes 2-3 drop to 0 in fft_peak): a SIMD issues by priority first, see AMCX_PRIO_MASK in amcx_wave_kernel.h.  Here two
// workgroups share a CU = two waves per SIMD, each paced by its quad's four barriers.  Same box, alternating, through
// the library (profiles/r4_wave_priority_ab.txt, section 6): phase A alone at priority 1 +6.8 % (the product), A + B
// +4.9 %, A + B + radix-4 rounds +6.6 %, everything but FFT passes 2-3 +4.9 %, radix-4 rounds alone +1.1 %, FFT pass 1
// alone +0.1 %.  Same instructions, bit-identical results.
#ifndef AMCX_QUAD_PRIO_MASK
#define AMCX_QUAD_PRIO_MASK 1
#endif
#define AMCX_QUAD_PRIO(b) __builtin_amdgcn_s_setprio((AMCX_QUAD_PRIO_MASK >> (b)) & 1)

__global__ __launch_bounds__(kThreads, 2) void amcx_features18_quad_kernel(
    const float2* __restrict__ iq, long long n_frames, long long row_stride,
    float* __restrict__ out, long long out_stride) {
  extern __shared__ float4 amcx_quad_smem[];
  char* smem = reinterpret_cast<char*>(amcx_quad_smem);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int q = __builtin_amdgcn_readfirstlane(tid >> 6);   // quarter of the frame this wave takes = residue r it produces
  char* t2 = smem;
  char* t3 = smem + C2::kT2Bytes;
  char* fb = smem + kOffFrames;                             // exchange area / FFT scratch, one region per wave
  float* stash_q = reinterpret_cast<float*>(smem + kOffStash);
  float* mu_part = reinterpret_cast<float*>(smem + kOffMu);
  float* mx_part = reinterpret_cast<float*>(smem + kOffMx);
  unsigned* const redo_flag = reinterpret_cast<unsigned*>(smem + kOffFlag);
  unsigned* const redo_mask = reinterpret_cast<unsigned*>(smem + kOffMask);
  unsigned* const cancel_flag = reinterpret_cast<unsigned*>(smem + kOffCFlag);
  unsigned* const cancel_mask = reinterpret_cast<unsigned*>(smem + kOffCMask);
  double* const part = reinterpret_cast<double*>(smem + kOffPart);
  if (tid == 0) { *redo_flag = 0; *cancel_flag = 0; }
  for (int w = tid; w < kMaskWords; w += kThreads) { redo_mask[w] = 0; cancel_mask[w] = 0; }

  // ---- tables of the 2048-point register FFT ----
  constexpr int R = C2::kFftRows;                           // 16
  build_fft_tables<C2::kFftN>(t2, t3, tid, kThreads);
  // this lane's twiddles of the radix-4 stage: W_8192^(q (2 l + b)), b = 0, 1
  float4 lw;
  {
    float s0, c0, s1, c1;
    sincospif((float)(q * (2 * lane)) * (2.0f / (float)kN), &s0, &c0);
    sincospif((float)(q * (2 * lane + 1)) * (2.0f / (float)kN), &s1, &c1);
    lw = make_float4(c0, -s0, c1, -s1);
  }
  __syncthreads();

  const int kkL = lane >> 3, n3L = lane & 7;
  char* ex = fb + q * kRegionBytes;                         // this wave's region: a round's rows, then its FFT exchange buffer
  LaneAddr la;
  la.tw2 = t2 + kkL * kTw2Stride;
  la.tw3 = t3 + lane * 8;
  la.ex1_w = ex + lane * 8;
  la.ex1_r = ex + (kkL * kEx1StrideKK + (n3L & 1) * kEx1StrideB + (n3L >> 1)) * 8;
  la.ex2_w = ex + lane * 8;
  la.ex2_r = ex + (n3L * kEx2StrideK2 + kkL * 8) * 8;

  // ---- work: batches of kBatch frames; workgroup w of G owns batches w, w + G, w + 2 G, ... ----
  // (interleaved, not one contiguous run per workgroup as in rounds 3-5: the slow paths -- exact f5 / f9 of +-pi ties, fp64
  //  moment sums of cancelling cumulants -- are taken by several per cent of the frames of one (modulation, SNR) cell
  //  and by none of another, and a workgroup that owned one cell set the launch's length; amcx_wave_kernel.h, wave_body)
  const long long n_batches = (n_frames + kBatch - 1) / kBatch;
  const long long n_wg = gridDim.x, wg = blockIdx.x;
  const int n_iters = (int)(n_batches > wg ? (n_batches - wg - 1) / n_wg + 1 : 0);
  auto first_frame_of = [&](int it) -> long long { return ((long long)it * n_wg + wg) * kBatch; };   // of the workgroup's batch `it`

  // the wave with quarter 0: the batch whose stash rows are complete once the next barrier has been passed
  long long pend_f0 = 0;
  int pend_n = 0, pend_buf = 0;
  // range_tag true: ONE frame, re-run on a copy multiplied by 2^-ex (the re-run pass at the end of the kernel); its
  // features follow from the scaled sums through the scaling laws (finalize_features<true>)
  auto finalise = [&](auto range_tag, long long f_first, int count, const float* stash, int ex) {
    constexpr bool RG = decltype(range_tag)::value;
    float feat[18];
    bool tie = false, marked = false, cancel = false;
    float kw0 = 0.f;
    if (lane < count) {
      const float* rows = stash + lane * kWavesPerQuad * kStashRow;
      auto sm = [&](int k) -> double {                      // shift-free sums: the four quarters added in fp64
        return ((double)rows[k] + (double)rows[kStashRow + k]) + ((double)rows[2 * kStashRow + k] + (double)rows[3 * kStashRow + k]);
      };
      FrameSums F;
      F.sA = sm(0); F.sBh = sm(1); F.sP = sm(2); F.sAA = sm(3); F.sX4 = sm(4); F.sAB = sm(5);
      F.sAP = sm(6); F.sBP = sm(7); F.sAAA = sm(8); F.sABB = sm(9); F.sAAB = sm(10);
      F.sBBB = sm(11); F.sAAP = sm(12); F.sX4P = sm(13); F.sABP = sm(14);
      F.sa = sm(15); F.sad1 = sm(16); F.sad2 = sm(17); F.sad4 = sm(18);
      // shifted sums: re-centred about quarter 0's shifts
      F.Kt = rows[28]; F.Kw = rows[29]; F.Ka = rows[30];
      kw0 = rows[29];
      Recentred th, ab, ws;
      float pk = 0.f;
      bool flagged = false;
#pragma unroll
      for (int h = 0; h < kWavesPerQuad; ++h) {
        const float* r = rows + h * kStashRow;
        th.add((double)kQuarter, (double)r[28] - F.Kt, r[19], r[20]);
        ab.add((double)kQuarter, (double)r[30] - F.Ka, r[21], r[22]);
        ws.add(h == kWavesPerQuad - 1 ? (double)(kQuarter - 1) : (double)kQuarter, (double)r[29] - F.Kw, r[23], r[24], r[25], r[26]);
        pk = __builtin_fmaxf(pk, r[27]);
        if (!(r[27] == r[27])) pk = r[27];                  // a NaN peak (non-finite sample) must survive the maximum
        flagged = flagged || r[31] != 0.0f;
      }
      F.std1 = th.s1; F.std2 = th.s2; F.sab1 = ab.s1; F.sab2 = ab.s2;
      F.swd1 = ws.s1; F.swd2 = ws.s2; F.swd3 = ws.s3; F.swd4 = ws.s4;
      F.gmax_raw = pk;
      F.pi_tie = flagged;
      {                                                     // fp32, on the summed values (amcx_math.h)
        const float s15[15] = {(float)F.sA, (float)F.sBh, (float)F.sP, (float)F.sAA, (float)F.sX4, (float)F.sAB, (float)F.sAP, (float)F.sBP,
                               (float)F.sAAA, (float)F.sABB, (float)F.sAAB, (float)F.sBBB, (float)F.sAAP, (float)F.sX4P, (float)F.sABP};
        cancel = cancellation_suspect(s15, (float)kN, (float)cancel_kappa(kN));
      }
      if constexpr (RG) {
        cancel = finalize_features<true>(F, kN, feat, ex) && cancel;
      } else {
        cancel = finalize_features(F, kN, feat) && cancel;
        // outside the fp32 sums' range: noted in the workgroup's re-run mask (below) and run again by the whole quad
        // in the pass at the end of this kernel, which overwrites the row -- the row is final when the launch is
        marked = is_outside_fp32_range(F, kN);
        cancel = cancel && !marked;
      }
      tie = !marked && __builtin_signbitf(feat[4]) && feat[4] == feat[4] && feat[4] != -__builtin_inff();
    }
    if constexpr (!RG) {
      const unsigned long long mk = __builtin_amdgcn_ballot_w64(marked);       // bits 0 .. count-1
      if (mk != 0 && lane == 0) {
        const unsigned rel = (unsigned)((f_first / kBatch - wg) / n_wg) * kBatch;   // the batch's place in this workgroup's own order: a multiple of kBatch
        redo_mask[rel >> 5] |= (unsigned)mk << (rel & 31);                     // this wave is the mask's only writer
        *redo_flag = 1;
      }
      const unsigned long long cm = __builtin_amdgcn_ballot_w64(cancel);       // cancelling cumulants: the pass at the end of the launch
      if (cm != 0 && lane == 0) {
        const unsigned rel = (unsigned)((f_first / kBatch - wg) / n_wg) * kBatch;
        cancel_mask[rel >> 5] |= (unsigned)cm << (rel & 31);
        *cancel_flag = 1;
      }
    }
    unsigned long long ties = __builtin_amdgcn_ballot_w64(tie);
    const float sct = RG ? __builtin_bit_cast(float, (127 - ex) << 23) : 1.0f;   // the 2^-ex the frame was multiplied by
    while (ties != 0) {                                     // phase steps within an fp32 ulp of +-pi: exact f5 / f9
      const int idx = __builtin_ctzll(ties);
      ties &= ties - 1;
      const float kwt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, kw0), idx));
      float f5x, f9x;
      wave_exact_frequency<kN>(iq + (f_first + idx) * row_stride, sct, kwt, lane, f5x, f9x);
      if (lane == idx) { feat[4] = f5x; feat[8] = f9x; }
    }
    if (lane < count) {
      float* dst = out + (f_first + lane) * out_stride;
#pragma unroll
      for (int j = 0; j < 18; ++j) dst[j] = feat[j];
    }
    if constexpr (RG) {                                     // a re-run frame that cancels as well (rare squared): this wave alone, now
      unsigned long long cz = __builtin_amdgcn_ballot_w64(cancel);
      while (cz != 0) {
        const int idx = __builtin_ctzll(cz);
        cz &= cz - 1;
        wave_exact_cumulants<kN>(iq + (f_first + idx) * row_stride, sct, ex / 2, lane, lane == idx, out + (f_first + idx) * out_stride);
      }
    }
  };

  // frame g of this workgroup's round `it`, if it exists
  auto frame_at = [&](int it, int g, long long& f) -> bool {
    f = first_frame_of(it) + g;
    return it < n_iters && f < n_frames;
  };
  typedef float v4f __attribute__((ext_vector_type(4)));
  // this wave's quarter of frame f, HBM -> registers: 16 x global_load_dwordx4, every byte read once -> non-temporal
  auto load_quarter = [&](v4f (&v)[kRowsQ], long long f) {
    const float2* src = iq + f * row_stride + q * kQuarter + 2 * lane;
    static_for<kRowsQ>([&](auto ii) {
      constexpr int i = decltype(ii)::value;
      v[i] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(src + 128 * i));
    });
  };
  // rows [ROW0, ROW0 + 8) of the quarter -> this wave's region of the exchange area
  auto publish_rows = [&](const float (&xr)[2 * kRowsQ], const float (&xi)[2 * kRowsQ], auto row0) {
    constexpr int ROW0 = decltype(row0)::value;
    static_for<kRoundRows>([&](auto ii) {
      constexpr int i = decltype(ii)::value;
      *reinterpret_cast<float4*>(ex + i * 1024 + lane * 16) =
          make_float4(xr[2 * (ROW0 + i)], xi[2 * (ROW0 + i)], xr[2 * (ROW0 + i) + 1], xi[2 * (ROW0 + i) + 1]);
    });
  };
  auto radix4_round = [&](float (&xr)[2 * kRowsQ], float (&xi)[2 * kRowsQ], auto row0) {
    constexpr int ROW0 = decltype(row0)::value;
    switch (q) {
      case 0: radix4_stage<0, ROW0>(xr, xi, fb, lane, lw); break;
      case 1: radix4_stage<1, ROW0>(xr, xi, fb, lane, lw); break;
      case 2: radix4_stage<2, ROW0>(xr, xi, fb, lane, lw); break;
      default: radix4_stage<3, ROW0>(xr, xi, fb, lane, lw); break;
    }
  };
  using Row0 = std::integral_constant<int, 0>;
  using Row8 = std::integral_constant<int, kRoundRows>;

  // ---- a frame, phases A and B: statistics sweep of this wave's quarter (xr / xi; nx = the first sample of the next
  // quarter), envelope about the exact mean, sums -> stash row (g, q), radix-4 stage in two rounds: xr / xi leave as y_q.
  // Four workgroup barriers, the same in all four waves.
  auto phases_ab = [&](float (&xr)[2 * kRowsQ], float (&xi)[2 * kRowsQ], const float2 nx, float* stash, int g)
      __attribute__((always_inline)) {
    Stats S;
    // ---- phase A: statistics sweep, rows 0-7 published ----
    AMCX_QUAD_PRIO(0);
    static_for<kRowsQ>([&](auto ii) {
      constexpr int i = decltype(ii)::value;
      float a0, a1;                                       // |x| is taken again in phase B: no room to park 8 KB per wave
      S.template row<i == 0, i == kRowsQ - 1>(xr[2 * i], xi[2 * i], xr[2 * i + 1], xi[2 * i + 1], lane, a0, a1);
    });
    publish_rows(xr, xi, Row0{});
    if (q < 3) {
      const float an = __builtin_amdgcn_sqrtf(__builtin_fmaf(nx.x, nx.x, __builtin_fmaf(nx.y, nx.y, kTinyPower)));
      const float w = wrapped_step(fast_angle(nx.x, nx.y, an), S.th_b1_prev);
      if (lane == 63) {                                   // lane 63 holds the quarter's last sample (row<.., LAST> gave it a null step)
        S.step(w);
        S.wmax = __builtin_fmaxf(S.wmax, __builtin_fabsf(w));
      }
    }
    {
      const float sa_w = wave_sum_l63(S.sa);
      if (lane == 63) mu_part[q] = sa_w;
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                      // (1) rows 0-7 of every quarter and the envelope partial sums are in LDS
    // ---- phase B: envelope about the exact mean, sums -> stash, radix-4 stage in two rounds ----
    AMCX_QUAD_PRIO(1);
    {
      const float mu = ((mu_part[0] + mu_part[1]) + (mu_part[2] + mu_part[3])) * (1.0f / (float)kN);
      static_for<2 * kRowsQ>([&](auto ee) {
        constexpr int e = decltype(ee)::value;
        S.envelope(__builtin_amdgcn_sqrtf(__builtin_fmaf(xr[e], xr[e], __builtin_fmaf(xi[e], xi[e], kTinyPower))), mu);
      });
      float* const row = stash + (g * kWavesPerQuad + q) * kStashRow;
      float r28[28] = {S.sA, S.sBh, S.sP, S.sAA, S.sX4, S.sAB, S.sAP, S.sBP, S.sAAA, S.sABB,
                       S.sAAB, S.sBBB, S.sAAP, S.sX4P, S.sABP, S.sa, S.sad1, S.sad2, S.sad4,
                       S.st1, S.st2, S.sab1, S.sab2, S.sw1, S.sw2, S.sw3, S.sw4, 0.f};
      const unsigned long long tie = __builtin_amdgcn_ballot_w64(S.wmax > kPi - kTieBand);
      reduce_store(r28, row, lane);
      if (lane == 63) {
        row[kNumSums + 1] = S.Kt;
        row[kNumSums + 2] = S.Kw;
        row[kNumSums + 3] = S.Ka;
        row[kNumSums + 4] = tie != 0 ? 1.0f : 0.0f;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    AMCX_QUAD_PRIO(2);
    radix4_round(xr, xi, Row0{});
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                      // (2) round 1 has been read
    publish_rows(xr, xi, Row8{});
    __syncthreads();                                      // (3) rows 8-15 of every quarter are in LDS
    radix4_round(xr, xi, Row8{});
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                      // (4) round 2 has been read: a wave's region is its FFT scratch now
    __builtin_amdgcn_sched_barrier(0);
  };
  // ---- phase C: 2048-point register FFT of y_q, its peak into the stash row ----
  auto phase_c = [&](float (&xr)[2 * kRowsQ], float (&xi)[2 * kRowsQ], float* stash, int g) __attribute__((always_inline)) {
    AMCX_QUAD_PRIO(3);
    const float peak = fft_peak<R>(xr, xi, la);
    const float pk = wave_max_l63(peak);
    if (lane == 63) stash[(g * kWavesPerQuad + q) * kStashRow + kNumSums] = pk;
    __builtin_amdgcn_sched_barrier(0);
  };

  // Rows 0-7 of the next frame's quarter are requested before this frame's FFT and land behind it; rows 8-15 at the head of
  // the frame itself, and arrive while rows 0-7 are swept.  (Until late in round 5 all sixteen rows were requested before the
  // FFT: with 64 more registers live across it the allocator parked lane addresses in scratch and reloaded them INSIDE the
  // FFT -- and a scratch reload waits on vmcnt(0), i.e. for the sixteen loads in front of it: the prefetch was waited for a
  // few hundred instructions after it had been issued.)
  constexpr int kHead = kRowsQ / 2;
  auto load_rows = [&](auto first, v4f (&v)[kHead], long long f) {
    constexpr int FIRST = decltype(first)::value;
    const float2* src = iq + f * row_stride + q * kQuarter + 2 * lane;
    static_for<kHead>([&](auto ii) {
      constexpr int i = decltype(ii)::value;
      v[i] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(src + 128 * (FIRST + i)));
    });
  };
  using Head = std::integral_constant<int, 0>;
  using Tail = std::integral_constant<int, kHead>;
  v4f nxt[kHead];
  {
    long long f_first;
    if (frame_at(0, 0, f_first)) load_rows(Head{}, nxt, f_first);
  }

  for (int it = 0; it < n_iters; ++it) {
    const long long f0 = first_frame_of(it);
    const long long left = n_frames - f0;
    const int n_here = left < kBatch ? (int)left : kBatch;
    float* const stash = stash_q + (it & 1) * kStashFloats;
    for (int g = 0; g < n_here; ++g) {                      // n_here is the same for the quad's four waves: so are the barriers
      float xr[2 * kRowsQ], xi[2 * kRowsQ];
      // this wave's quarter: rows 0-7 were requested a frame ago, rows 8-15 are requested now
      {
        v4f late[kHead];
        load_rows(Tail{}, late, f0 + g);
        static_for<kHead>([&](auto ii) {
          constexpr int i = decltype(ii)::value;
          xr[2 * i] = nxt[i].x; xi[2 * i] = nxt[i].y; xr[2 * i + 1] = nxt[i].z; xi[2 * i + 1] = nxt[i].w;
          xr[2 * (kHead + i)] = late[i].x; xi[2 * (kHead + i)] = late[i].y;
          xr[2 * (kHead + i) + 1] = late[i].z; xi[2 * (kHead + i) + 1] = late[i].w;
        });
      }
      // the first sample of the next quarter: the phase step that crosses the quarter boundary
      float2 nx = make_float2(1.f, 0.f);
      if (q < 3) nx = iq[(f0 + g) * row_stride + (q + 1) * kQuarter];
      phases_ab(xr, xi, nx, stash, g);
      // The head of the next frame's quarter is requested here, before the FFT, and lands behind it.  A batch to finalise
      // (the wave with quarter 0, once per batch) runs after the FFT, when y_q is dead: its ~200 registers and the eight
      // requested rows fit side by side (with all sixteen rows a frame ahead the request had to wait for it).
      const bool finalise_now = q == 0 && g == 0 && pend_n > 0;
      long long f_next;                                     // (it, g + 1), or the first frame of the next round
      const bool more = g + 1 < n_here ? frame_at(it, g + 1, f_next) : frame_at(it + 1, 0, f_next);
      if (more) load_rows(Head{}, nxt, f_next);
      __builtin_amdgcn_sched_barrier(0);
      phase_c(xr, xi, stash, g);
      if (finalise_now) {                                   // the previous batch: every wave is past its last FFT (barrier 1 of this frame)
        finalise(std::false_type{}, pend_f0, pend_n, stash_q + pend_buf * kStashFloats, 0);
        pend_n = 0;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (q == 0) { pend_f0 = f0; pend_n = n_here; pend_buf = it & 1; }
  }
  __syncthreads();                                          // the last batch's FFT peaks are in the stash
  if (q == 0) {
    if (pend_n > 0) finalise(std::false_type{}, pend_f0, pend_n, stash_q + pend_buf * kStashFloats, 0);
    __threadfence();                                        // this wave's rows are out before the re-run pass writes some of them again
  }
  __syncthreads();

  // ---- re-run pass: frames outside the fp32 sums' range (never on ordinary data: one LDS word says so) ----
  // The finaliser noted them in the workgroup's LDS mask.  All four waves walk that mask -- the same LDS words behind
  // the barrier above, so the same list in every wave, and the barriers stay common -- and run each noted frame again,
  // multiplied by 2^-ex first (ex the even-rounded exponent of the frame's largest component: every component below 4,
  // no sixth-order product can overflow); the scaled finaliser undoes it through the features' scaling laws
  // (finalize_features<true>), as the wave kernels do (amcx_wave_kernel.h, rerun_scaled).  A launch leaves every row
  // final; nothing is read back from the result matrix (until round 5 the note was f5 = -inf in the row itself).
  if (__builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile unsigned*>(redo_flag)) != 0) {
    float* const stash = stash_q;
    const int n_words = (n_iters * kBatch + 31) >> 5;         // bit L: frame L % kBatch of the workgroup's batch L / kBatch
    for (int w = 0; w < n_words; ++w) {
      unsigned todo = __builtin_amdgcn_readfirstlane(reinterpret_cast<volatile unsigned*>(redo_mask)[w]);
      while (todo != 0) {
        const int idx = __builtin_ctz(todo);
        todo &= todo - 1;
        const int local = 32 * w + idx;
        const long long f = first_frame_of(local / kBatch) + local % kBatch;
        float xr[2 * kRowsQ], xi[2 * kRowsQ];
        {
          v4f v[kRowsQ];
          load_quarter(v, f);
          static_for<kRowsQ>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            xr[2 * i] = v[i].x; xi[2 * i] = v[i].y; xr[2 * i + 1] = v[i].z; xi[2 * i + 1] = v[i].w;
          });
        }
        float2 nx = make_float2(1.f, 0.f);
        if (q < 3) nx = iq[f * row_stride + (q + 1) * kQuarter];
        float m = 0.f;                                      // largest |component| of the FRAME: NaNs drop out of the maximum
        static_for<2 * kRowsQ>([&](auto ee) {
          constexpr int e = decltype(ee)::value;
          m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(xr[e])), __builtin_fabsf(xi[e]));
        });
        m = wave_max_l63(m);
        if (lane == 63) mx_part[q] = m;
        __syncthreads();
        m = __builtin_fmaxf(__builtin_fmaxf(mx_part[0], mx_part[1]), __builtin_fmaxf(mx_part[2], mx_part[3]));
        int ex = 0;                                         // an infinite component keeps 0: the sums go NaN
        if (m >= 0x1p-125f && m <= 3.4028235e38f) ex = (((__builtin_bit_cast(int, m) >> 23) & 0xff) - 127) & ~1;
        ex = __builtin_amdgcn_readfirstlane(ex);
        const float sc = __builtin_bit_cast(float, (127 - ex) << 23);       // 2^-ex, exact
        static_for<2 * kRowsQ>([&](auto ee) {
          constexpr int e = decltype(ee)::value;
          xr[e] *= sc; xi[e] *= sc;
        });
        nx.x *= sc; nx.y *= sc;
        phases_ab(xr, xi, nx, stash, 0);
        phase_c(xr, xi, stash, 0);
        __syncthreads();                                    // the four peaks are in the stash
        if (q == 0) finalise(std::true_type{}, f, 1, stash, ex);
        // (the next marked frame's stash rows are written behind ITS barrier (1), which the wave with quarter 0 joins
        //  only after the finaliser above; mx_part is not read by it)
      }
    }
  }

  // ---- cancellation pass: |C20| ... |C63| of the frames on the cancellation list again, from fp64 moment sums ----
  // All four waves walk the list (the same LDS words behind the barrier that ended the frame loop: the same frames in the
  // same order in every wave, so the barriers below are common): wave q sweeps quarter q in fp64 (wave_exact_moments), lane 0
  // of each wave leaves its 15 sums in LDS, the wave with quarter 0 adds the four in a fixed order and overwrites columns 9-17
  // of the row its finaliser stored (same wave, behind two barriers).  Reference: features.py:46-58, 116-185 in complex128.
  if (__builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile unsigned*>(cancel_flag)) != 0) {
    const int n_words = (n_iters * kBatch + 31) >> 5;
    for (int w = 0; w < n_words; ++w) {
      unsigned todo = __builtin_amdgcn_readfirstlane(reinterpret_cast<volatile unsigned*>(cancel_mask)[w]);
      while (todo != 0) {
        const int idx = __builtin_ctz(todo);
        todo &= todo - 1;
        const int local = 32 * w + idx;
        const long long f = first_frame_of(local / kBatch) + local % kBatch;
        double t[15];
        wave_exact_moments<kQuarter>(iq + f * row_stride + q * kQuarter, 1.0f, lane, t);
        if (lane == 0) {
#pragma unroll
          for (int k = 0; k < 15; ++k) part[q * 16 + k] = t[k];
        }
        __syncthreads();
        if (q == 0 && lane == 0) {
          double u[15];
#pragma unroll
          for (int k = 0; k < 15; ++k) u[k] = (part[k] + part[16 + k]) + (part[32 + k] + part[48 + k]);
          float* const dst = out + f * out_stride;
          moment_features(u[0], u[1], u[2], u[3], u[4], u[5], u[6], u[7], u[8], u[9], u[10], u[11], u[12], u[13], u[14], (double)kN,
                          [&](int j, int, double v) { dst[j] = (float)v; });
        }
        __syncthreads();                                    // `part` is free for the next frame
      }
    }
  }
}

inline hipError_t launch_quad(const float2* iq, int64_t n_frames, int64_t row_stride, float* out,
                              int64_t out_stride, hipStream_t stream, int cus) {
  const int64_t full_grid = (int64_t)cus * kWGsPerCU;        // persistent: two resident workgroups per CU
  // a workgroup's re-run mask covers kMaskFrames frames of its own run: longer inputs (more than 8.4 M frames of
  // 64 KiB at 512 workgroups -- beyond one device's memory unless rows overlap) go as several launches
  int64_t per_launch = full_grid * kMaskFrames;
  // tests only: cut at this many frames (any cut is valid; the real one needs more frames than a device holds).  Read ONCE
  // per process (a function-local static: getenv on every launch raced with setenv / putenv from other threads --
  // Python writes os.environ while DeviceFanOut's threads launch with the GIL released)
  static const long long test_split = [] { const char* t = getenv("AMCX_TEST_QUAD_SPLIT"); return t ? atoll(t) : 0LL; }();
  if (test_split >= kBatch && test_split < per_launch) per_launch = test_split / kBatch * kBatch;
  for (int64_t f0 = 0; f0 < n_frames; f0 += per_launch) {
    const int64_t n_here = n_frames - f0 < per_launch ? n_frames - f0 : per_launch;
    const int64_t n_batches = (n_here + kBatch - 1) / kBatch;
    int64_t grid = full_grid;
    if (grid > n_batches) grid = n_batches;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(amcx_features18_quad_kernel, dim3((unsigned)grid), dim3(kThreads), kLdsBytes, stream,
                       iq + f0 * row_stride, (long long)n_here, (long long)row_stride, out + f0 * out_stride,
                       (long long)out_stride);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace quad
}  // namespace amcx
