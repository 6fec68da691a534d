// N = 16384 and N = 32768: EIGHT / SIXTEEN wavefronts per frame, each holding one 2048-sample block of it in registers.
//
// frame_size is a free integer in the reference (config.py:96) and np.fft.fft takes any length (features.py:68); rounds
// 1-4 stopped at 8192.  This is the quad kernel's scheme (amcx_quad_kernel.h: four waves, one radix-4 stage) carried
// one level further.  A workgroup is a GROUP of W = N / 2048 waves (one workgroup per CU: W = 8 -> 2 waves per SIMD at
// up to 256 VGPRs, W = 16 -> 4 per SIMD at 128).  Wave q loads block q of the frame (samples [2048 q, 2048 q + 2048), the
// N = 2048 kernel's register layout: 16 x global_load_dwordx4, every byte read from HBM once), runs the N = 2048
// statistics sweep on it with its own shifts -- the finaliser re-centres the W blocks' shifted sums in fp64 -- plus the
// one phase step that crosses into the next block (reference: np.diff(np.unwrap(np.angle(x))) over the whole frame,
// features.py:27-31), and the group takes the spectrum by decimation in frequency ACROSS its waves, in stages:
//
//   a stage of radix rho on a sub-sequence z of M blocks (length L = 2048 M), held by M consecutive waves:
//       z_r[m] = W_L^(r m) * sum_j z[m + j L / rho] W_rho^(r j),    r = 0 .. rho-1,   m = 0 .. L / rho - 1
//   FFT_L(z)[rho k + r] = FFT_(L/rho)(z_r)[k].  The wave with block c = j (M / rho) + p of z reads the blocks p + j' (M / rho)
//   of its rho - 1 partners through LDS and leaves with block p of z_(r = j): rho sub-sequences of M / rho blocks
//   each, held by consecutive waves again.  W = 8: radix 2, then 4.  W = 16: radix 4, then 4.  After the last stage every
//   wave holds one 2048-point sequence and runs the N = 2048 register FFT on it (fft_peak<16>, amcx_wave_kernel.h); the
//   frame's spectral peak is the maximum over the W waves (only max |X|^2 is wanted: the order of the bins is nobody's
//   business).
//
// The twiddle W_L^(r m), m = 2048 p + 128 i + 2 l + b (row i, lane l), is applied as two factors: a row factor
// W_L^(r (2048 p + 128 i)) = W_256^(r (16 p + i) 32768 / L), read from a 256-entry table in LDS with a wave-uniform
// address, and the lane's own W_L^(r (2 l + b)) = (W_L^(2 l + b))^r: the base comes from a 128-entry table per stage in LDS
// (one ds_read_b128 per round) and is squared / cubed on the spot -- holding the factors of both stages in registers
// instead (8 VGPRs, all frame long) cost the 128-register W = 16 kernel scratch traffic.  The units W_rho^(r j) are
// 0 / +-1 in scalar registers (radix_round below: branch-free).
// W = 16: as in the quad, a stage goes in two rounds of eight rows (a wave's region of the exchange area holds eight rows of
// 1 KiB in a round and is its FFT scratch afterwards): four barriers per stage, the same in every wave.  W = 8: the LDS
// holds eight regions of 16 KiB, a stage is ONE round of sixteen rows: two barriers per stage, and a block's registers are
// all free between its sweep and the first stage.
//
// Frames outside the fp32 sums' range are found by the finaliser, recorded in a BIT MASK IN LDS (one bit per frame of
// the workgroup's current epoch of <= 2048 frames) and re-run by the whole group at the end of the epoch, multiplied by an
// exact power of two first (the wave kernels' scheme, finalize_features<true>); a row is stored once, final, and
// nothing is ever read back from the caller's result matrix (the quad kernel, N = 8192, has the same kind of mask for
// its workgroup's whole run of frames since late in round 5; before that it marked such frames in band).
//
// LDS per workgroup, W = 16: FFT tables 16 256 + W_256 table 2 048 + lane factors 2 048 + 16 regions of 8 672 + stash 2 x 2 112
// (two frames) + partial sums / mask 384 = 163 712 bytes; W = 8: 16 256 + 2 048 + 2 048 + 8 x 16 384 + stash 8 x 1 056 + 320 = 160 192.
// Algorithmic HBM bytes per frame: 8 N read + 72 written.
#pragma once

#include "amcx_quad_kernel.h"

namespace amcx {
namespace group {

using namespace wave;
using quad::Recentred;
using quad::reduce_store;

constexpr int kBlock = 2048, kRowsB = 16;
using C2 = Cfg<2048>;                                        // the register FFT every wave runs
constexpr int kTabBytes = C2::kT2Bytes + C2::kT3Bytes;
constexpr int kTw256Bytes = 256 * 8;
constexpr int kLaneTwBytes = 2 * 128 * 8;                   // [stage][2 l + b]: W_L^(2 l + b)
constexpr int kMaskWords = 64, kEpochFrames = 32 * kMaskWords;      // frames per epoch = bits of the re-run mask

template <int W>
struct G {
  static_assert(W == 8 || W == 16, "N = 16384 (eight waves) and N = 32768 (sixteen)");
  static constexpr int kN = kBlock * W, kThreads = 64 * W;
  static constexpr int kRadix0 = W == 8 ? 2 : 4;            // first stage, over blocks (W / kRadix0) apart
  static constexpr int kStride0 = W / kRadix0;              // = 4: the second stage is the quad's radix 4 over neighbours
  static_assert(kStride0 == 4, "two stages, the second of radix 4");
  static constexpr int kBatch = W == 16 ? 2 : 8;            // frames finalised together (W = 16: what the LDS left over holds)
  // W = 8: the first HALF of the next frame's block (rows 0-7, 32 registers) is requested before this frame's FFT and lands
  // behind it (all sixteen rows did not fit next to the FFT's own registers and were stored to scratch straight from the
  // load, behind a wait); the second half is requested at the head of the next sweep and arrives while the first is swept.
  // W = 16 (128 registers): nothing ahead of the frame -- a head of eight or of four rows requested between the FFT's two
  // halves (a hook in fft_peak: built, measured at the ISA) costs 90 - 126 spilled values per frame --; rows 0-11 are
  // requested at the head of the sweep, rows 12-15 when rows 0-3 have been swept and released.
  static constexpr int kHeadRows = W == 8 ? 8 : 0;          // rows requested a frame ahead
  static constexpr int kLateSplit = W == 8 ? kRowsB : 12;    // rows [kHeadRows, kLateSplit) are requested before row 0 is swept,
  static constexpr int kLateAt = 4;                          // rows [kLateSplit, 16) before row kLateAt
  // rows of a block exchanged per round of a stage: all sixteen where the LDS holds W regions of 16 KiB (W = 8: two barriers
  // per stage), eight otherwise (W = 16: the region is the FFT exchange buffer's 8 672 bytes, four barriers per stage)
  static constexpr int kRoundRows = W == 8 ? 16 : 8;
  static constexpr int kRounds = kRowsB / kRoundRows;
  static constexpr int kRegionBytes = W == 16 ? 8672 : 16384;
  static constexpr int kStashRow = kStashStride;
  static constexpr int kStashFloats = kBatch * W * kStashRow;         // one buffer
  static constexpr int kOffTw = kTabBytes;
  static constexpr int kOffLaneTw = kOffTw + kTw256Bytes;
  static constexpr int kOffFrames = kOffLaneTw + kLaneTwBytes;
  static constexpr int kOffStash = kOffFrames + W * kRegionBytes;
  static constexpr int kOffMu = kOffStash + kStashFloats * 4;
  static constexpr int kOffMx = kOffMu + W * 4;
  static constexpr int kOffMask = kOffMx + W * 4;
  static constexpr int kLdsBytes = kOffMask + kMaskWords * 4;
  static_assert(kLdsBytes <= 163840, "one workgroup per CU");
  static_assert(kExchangeBytes <= kRegionBytes && kRoundRows * 1024 <= kRegionBytes && kRegionBytes % 16 == 0,
                "a wave's region holds a round's rows and its FFT exchange buffer");
  static_assert(kEpochFrames % kBatch == 0, "an epoch is whole batches");
};

// One round (rows [ROW0, ROW0 + NROWS)) of a radix-RHO stage for the wave whose residue is r (= its own position among the RHO
// partners; wave-uniform).  Every term comes from LDS, the wave's own included (`first`: the region of partner j' = 0,
// `step`: bytes from one partner's region to the next); the results -- block p of z_r -- go to xr / xi.  Row factor:
// tw256[(k0 + i dk) & 255]; lane factor: (lane_tw[2 l + b])^r.
// r is a RUN-TIME value here and the code is branch-free: the units W_RHO^(r j') = 0 / +-1 sit in scalar registers and a
// term is added with four FMAs (exact: the products are) instead of two additions behind a switch on r.  The first version
// had the switch (the quad kernel's way, r a template parameter): at 128 registers the values merged behind its four arms
// -- 32 or 64 of them -- were given stack slots, ~90 scratch stores per frame and wave.
// PACED: two rows' reads in flight at a time (left alone the compiler issues all of a round's ds_read_b128 -- 128
// registers of results -- at its head and parks them in scratch).
template <int RHO, int ROW0, int NROWS, bool PACED>
__device__ __forceinline__ void radix_round(int r, float (&xr)[2 * kRowsB], float (&xi)[2 * kRowsB], const char* first, int step,
                                            int lane, const char* lane_tw, const float2* tw256, int k0, int dk) {
  float ur[RHO], ui[RHO];                                   // W_RHO^(r j') = ur + i ui
  static_for<RHO>([&](auto jj) {
    constexpr int j = decltype(jj)::value;
    const int k = (r * j) & (RHO - 1);
    if constexpr (RHO == 4) {
      ur[j] = (k == 0 ? 1.f : 0.f) - (k == 2 ? 1.f : 0.f);
      ui[j] = (k == 3 ? 1.f : 0.f) - (k == 1 ? 1.f : 0.f);
    } else {
      ur[j] = k == 0 ? 1.f : -1.f;
      ui[j] = 0.f;
    }
  });
  // this lane's factors W_L^(r (2 l + b)), b = 0, 1: the table's base to the power r
  float4 lw;
  {
    const float4 b = *reinterpret_cast<const float4*>(lane_tw + lane * 16);
    const float4 sq = make_float4(__builtin_fmaf(b.x, b.x, -(b.y * b.y)), 2.0f * b.x * b.y,
                                  __builtin_fmaf(b.z, b.z, -(b.w * b.w)), 2.0f * b.z * b.w);
    const float4 cu = make_float4(__builtin_fmaf(sq.x, b.x, -(sq.y * b.y)), __builtin_fmaf(sq.x, b.y, sq.y * b.x),
                                  __builtin_fmaf(sq.z, b.z, -(sq.w * b.w)), __builtin_fmaf(sq.z, b.w, sq.w * b.z));
    lw = r == 0 ? make_float4(1.f, 0.f, 1.f, 0.f) : r == 1 ? b : r == 2 ? sq : cu;
  }
  static_for<NROWS>([&](auto ii) {
    constexpr int i = ROW0 + decltype(ii)::value;
    float a0r = 0.f, a0i = 0.f, a1r = 0.f, a1i = 0.f;
    static_for<RHO>([&](auto jj) {
      constexpr int j = decltype(jj)::value;
      const float4 v = *reinterpret_cast<const float4*>(first + j * step + (i - ROW0) * 1024 + lane * 16);
      a0r = __builtin_fmaf(v.x, ur[j], a0r);
      a0i = __builtin_fmaf(v.y, ur[j], a0i);
      a1r = __builtin_fmaf(v.z, ur[j], a1r);
      a1i = __builtin_fmaf(v.w, ur[j], a1i);
      if constexpr (RHO == 4) {
        a0r = __builtin_fmaf(-v.y, ui[j], a0r);
        a0i = __builtin_fmaf(v.x, ui[j], a0i);
        a1r = __builtin_fmaf(-v.w, ui[j], a1r);
        a1i = __builtin_fmaf(v.z, ui[j], a1i);
      }
    });
    const float2 t = tw256[(k0 + i * dk) & 255];            // the same address in every lane: one broadcast read
    const float b0r = __builtin_fmaf(a0r, t.x, -(a0i * t.y)), b0i = __builtin_fmaf(a0r, t.y, a0i * t.x);
    const float b1r = __builtin_fmaf(a1r, t.x, -(a1i * t.y)), b1i = __builtin_fmaf(a1r, t.y, a1i * t.x);
    xr[2 * i] = __builtin_fmaf(b0r, lw.x, -(b0i * lw.y));
    xi[2 * i] = __builtin_fmaf(b0r, lw.y, b0i * lw.x);
    xr[2 * i + 1] = __builtin_fmaf(b1r, lw.z, -(b1i * lw.w));
    xi[2 * i + 1] = __builtin_fmaf(b1r, lw.w, b1i * lw.z);
    // (what holds the next rows' reads back: an empty asm that CONSUMES this pair of rows' results and clobbers memory --
    //  the reads behind it cannot be issued before it, and it cannot be reached before the arithmetic in front of it is
    //  done.  __builtin_amdgcn_sched_barrier, which touches no memory, does not hold an LDS read back; a bare memory
    //  clobber holds the reads in order but lets all the arithmetic sink behind them.)
    if constexpr (PACED && (i & 1) == 1)
      asm volatile("" : "+v"(xr[2 * i - 2]), "+v"(xi[2 * i - 2]), "+v"(xr[2 * i - 1]), "+v"(xi[2 * i - 1]),
                        "+v"(xr[2 * i]), "+v"(xi[2 * i]), "+v"(xr[2 * i + 1]), "+v"(xi[2 * i + 1]) : : "memory");
  });
}

// Wave priority by section, as in the quad kernel (AMCX_QUAD_PRIO_MASK there): bit 0 the statistics sweep, 1 envelope +
// reduction, 2 the radix rounds, 3 pass 1 of the register FFT (its passes 2-3 drop to 0 in fft_peak).  Measured: masks 0, 1, 3
// and 15 do not differ at either size (profiles/r5_group_prio_ab.txt) -- the one workgroup's waves move in lock step.
#ifndef AMCX_GROUP_PRIO_MASK
#define AMCX_GROUP_PRIO_MASK 1
#endif
#define AMCX_GROUP_PRIO(b) __builtin_amdgcn_s_setprio((AMCX_GROUP_PRIO_MASK >> (b)) & 1)

// The batch finaliser, by the wave with block 0: lane g turns the W stash rows of frame f_first + g into 18 features in
// fp64.  RG true: ONE frame, re-run on a copy multiplied by 2^-ex (the re-run pass at the end of an epoch); its features
// follow from the scaled sums through the scaling laws (finalize_features<true>).  epoch_f0: the frame that owns bit 0 of
// the re-run mask.  NOT inlined: the fp64 algebra over W re-centred blocks wants ~200 registers; inside the frame loop it
// made the 128-register W = 16 kernel spill ~900 values per call -- as a function it has an allocation of its own and the
// loop keeps its.
template <int W, bool RG>
__device__ __attribute__((noinline)) void group_finalise(const float2* __restrict__ iq, long long row_stride,
                                                         float* __restrict__ out, long long out_stride, long long f_first,
                                                         int count, const float* stash, int ex_pow, long long epoch_f0,
                                                         unsigned* redo_mask, int lane) {
  constexpr int kN = G<W>::kN, kStashRow = G<W>::kStashRow;
  float feat[18];
  bool tie = false, marked = false, cancel = false;
  float kw0 = 0.f;
  if (lane < count) {
    const float* rows = stash + lane * W * kStashRow;
    // The sums are fetched GROUP BY GROUP, each right before the features that need it (amcx_math.h: phase_features,
    // envelope_features, moment_features = finalize_features in pieces): all thirty at once, next to the re-centring
    // accumulators, do not fit 128 registers.  z collects 0 * x of every sum: NaN as soon as one is not finite
    // (is_outside_fp32_range's test).
    double z = 0.0;
    auto sm = [&](int k) -> double {                        // a shift-free sum: the W blocks added in fp64
      double t = 0.0;
#pragma unroll
      for (int h = 0; h < W; h += 2) t += (double)rows[h * kStashRow + k] + (double)rows[(h + 1) * kStashRow + k];
      z = __builtin_fma(t, 0.0, z);
      return t;
    };
    [[maybe_unused]] const int hx = ex_pow / 2;             // ex is even: 2^(hx * twice_order) is exact
    auto put = [&](int j, int twice_order, double v) {
      if constexpr (RG) v = __builtin_ldexp(v, hx * twice_order);
      feat[j] = (float)v;
    };
    const double n = (double)kN;
    // ---- shifted sums, re-centred about block 0's shifts; the peak; the tie flag
    const double Kt = rows[28], Kw = rows[29], Ka = rows[30];
    kw0 = rows[29];
    Recentred th, ab, ws;
    float pk = 0.f;
    bool flagged = false;
#pragma unroll 1
    for (int h = 0; h < W; ++h) {
      const float* r = rows + h * kStashRow;
      th.add((double)kBlock, (double)r[28] - Kt, r[19], r[20]);
      ab.add((double)kBlock, (double)r[30] - Ka, r[21], r[22]);
      ws.add(h == W - 1 ? (double)(kBlock - 1) : (double)kBlock, (double)r[29] - Kw, r[23], r[24], r[25], r[26]);
      pk = __builtin_fmaxf(pk, r[27]);
      if (!(r[27] == r[27])) pk = r[27];                    // a NaN peak (non-finite sample) must survive the maximum
      flagged = flagged || r[31] != 0.0f;
    }
    z = __builtin_fma((double)pk, 0.0, z);
    const double sP = sm(2);
    const bool zero_frame = sP <= 2.0 * n * (double)kTinyPower;
    const bool all_nan = !(__builtin_fabs(sP) <= 1.79e308) || !(pk == pk);   // non-finite input anywhere: 18 NaNs (finalize_features)
    if (all_nan) {
#pragma unroll
      for (int j = 0; j < 18; ++j) feat[j] = __builtin_nanf("");
    } else {
      put(0, 4, (double)pk * (1.0 / n));
      phase_features(th.s1, th.s2, ab.s1, ab.s2, n, put);
      frequency_features(Kw, ws.s1, ws.s2, ws.s3, ws.s4, kN, feat[4], feat[8]);
      if (flagged) feat[4] = -feat[4];                      // picked up below (wave_exact_frequency)
    }
    {
      const double sa = sm(15), sad2 = sm(17), sad4 = sm(18);
      if (!all_nan) {
        double sad1 = 0.0;                                  // (not part of the finiteness test)
#pragma unroll
        for (int h = 0; h < W; h += 2) sad1 += (double)rows[h * kStashRow + 16] + (double)rows[(h + 1) * kStashRow + 16];
        envelope_features(sa, sad1, sad2, sad4, zero_frame, n, put);
      }
    }
    {
      const double sA = sm(0), sBh = sm(1), sAA = sm(3), sX4 = sm(4), sAB = sm(5), sAP = sm(6), sBP = sm(7), sAAA = sm(8),
                   sABB = sm(9), sAAB = sm(10), sBBB = sm(11), sAAP = sm(12), sX4P = sm(13), sABP = sm(14);
      if (!all_nan) {
        if (zero_frame) {                                   // the guard's kTinyPower must not leak into |C20| ... |C63| of a zero frame
#pragma unroll
          for (int j = 9; j < 18; ++j) feat[j] = 0.f;
        } else {
          const float s15[15] = {(float)sA, (float)sBh, (float)sP, (float)sAA, (float)sX4, (float)sAB, (float)sAP, (float)sBP,
                                 (float)sAAA, (float)sABB, (float)sAAB, (float)sBBB, (float)sAAP, (float)sX4P, (float)sABP};
          cancel = cancellation_suspect(s15, (float)kN, (float)cancel_kappa(kN));       // fp32, on the summed values (amcx_math.h)
          moment_features(sA, sBh, sP, sAA, sX4, sAB, sAP, sBP, sAAA, sABB, sAAB, sBBB, sAAP, sX4P, sABP, n, put);
        }
      }
    }
    if constexpr (!RG) {
      // is_outside_fp32_range, on the sums as they went by: some sum not finite, or the mean power outside the range the fp32
      // sums are trusted in (an all-zero frame stays) -> re-run by the whole group at the end of the epoch; not stored now
      const bool zeros = zero_frame && th.s2 == 0.0 && ab.s2 == 0.0 && Kt == 0.0;
      const double pbar = sP / n;
      const bool outside = (sP == sP) && ((z != z) || (!zeros && !(pbar >= kRangeLoPower && pbar <= kRangeHiPower)));
      if (outside) {
        marked = true;
        const unsigned bit = (unsigned)(f_first + lane - epoch_f0);
        __hip_atomic_fetch_or(&redo_mask[bit >> 5], 1u << (bit & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    tie = __builtin_signbitf(feat[4]) && feat[4] == feat[4] && !marked;
    cancel = cancel && !marked;
  }
  unsigned long long ties = __builtin_amdgcn_ballot_w64(tie);
  const float sct = RG ? __builtin_bit_cast(float, (127 - ex_pow) << 23) : 1.0f;     // the 2^-ex the frame was multiplied by
  while (ties != 0) {                                       // phase steps within an fp32 ulp of +-pi: exact f5 / f9
    const int idx = __builtin_ctzll(ties);
    ties &= ties - 1;
    const float kwt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, kw0), idx));
    float f5x, f9x;
    wave_exact_frequency<kN>(iq + (f_first + idx) * row_stride, sct, kwt, lane, f5x, f9x);
    if (lane == idx) { feat[4] = f5x; feat[8] = f9x; }
  }
  if (lane < count && !marked) {
    float* dst = out + (f_first + lane) * out_stride;
#pragma unroll
    for (int j = 0; j < 18; ++j) dst[j] = feat[j];
  }
  unsigned long long cz = __builtin_amdgcn_ballot_w64(cancel);
  while (cz != 0) {                                         // a cumulant that cancels below what fp32 sums resolve: ids 10-18 from fp64 sums, over the stored row
    const int idx = __builtin_ctzll(cz);
    cz &= cz - 1;
    wave_exact_cumulants<kN>(iq + (f_first + idx) * row_stride, sct, RG ? ex_pow / 2 : 0, lane, lane == idx, out + (f_first + idx) * out_stride);
  }
}

template <int W>
__global__ __launch_bounds__(64 * W, 1) void amcx_features18_group_kernel(
    const float2* __restrict__ iq, long long n_frames, long long row_stride,
    float* __restrict__ out, long long out_stride) {
  using Cg = G<W>;
  constexpr int kN = Cg::kN, kBatch = Cg::kBatch, kThreads = Cg::kThreads, kRegionBytes = Cg::kRegionBytes;
  constexpr int kStashRow = Cg::kStashRow;
  extern __shared__ float4 amcx_group_smem[];
  char* smem = reinterpret_cast<char*>(amcx_group_smem);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int q = __builtin_amdgcn_readfirstlane(tid >> 6);   // block of the frame this wave takes
  char* t2 = smem;
  char* t3 = smem + C2::kT2Bytes;
  float2* const tw256 = reinterpret_cast<float2*>(smem + Cg::kOffTw);
  char* fb = smem + Cg::kOffFrames;                         // exchange area / FFT scratch, one region per wave
  float* stash_q = reinterpret_cast<float*>(smem + Cg::kOffStash);
  float* mu_part = reinterpret_cast<float*>(smem + Cg::kOffMu);
  float* mx_part = reinterpret_cast<float*>(smem + Cg::kOffMx);
  unsigned* const redo_mask = reinterpret_cast<unsigned*>(smem + Cg::kOffMask);
  for (int e = tid; e < kMaskWords; e += kThreads) redo_mask[e] = 0;

  // ---- tables: the 2048-point register FFT's, and W_256^k ----
  constexpr int R = C2::kFftRows;                           // 16
  build_fft_tables<C2::kFftN>(t2, t3, tid, kThreads);
  for (int e = tid; e < 256; e += kThreads) {
    float sn, cs;
    sincospif((float)e * (1.0f / 128.0f), &sn, &cs);
    tw256[e] = make_float2(cs, -sn);
  }
  // ---- this wave's place in the two stages ----
  // stage 0: sub-sequence = the frame, M = W blocks, radix kRadix0, partners kStride0 = 4 blocks apart
  const int j0 = q / Cg::kStride0, p0 = q % Cg::kStride0;
  // stage 1: sub-sequence z_(j0), 4 blocks held by waves 4 j0 .. 4 j0 + 3, radix 4 over all of them: j1 = p0, p1 = 0
  const int j1 = p0;
  constexpr int kL0 = kN, kL1 = kBlock * Cg::kStride0;       // lengths of the two sub-sequences
  const char* const lw0 = smem + Cg::kOffLaneTw;           // W_L0^(2 l + b) at [lane][b]
  const char* const lw1 = lw0 + 128 * 8;
  for (int e = tid; e < 256; e += kThreads) {
    float sn, cs;
    sincospif((float)(e & 127) * (2.0f / (float)(e < 128 ? kL0 : kL1)), &sn, &cs);
    reinterpret_cast<float2*>(smem + Cg::kOffLaneTw)[e] = make_float2(cs, -sn);
  }
  // row factors: tw256[(k0 + i dk) & 255], k = r (16 p + i) * 32768 / L
  const int dk0 = j0 * (32768 / kL0), k00 = dk0 * 16 * p0;
  const int dk1 = j1 * (32768 / kL1), k01 = 0;
  const char* const first0 = fb + p0 * kRegionBytes;        // stage 0: partners p0 + 4 j'
  constexpr int kStep0 = Cg::kStride0 * kRegionBytes;
  const char* const first1 = fb + (q - p0) * kRegionBytes;  // stage 1: partners 4 j0 + j'
  constexpr int kStep1 = kRegionBytes;
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

  // ---- work: batches of kBatch frames; workgroup w owns a contiguous run of them, taken in epochs ----
  const long long n_batches = (n_frames + kBatch - 1) / kBatch;
  const long long per_wg = (n_batches + gridDim.x - 1) / gridDim.x;
  const long long wb0 = (long long)blockIdx.x * per_wg;
  long long wb1 = wb0 + per_wg;
  if (wb1 > n_batches) wb1 = n_batches;
  constexpr int kEpochBatches = kEpochFrames / kBatch;

  typedef float v4f __attribute__((ext_vector_type(4)));
  // this wave's block of frame f, HBM -> registers: 16 x global_load_dwordx4, every byte read once -> non-temporal
  auto load_block = [&](v4f (&v)[kRowsB], long long f) {
    const float2* src = iq + f * row_stride + q * kBlock + 2 * lane;
    static_for<kRowsB>([&](auto ii) {
      constexpr int i = decltype(ii)::value;
      v[i] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(src + 128 * i));
    });
  };
  constexpr int kRoundRows = Cg::kRoundRows, kRounds = Cg::kRounds;
  // rows [ROW0, ROW0 + kRoundRows) of the block -> this wave's region of the exchange area
  auto publish_rows = [&](const float (&xr)[2 * kRowsB], const float (&xi)[2 * kRowsB], auto row0) {
    constexpr int ROW0 = decltype(row0)::value;
    static_for<kRoundRows>([&](auto ii) {
      constexpr int i = decltype(ii)::value;
      *reinterpret_cast<float4*>(ex + i * 1024 + lane * 16) =
          make_float4(xr[2 * (ROW0 + i)], xi[2 * (ROW0 + i)], xr[2 * (ROW0 + i) + 1], xi[2 * (ROW0 + i) + 1]);
    });
  };
  using Row0 = std::integral_constant<int, 0>;
  using Row8 = std::integral_constant<int, 8>;
  constexpr int kHeadRows = Cg::kHeadRows, kLateSplit = Cg::kLateSplit, kLateAt = Cg::kLateAt;

  // ---- a frame, phases A and B: statistics sweep of this wave's block (v: its 16 rows as loaded; nx = the first sample
  // of the next block), envelope about the exact mean, sums -> stash row (g, q), the two stages: xr / xi leave as this
  // wave's 2048-point sequence.  Eight workgroup barriers (W = 16; four at W = 8), the same in every wave.
  // Register economy (what the 128-register W = 16 kernel lives on; a spilled register is a 256-byte transaction per wave
  // that goes all the way to HBM -- the first version moved 4x the frame's bytes that way): a row of the first half is
  // written to the wave's region as soon as it has been swept and its registers are free from then on -- the envelope's
  // second sweep and the wave's own term of stage 0 read it back from LDS --; LATE: rows 12-15 are requested only then.
  auto phases_ab = [&](auto late_tag, v4f (&v)[kRowsB], const float2* late_src, const float2 nx, float* stash, int g,
                       float (&xr)[2 * kRowsB], float (&xi)[2 * kRowsB]) __attribute__((always_inline)) {
    constexpr bool LATE = decltype(late_tag)::value;
    Stats S;
    // ---- phase A: statistics sweep, rows 0-7 published on the way ----
    asm volatile("; MARK gA");
    AMCX_GROUP_PRIO(0);
    static_for<kRowsB>([&](auto ii) {
      constexpr int i = decltype(ii)::value;
      if constexpr (LATE && i == 0) {
        static_for<kLateSplit - kHeadRows>([&](auto kk) {
          constexpr int k = kHeadRows + decltype(kk)::value;
          v[k] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(late_src + 128 * k));
        });
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (LATE && i == kLateAt && kLateSplit < kRowsB) {
        __builtin_amdgcn_sched_barrier(0);
        static_for<kRowsB - kLateSplit>([&](auto kk) {
          constexpr int k = kLateSplit + decltype(kk)::value;
          v[k] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(late_src + 128 * k));
        });
        __builtin_amdgcn_sched_barrier(0);
      }
      float a0, a1;                                       // |x| is taken again in phase B
      S.template row<i == 0, i == kRowsB - 1>(v[i].x, v[i].y, v[i].z, v[i].w, lane, a0, a1);
      if constexpr (i < kRoundRows) {
        *reinterpret_cast<v4f*>(ex + i * 1024 + lane * 16) = v[i];
      } else {
        xr[2 * i] = v[i].x; xi[2 * i] = v[i].y; xr[2 * i + 1] = v[i].z; xi[2 * i + 1] = v[i].w;
      }
      // A row is FINISHED before the next one starts.  Left alone the compiler runs the sweep as two passes over the rows
      // -- angles and steps of all sixteen first (the serial chain), every sum afterwards -- and parks two angles per row in
      // scratch in between: 32 stores and 32 loads per frame at 128 registers (W = 16).  The empty asm consumes the running sums at
      // the end of each row, so they have to be up to date there.
      {
        asm volatile("" : "+v"(S.sA), "+v"(S.sBh), "+v"(S.sP), "+v"(S.sAA), "+v"(S.sX4), "+v"(S.sAB), "+v"(S.sAP), "+v"(S.sBP),
                          "+v"(S.sAAA), "+v"(S.sABB), "+v"(S.sAAB), "+v"(S.sBBB), "+v"(S.sAAP), "+v"(S.sX4P), "+v"(S.sABP),
                          "+v"(S.sa), "+v"(S.st1), "+v"(S.st2), "+v"(S.sab1), "+v"(S.sab2), "+v"(S.sw1), "+v"(S.sw2),
                          "+v"(S.sw3), "+v"(S.sw4));
        __builtin_amdgcn_sched_barrier(0);
      }
    });
    if (q < W - 1) {
      const float an = __builtin_amdgcn_sqrtf(__builtin_fmaf(nx.x, nx.x, __builtin_fmaf(nx.y, nx.y, kTinyPower)));
      const float w = wrapped_step(fast_angle(nx.x, nx.y, an), S.th_b1_prev);
      if (lane == 63) {                                   // lane 63 holds the block's last sample (row<.., LAST> gave it a null step)
        S.step(w);
        S.wmax = __builtin_fmaxf(S.wmax, __builtin_fabsf(w));
      }
    }
    {
      const float sa_w = wave_sum_l63(S.sa);
      if (lane == 63) mu_part[q] = sa_w;
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                      // (1) rows 0-7 of every block and the envelope partial sums are in LDS
    // ---- phase B: envelope about the exact mean, sums -> stash, the stages ----
    asm volatile("; MARK gB");
    AMCX_GROUP_PRIO(1);
    {
      float mu = 0.f;
#pragma unroll
      for (int h = 0; h < W; h += 4) mu += (mu_part[h] + mu_part[h + 1]) + (mu_part[h + 2] + mu_part[h + 3]);
      mu *= (1.0f / (float)kN);
      static_for<kRoundRows>([&](auto ii) {               // the published rows: back from the wave's region
        constexpr int i = decltype(ii)::value;
        const v4f r = *reinterpret_cast<const v4f*>(ex + i * 1024 + lane * 16);
        S.envelope(__builtin_amdgcn_sqrtf(__builtin_fmaf(r.x, r.x, __builtin_fmaf(r.y, r.y, kTinyPower))), mu);
        S.envelope(__builtin_amdgcn_sqrtf(__builtin_fmaf(r.z, r.z, __builtin_fmaf(r.w, r.w, kTinyPower))), mu);
      });
      static_for<2 * (kRowsB - kRoundRows)>([&](auto ee) {   // the others (W = 16: rows 8-15): registers
        constexpr int e = 2 * kRoundRows + decltype(ee)::value;
        // |x| is taken AGAIN.  Left to itself the compiler recognises the sweep's square roots and keeps them alive across
        // the barrier; the empty asm makes the two operands opaque, so the second square root is really taken.
        asm volatile("" : "+v"(xr[e]), "+v"(xi[e]));
        S.envelope(__builtin_amdgcn_sqrtf(__builtin_fmaf(xr[e], xr[e], __builtin_fmaf(xi[e], xi[e], kTinyPower))), mu);
      });
      float* const row = stash + (g * W + q) * kStashRow;
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
    asm volatile("; MARK gS0");
    AMCX_GROUP_PRIO(2);
    // stage 0 (every term from LDS, the wave's own included: a published row's registers are free)
    radix_round<Cg::kRadix0, 0, kRoundRows, true>(j0, xr, xi, first0, kStep0, lane, lw0, tw256, k00, dk0);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                      // (2) round 1 has been read
    if constexpr (kRounds == 2) {
      publish_rows(xr, xi, Row8{});
      __syncthreads();                                    // (3) rows 8-15 of every block are in LDS
      radix_round<Cg::kRadix0, 8, 8, true>(j0, xr, xi, first0, kStep0, lane, lw0, tw256, k00, dk0);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();                                    // (4) round 2 has been read
    }
    // stage 1: the wave's block of z_(j0)
    asm volatile("; MARK gS1");
    publish_rows(xr, xi, Row0{});
    __syncthreads();                                      // (5)
    radix_round<4, 0, kRoundRows, true>(j1, xr, xi, first1, kStep1, lane, lw1, tw256, k01, dk1);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (kRounds == 2) {
      __syncthreads();                                    // (6)
      publish_rows(xr, xi, Row8{});
      __syncthreads();                                    // (7)
      radix_round<4, 8, 8, true>(j1, xr, xi, first1, kStep1, lane, lw1, tw256, k01, dk1);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                                      // (8) a wave's region is its FFT scratch now
    __builtin_amdgcn_sched_barrier(0);
  };
  // ---- phase C: 2048-point register FFT of the wave's sequence, its peak into the stash row ----
  auto phase_c = [&](float (&xr)[2 * kRowsB], float (&xi)[2 * kRowsB], float* stash, int g) __attribute__((always_inline)) {
    asm volatile("; MARK gC");
    AMCX_GROUP_PRIO(3);
    const float peak = fft_peak<R>(xr, xi, la);
    const float pk = wave_max_l63(peak);
    if (lane == 63) stash[(g * W + q) * kStashRow + kNumSums] = pk;
    __builtin_amdgcn_sched_barrier(0);
  };

  for (long long eb0 = wb0; eb0 < wb1; eb0 += kEpochBatches) {
    const long long eb1 = eb0 + kEpochBatches < wb1 ? eb0 + kEpochBatches : wb1;
    const int n_iters = (int)(eb1 - eb0);
    const long long epoch_f0 = eb0 * kBatch;
    // frame g of this epoch's round `it`, if it exists
    auto frame_at = [&](int it, int g, long long& f) -> bool {
      f = (eb0 + it) * kBatch + g;
      return it < n_iters && f < n_frames;
    };
    v4f nxt[kRowsB];                                        // rows 0 .. kHeadRows-1 hold the NEXT frame's, requested a frame ago
    auto request_head = [&](long long f) {
      const float2* src = iq + f * row_stride + q * kBlock + 2 * lane;
      static_for<kHeadRows>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        nxt[i] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(src + 128 * i));
      });
    };
    if constexpr (kHeadRows > 0) {
      long long f_first;
      if (frame_at(0, 0, f_first)) request_head(f_first);
    }
    for (int it = 0; it < n_iters; ++it) {
      const long long f0 = (eb0 + it) * kBatch;
      const long long left = n_frames - f0;
      const int n_here = left < kBatch ? (int)left : kBatch;
      float* const stash = stash_q;
      for (int g = 0; g < n_here; ++g) {                    // n_here is the same for all of the group's waves: so are the barriers
        float xr[2 * kRowsB], xi[2 * kRowsB];
        // the first sample of the next block: the phase step that crosses the block boundary
        float2 nx = make_float2(1.f, 0.f);
        if (q < W - 1) nx = iq[(f0 + g) * row_stride + (q + 1) * kBlock];
        const float2* src = iq + (f0 + g) * row_stride + q * kBlock + 2 * lane;
        phases_ab(std::true_type{}, nxt, src, nx, stash, g, xr, xi);
        // W = 8: the head of the next frame's block is requested here, before the FFT, and lands behind it
        if constexpr (kHeadRows > 0) {
          long long f_next = 0;                             // (it, g + 1), or the first frame of the next round
          const bool more = g + 1 < n_here ? frame_at(it, g + 1, f_next) : frame_at(it + 1, 0, f_next);
          if (more) request_head(f_next);
        }
        __builtin_amdgcn_sched_barrier(0);
        phase_c(xr, xi, stash, g);
      }
      // The batch's stash rows are complete once every wave has stored its last peak: one barrier, then the wave with
      // block 0 finalises while the others start on the next frame's statistics sweep -- they meet it again at that
      // frame's barrier (1), BEFORE anybody writes a stash row (phase B), so one buffer is enough.
      __syncthreads();
      if (q == 0) {
        asm volatile("; MARK gFin");
        group_finalise<W, false>(iq, row_stride, out, out_stride, f0, n_here, stash, 0, epoch_f0, redo_mask, lane);
      }
    }
    __syncthreads();                                        // the epoch's re-run mask is complete

    // ---- re-run pass: frames outside the fp32 sums' range (never on ordinary data) ----
    // Every wave reads the same 64 mask words from LDS behind the barrier -- lane l word l -- so the list is the same in
    // every wave and the barriers below stay common.  Each marked frame is run again multiplied by 2^-ex (ex the
    // even-rounded exponent of the frame's largest component: every component below 4, no sixth-order product can
    // overflow); the scaled finaliser undoes it through the features' scaling laws.
    asm volatile("; MARK gRedo");
    const unsigned my_word = redo_mask[lane];
    unsigned long long words = __builtin_amdgcn_ballot_w64(my_word != 0);
    if (words != 0) {
      __syncthreads();                                      // every wave holds its copy: the mask can be cleared
      if (q == 0) redo_mask[lane] = 0;
      float* const stash = stash_q;
      while (words != 0) {
        const int wi = __builtin_ctzll(words);
        words &= words - 1;
        unsigned bits = (unsigned)__builtin_amdgcn_readlane((int)my_word, wi);
        while (bits != 0) {
          const int bi = __builtin_ctz(bits);
          bits &= bits - 1;
          const long long f = epoch_f0 + 32 * wi + bi;
          float xr[2 * kRowsB], xi[2 * kRowsB];
          v4f v[kRowsB];
          load_block(v, f);
          float2 nx = make_float2(1.f, 0.f);
          if (q < W - 1) nx = iq[f * row_stride + (q + 1) * kBlock];
          float m = 0.f;                                    // largest |component| of the FRAME: NaNs drop out of the maximum
          static_for<kRowsB>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(v[i].x), __builtin_fabsf(v[i].y))),
                                __builtin_fmaxf(__builtin_fabsf(v[i].z), __builtin_fabsf(v[i].w)));
          });
          m = wave_max_l63(m);
          if (lane == 63) mx_part[q] = m;
          __syncthreads();
          m = 0.f;
#pragma unroll
          for (int h = 0; h < W; ++h) m = __builtin_fmaxf(m, mx_part[h]);
          int ex_pow = 0;                                   // an infinite component keeps 0: the sums go NaN
          if (m >= 0x1p-125f && m <= 3.4028235e38f) ex_pow = (((__builtin_bit_cast(int, m) >> 23) & 0xff) - 127) & ~1;
          ex_pow = __builtin_amdgcn_readfirstlane(ex_pow);
          const float sc = __builtin_bit_cast(float, (127 - ex_pow) << 23);     // 2^-ex, exact
          static_for<kRowsB>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            v[i] *= sc;
          });
          nx.x *= sc; nx.y *= sc;
          phases_ab(std::false_type{}, v, nullptr, nx, stash, 0, xr, xi);
          phase_c(xr, xi, stash, 0);
          __syncthreads();                                  // the W peaks are in the stash
          if (q == 0) group_finalise<W, true>(iq, row_stride, out, out_stride, f, 1, stash, ex_pow, epoch_f0, redo_mask, lane);
          // (the next marked frame's stash rows are written behind ITS barrier (1), which the wave with block 0 joins
          //  only after the finaliser above; mx_part is written again behind that finaliser too -- by wave 0 itself -- or,
          //  by the other waves, read only behind the barrier that follows their writes)
          __syncthreads();
        }
      }
    }
    __syncthreads();                                        // the next epoch's marks go into a cleared mask
  }
}

template <int W>
inline hipError_t launch_group(const float2* iq, int64_t n_frames, int64_t row_stride, float* out,
                               int64_t out_stride, hipStream_t stream, int cus) {
  using Cg = G<W>;
  auto kern = amcx_features18_group_kernel<W>;
  static bool lds_attr_set[64] = {};                        // > 64 KiB of dynamic LDS needs the attribute, once per device
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64 || !lds_attr_set[dev]) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Cg::kLdsBytes);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) lds_attr_set[dev] = true;     // benign race: idempotent
  }
  const int64_t n_batches = (n_frames + Cg::kBatch - 1) / Cg::kBatch;
  int64_t grid = (int64_t)cus;                              // persistent: one resident workgroup per CU
  if (grid > n_batches) grid = n_batches;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(Cg::kThreads), Cg::kLdsBytes, stream, iq,
                     (long long)n_frames, (long long)row_stride, out, (long long)out_stride);
  return hipGetLastError();
}

}  // namespace group
}  // namespace amcx
