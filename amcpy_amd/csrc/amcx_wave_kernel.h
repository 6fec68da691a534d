// AMCX_VARIANT_WAVE: one wavefront (64 lanes) per frame, frame held in registers.
// Instantiated for the power-of-two frame sizes 128 ... 4096; N = 8192 runs four waves per frame on this header's
// machinery (amcx_quad_kernel.h).  Experiments and ablations that were built against this header and not adopted
// live in tools/experiments/ (patches against a named commit); nothing of them is compiled here.
//
// Why not "one 256-thread workgroup per frame": this path is bound by the board's power cap
// and by VALU issue, not by HBM (~105 fp32 VALU instructions per sample; DESIGN.md section 4.6, HISTORY.md section
// 4.3), so the design minimises instructions per sample:
//   * 16-64 samples per lane amortise every cross-lane reduction over 4-8x more
//     work than a 256-thread block would (8 samples per lane at N = 2048);
//   * waves never synchronise with each other: no s_barrier in the frame loop,
//     the LDS exchange buffer is private to the wave;
//   * the FFT is three register passes (16 or 8 points over the rows with compile-time
//     twiddles, then 16 and 8 points whose butterflies carry the inter-pass twiddles: 6 FMAs
//     each, twisted_dit) joined by two conflict-free LDS transposes, instead of log2(N)
//     shared-memory radix-2 stages (176 B/sample of LDS traffic -> 32 B);
//   * centred statistics are one-pass shifted sums (shift = mean of the first
//     64 samples' value); the envelope alone needs a second sweep over |x|,
//     because f4 needs mean|a - mu| with the exact mu;
//   * the fp64 scalar algebra that turns 27 sums into 18 features runs once
//     per batch of kFramesPerWave frames with one frame per lane.
//
// Index maps (verified against np.fft.fft with the LDS bank rules in
// tools/wave_fft_model.py).  A frame is ROWS = N/128 rows of 128 samples:
//   load     lane l, row i, b in {0,1}  <- x[128 i + 2 l + b]   (global_load_dwordx4, coalesced)
// Register FFT of NF = 128 R points (R = 16 -> 2048, 8 -> 1024), NF = R * 16 * 8:
//   pass 1   R-point DFT over i   -> k1   (no twiddle here: W_NF^((8 n2 + n3) k1) rides passes 2 and 3)
//   xchg 1   phases g (R = 16: k1 = 2 kk + g, the two independent halves of pass 1's output; R = 8: one phase,
//            k1 = kk):  LDS[kk*136 + b*68 + l]; reader lane l' (kk = l'>>3, n3 = l'&7) takes n2 = 0..15 at
//            [kk*136 + (n3&1)*68 + 4 n2 + (n3>>1)]
//   pass 2   16-point DFT over n2 of z[n2] (W_(NF/8)^k1)^n2 -> k2   (twisted decimation in time, factors T2[slot][15])
//   xchg 2   LDS[k2*65 + 8 kk + n3]; reader lane l'' (kk = l''>>3, k2 = (l''&7) + 8 j)
//   pass 3   8-point DFT over n3 of u[n3] (W_NF^(R k2 + k1))^n3 -> X[k1 + R k2 + 16 R k3]   (factors T3[g][j][7][lane]);
//            only max |X|^2 is kept
// N = 4096 is one radix-2 decimation-in-frequency split in front of that machine:
//   X[2k]   = FFT_2048(x[n] + x[n+2048]),  X[2k+1] = FFT_2048((x[n] - x[n+2048]) W_4096^n).
// The whole 32 KiB frame sits in 128 VGPRs, so that variant runs 2 waves per SIMD
// (8-wave workgroups, 256 VGPRs each) and reads every byte exactly once.  (Round 1c
// kept 3 waves per SIMD and read both halves a second time for the difference
// branch: L2 holds 4 MiB per XCD against 12 MiB of frames in flight, so the second
// read came over the fabric -- FETCH_SIZE 2.0x the algorithmic bytes and 23 % of
// wave-cycles waiting on it, profiles/r1d_n4096_summary.json.)
// (Round 4 built the alternative -- two waves per frame at the N = 2048 kernel's 128 registers and 4 waves per SIMD,
// the halves crossing through LDS: tools/experiments/amcx_pair_kernel.h -- and measured it 1.7 % SLOWER on the same
// box, profiles/r4_pair_vs_wave4096_ab.txt.  Rounds 1-3 also carried a one-wave N = 8192: two splits, the frame
// streamed three times, 3.44x the algorithmic traffic, 75 spilled registers; the quad kernel replaced it in round 3
// and the code was removed in round 4.)
// Every ds_write_b64 / ds_read_b64 / ds_read_b128 of the exchanges and twiddle
// tables is bank-conflict free and addressed as lane_base + immediate.
//
// One workgroup per CU.  N <= 512: 768 threads = 12 waves = 3 per SIMD.  N = 1024 and N = 2048: 1024 threads = 16 waves =
// 4 per SIMD (N = 1024: 127 VGPRs, no second register set for the next frame -- the other waves cover the load; one fp64
// value of the per-batch finaliser in scratch since round 6's predicate;
// N = 2048: 128 VGPRs; two lane-dependent addresses are spilled in the prologue and reloaded once per frame each (at the end of
// the wave reduction and of the FFT; no load is in flight at either place; forming them per frame instead removed the scratch
// traffic and cost 0.4 %, profiles/r6_layout_ab.txt), one fp64 value per batch of four in the finaliser:
// amcpy_amd/csrc/kernel_resources.json; the exchange buffer at its exact 8672 bytes and batches of four frames make
// the LDS fit): 4.3 % fewer SIMD cycles per frame than with 12 waves, of which the power cap takes 2.5 % back
// as clock -- +1.2 ... +1.9 % frames/s for the kernel alone, +0.5 % through the library's step (HISTORY.md section 4.1;
// tools/experiments/r5_lab_branches.patch holds the 12-wave form).
// (Two 6-wave workgroups did NOT co-reside at the 160 VGPRs of round 1a: their waves
// landed 2,2,1,1 on the SIMDs, profiles/r1a.)  LDS per workgroup at N = 2048:
// factor tables 1920 + 14336 B, 16 x 8672 B exchange, 16 x 528 B stash = 159.6 KB.
// Algorithmic HBM bytes per frame: 8*N read + 72 written.
#pragma once

#include <type_traits>
#include <utility>

#include "amcx_math.h"

// WAVE PRIORITY.  A SIMD issues vector instructions by priority first, then by age (MI355X_MICROARCH.md, "Two waves per SIMD").
// Four waves share one here, each somewhere else in its frame: in the statistics / envelope sweeps (dense fp32 arithmetic, every
// issue slot used), in the wave reduction (dependent permlane / DPP chains), or in the FFT (three register passes between LDS
// exchanges: it waits a lot).  With all of them at priority 0 the oldest wave wins every contested slot whatever it is doing.
// Raising a wave to s_setprio 1 for everything EXCEPT its FFT -- bits 0-2 below -- lets the arithmetic-dense phases run at
// full rate and the FFT fill the gaps its own waits leave: same instructions, same results, and, through the library on one
// box in alternating runs (profiles/r4_wave_priority_ab.txt):
//     N = 128 +4.7 %   256 +3.2 %   512 +1.5 %   1024 +4.3 %   2048 +4.4 ... +6.2 %   4096 +5.2 ... +5.9 %
// The level does not matter (1, 2, 3: +6.0 / +5.9 / +5.7 % at N = 2048), the extent does: sweep only +-0; sweep + envelope
// +1.1 %; sweep + envelope + reduction +6.0 % (this); + FFT pass 1 +4.3 %; the reduction alone +0.6 %; + the finaliser: no
// change.  (The opposite choice -- priority for the FFT -- costs 2.9 %, a static priority for half of the waves 3.2 %.)
// Different levels per section (a variant kept in tools/experiments/r5_lab_branches.patch): reduction above envelope above sweep +0.2 ... +0.5 % (noise level), the
// sweep above the others -2.7 %.
// Which sections of a frame run at s_setprio 1: bit 0 statistics sweep, 1 envelope sweep, 2 wave reduction, 3 FFT pass 1,
// 4 FFT passes 2-3, 5 the batch finaliser; N = 4096 only: 6 the radix-2 split stage in front of the two FFTs, 7 pass 1 of
// the second FFT (passes 2-3 of either drop to bit 4's level in fft_peak) -- measured: -2.6 % with the split stage at priority 1,
// -4.4 ... -4.6 % with a pass 1 as well (profiles/r4_wave_priority_ab.txt, section 10): 7 is the product at every size.
#ifndef AMCX_PRIO_MASK
#define AMCX_PRIO_MASK 7
#endif
#define AMCX_PRIO_OF(b) ((AMCX_PRIO_MASK >> (b)) & 1)

namespace amcx {
namespace wave {

constexpr int kNumSums = 27;                    // reduced per-lane sums; the spectral peak rides in slot 27
constexpr int kStashStride = 33;                // 32 floats used per frame; odd -> conflict-free column reads

constexpr int kEx1StrideKK = 136, kEx1StrideB = 68;   // complex units (tools/wave_fft_model.py)
constexpr int kEx2StrideK2 = 65;
constexpr int kExchangeBytes = (7 * kEx1StrideKK + kEx1StrideB + 64) * 8;   // 8672: last complex slot exchange 1 touches; >= 16*65*8 = 8320
// twiddles of the twisted decimation-in-time passes (see fft_peak): stage s of a LEN-point pass
// holds 2^s factors, flat index 2^s - 1 + k
constexpr int kTw2Stride = 15 * 8;                     // pass 2: 15 complex per k1 slot (bytes)
constexpr int kTw3Row = 64 * 8;                        // pass 3: [phase][j][7][lane] complex, one row per factor

template <int N>
struct Cfg {
  static_assert(N == 1024 || N == 2048 || N == 4096, "one wave per frame (128 ... 512: amcx_short_kernel.h; 8192 up: quad / group)");
  static constexpr bool kSplit = N == 4096;            // radix-2 DIF split in front of a 2048 FFT
  // frames per wave per batch, finalised together one frame per lane
  // (N = 2048 runs 16 waves per CU, below: batches of four are what its LDS then holds)
  static constexpr int kFramesPerWave = N == 2048 ? 4 : 8;
  // how many times per frame the per-lane fp32 sums are reduced into the stash and started
  // afresh (the finaliser adds the rows in fp64): a lane that runs 64 samples into one accumulator
  // moves the worst scaled error of the N = 4096 sweep from 5.8e-6 to 6.9e-6
  static constexpr int kFlushes = N == 4096 ? 2 : 1;
  static constexpr int kStashBytes = kFramesPerWave * kFlushes * kStashStride * 4;
  // frames per grab over the last stretch of a workgroup's slice (levels the waves' finish)
  static constexpr int kTailChunk = 2;
  // frames per interleaved run of a workgroup (wave_body: work distribution)
  static constexpr int kRunFrames = kFramesPerWave;
  // waves per workgroup = waves per CU: 2 per SIMD when the frame alone takes 128 VGPRs (N = 4096),
  // 4 per SIMD at N = 2048 and N = 1024: the kernels fit 128 VGPRs there (2048: four spilled
  // dwords, 1024: two -- kernel_resources.json).  N = 2048: the fourth wave hides 4.3 % of the SIMD's cycles, +0.5 %
  // through the library's step in round 3 (profiles/r3_waves16_ab.txt), +2.5 % under wave priority
  // (profiles/r4_wave_priority_ab.txt, section 7).  N = 1024 (round 4): 16 waves AND no second register set, +2.9 %
  // same box through the library (section 8); 16 waves with the prefetch kept spill 37 registers, -2.2 %.
  // (Until late in round 5 this kernel also ran 128 ... 512 -- 8 / 4 / 2 frames sharing one run of FFT passes 2-3, twelve
  // waves per CU, a second register set for the next frame: tools/experiments/r5_wave_short_frames.patch has those paths.)
  static constexpr int kWavesPerWG = kSplit ? 8 : 16;
  static constexpr int kThreads = 64 * kWavesPerWG;
  static constexpr int kTailFrames = kWavesPerWG * kFramesPerWave;
  static constexpr int kRows = N / 128;                // rows of 128 samples per frame
  static constexpr int kHeldRows = kRows;              // rows a wave holds in registers
  static constexpr int kFftRows = kSplit ? 16 : kRows; // R: rows one register FFT holds
  static constexpr int kFftN = 128 * kFftRows;
  static constexpr int kPhases = kFftRows >= 8 ? kFftRows / 8 : 1;   // exchange phases of 8 k1 slots
  static constexpr int kT2Bytes = 8 * kPhases * kTw2Stride;          // [k1 slot][15] complex
  static constexpr int kT3Bytes = kPhases * 2 * 7 * kTw3Row;         // [phase][j][7][lane] complex
  static constexpr int kT4Bytes = kSplit ? 64 * 16 : 0;       // [lane][b] complex: W_4096^(2l+b)
  static constexpr int kTableBytes = kT2Bytes + kT3Bytes + kT4Bytes;
  static constexpr int kCounterOffset = kTableBytes + kWavesPerWG * (kExchangeBytes + kStashBytes);
  static constexpr int kLdsBytes = kCounterOffset + 16;       // + the workgroup's two work counters
  static_assert(kLdsBytes <= 163840, "one workgroup per CU must fit in 160 KiB of LDS");
  // the envelope's second sweep reads |x| back from the exchange buffer, or, when the frame is too
  // large to park there, takes the square roots again from the registers
  static_assert(kFramesPerWave <= 64, "one frame per lane in the finaliser");
  static_assert(kSplit || 2 * kRows * 64 * 4 <= kExchangeBytes, "|x| parking must fit the exchange buffer");
};

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

// cos / sin of 2*pi*j/32, j = 0..15
constexpr float kC32[16] = {1.f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f,
                            0.70710678118654752f, 0.55557023301960218f, 0.38268343236508977f, 0.19509032201612825f,
                            0.f, -0.19509032201612825f, -0.38268343236508977f, -0.55557023301960218f,
                            -0.70710678118654752f, -0.83146961230254524f, -0.92387953251128674f, -0.98078528040323043f};
constexpr float kS32[16] = {0.f, 0.19509032201612825f, 0.38268343236508977f, 0.55557023301960218f,
                            0.70710678118654752f, 0.83146961230254524f, 0.92387953251128674f, 0.98078528040323043f,
                            1.f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f,
                            0.70710678118654752f, 0.55557023301960218f, 0.38268343236508977f, 0.19509032201612825f};

// (r, i) *= W_32^J = cos - i sin (J = 0..15), trivial cases folded at compile time
template <int J>
__device__ __forceinline__ void mul_w32(float& r, float& i) {
  constexpr float h = 0.70710678118654752f;
  if constexpr (J == 0) {
  } else if constexpr (J == 8) {            // -i
    const float t = r; r = i; i = -t;
  } else if constexpr (J == 4) {            // (1 - i)/sqrt2
    const float t = (r + i) * h; i = (i - r) * h; r = t;
  } else if constexpr (J == 12) {           // (-1 - i)/sqrt2
    const float t = (i - r) * h; i = (-r - i) * h; r = t;
  } else {
    constexpr float c = kC32[J], s = kS32[J];
    const float t = __builtin_fmaf(r, c, i * s);
    i = __builtin_fmaf(i, c, -(r * s));
    r = t;
  }
}

// (r, i) *= W_64^J = exp(-2 pi i J / 64), J = 0..31 (the N = 8192 split)
constexpr float kCosPi32[17] = {1.f, 0.99518472667219689f, 0.98078528040323043f, 0.95694033573220882f,
                                0.92387953251128674f, 0.88192126434835503f, 0.83146961230254524f,
                                0.77301045336273696f, 0.70710678118654752f, 0.63439328416364549f,
                                0.55557023301960218f, 0.47139673682599764f, 0.38268343236508977f,
                                0.29028467725446236f, 0.19509032201612825f, 0.09801714032956060f, 0.f};
template <int J>
__device__ __forceinline__ void mul_w64(float& r, float& i) {
  static_assert(J >= 0 && J < 32, "upper half-plane only");
  if constexpr (J % 2 == 0) {
    mul_w32<J / 2>(r, i);
  } else {
    constexpr float c = J <= 16 ? kCosPi32[J] : -kCosPi32[32 - J];
    constexpr float s = kCosPi32[J <= 16 ? 16 - J : J - 16];
    const float t = __builtin_fmaf(r, c, i * s);
    i = __builtin_fmaf(i, c, -(r * s));
    r = t;
  }
}

// In-place radix-2 decimation-in-frequency DFT of LEN <= 16 points at [OFF, OFF+LEN);
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
      mul_w32<j * (32 / LEN)>(dr, di);
      re[OFF + j + H] = dr;
      im[OFF + j + H] = di;
    });
    dif<H, OFF, R>(re, im);
    dif<H, OFF + H, R>(re, im);
  }
}

// first stage only of dif<LEN, OFF>: afterwards [OFF, OFF+LEN/2) and [OFF+LEN/2, OFF+LEN) are two
// independent half-length problems (even / odd output frequencies)
template <int LEN, int OFF, int R>
__device__ __forceinline__ void dif_stage(float (&re)[R], float (&im)[R]) {
  constexpr int H = LEN / 2;
  static_for<H>([&](auto jj) {
    constexpr int j = decltype(jj)::value;
    const float ar = re[OFF + j], ai = im[OFF + j], br = re[OFF + j + H], bi = im[OFF + j + H];
    re[OFF + j] = ar + br;
    im[OFF + j] = ai + bi;
    float dr = ar - br, di = ai - bi;
    mul_w32<j * (32 / LEN)>(dr, di);
    re[OFF + j + H] = dr;
    im[OFF + j + H] = di;
  });
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
// maximum of non-negative values, valid in lane 63 (lanes without a source row take 0)
__device__ __forceinline__ float wave_max_l63(float v) {
  v = __builtin_fmaxf(v, dpp<kQuadXor1>(v));
  v = __builtin_fmaxf(v, dpp<kQuadXor2>(v));
  v = __builtin_fmaxf(v, dpp<kRowHalfMirror>(v));
  v = __builtin_fmaxf(v, dpp<kRowMirror>(v));
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

// Diagnostic build only (tools/wave_stamps.hip defines AMCX_WAVE_STAMPS): per-section
// s_memtime deltas summed per wave into a buffer nothing else reads.  The product
// build compiles none of this.
// AMCX_WAVE_STAMPS == 2 (tools/wave_clock.hip): the product instruction stream with ONE
// s_memtime / s_memrealtime pair around the whole frame loop -- the in-kernel clock the way
// MI355X_MICROARCH.md "DVFS give-back" (6) prescribes; no per-section stamp executes.
#ifdef AMCX_WAVE_STAMPS
#define AMCX_STAMP_ARG , unsigned long long* __restrict__ stamp_out
#if AMCX_WAVE_STAMPS == 1
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
#define AMCX_STAMP(sec) do { } while (0)
#endif
#else
#define AMCX_STAMP_ARG
#define AMCX_STAMP(sec) do { } while (0)
#endif
constexpr int kStampSections = 8;

// ---------------------------------------------------------------------------
// Per-lane running sums of the statistics sweep (27 values + the three shifts)
// ---------------------------------------------------------------------------
// How a frame's lanes see each other, for the sweep: all 64 lanes of the wave (one wave per frame: the kernels below, the
// quad and group kernels), or one 16-lane DPP row (four frames per wave: amcx_short_kernel.h, RowLanes).
struct WaveLanes {
  static constexpr float kInvLanes = 1.0f / 64.0f;
  static __device__ __forceinline__ float sum_all(float v) { return bcast_l63(wave_sum_l63(v)); }   // the total, in every lane
  static __device__ __forceinline__ float next_lane(float v) { return dpp<kWaveRol1>(v); }          // lane l reads lane l + 1 (last -> first)
  static __device__ __forceinline__ bool is_last(int lane) { return lane == 63; }
};

template <class Lanes>
struct StatsT {
  float sA = 0, sBh = 0, sP = 0, sAA = 0, sX4 = 0, sAB = 0, sAP = 0, sBP = 0;
  float sAAA = 0, sABB = 0, sAAB = 0, sBBB = 0, sAAP = 0, sX4P = 0, sABP = 0;
  float sa = 0, st1 = 0, st2 = 0, sab1 = 0, sab2 = 0, sw1 = 0, sw2 = 0, sw3 = 0, sw4 = 0;
  float sad1 = 0, sad2 = 0, sad4 = 0;
  float Kt = 0, Kw = 0, Ka = 0;
  float th_b1_prev = 0;   // angle of sample (i-1, b=1), waiting for its right neighbour
  float rot_prev = 0;     // wave_rol1(angle(i-1, b=0))
  float wmax = 0;         // largest |step| this lane saw: > pi - kTieBand flags the frame (kTieBand)

  __device__ __forceinline__ void step(float w) {
    const float d = w - Kw, d2 = d * d;
    sw1 += d; sw2 += d2;
    sw3 = __builtin_fmaf(d2, d, sw3);
    sw4 = __builtin_fmaf(d2, d2, sw4);
  }

  // one row = samples (i, b=0) and (i, b=1) of this lane; returns |x| of both
  template <bool FIRST, bool LAST>
  __device__ __forceinline__ void row(float re0, float im0, float re1, float im1, int lane,
                                      float& a0, float& a1) {
    float th[2];
    const float res[2] = {re0, re1}, ims[2] = {im0, im1};
    float av[2];
    static_for<2>([&](auto bb) {
      constexpr int b = decltype(bb)::value;
      const float re = res[b], im = ims[b];
      const float q = __builtin_fmaf(im, im, kTinyPower);   // zero guard for fast_angle, free in the fma
      const float P = __builtin_fmaf(re, re, q);
      const float A = __builtin_fmaf(re, re, -q);
      const float Bh = re * im;
      const float AA = A * A, BB = Bh * Bh, AP = A * P;
      const float X4 = __builtin_fmaf(-4.0f, BB, AA);       // Re x^4, summed as such (amcx_math.h)
      sA += A; sBh += Bh; sP += P; sAA += AA; sX4 += X4; sAP += AP;
      sAB = __builtin_fmaf(A, Bh, sAB);
      sBP = __builtin_fmaf(Bh, P, sBP);
      sAAA = __builtin_fmaf(AA, A, sAAA);
      sABB = __builtin_fmaf(A, BB, sABB);
      sAAB = __builtin_fmaf(AA, Bh, sAAB);
      sBBB = __builtin_fmaf(BB, Bh, sBBB);
      sAAP = __builtin_fmaf(AA, P, sAAP);
      sX4P = __builtin_fmaf(X4, P, sX4P);
      sABP = __builtin_fmaf(AP, Bh, sABP);
      av[b] = __builtin_amdgcn_sqrtf(P);
      sa += av[b];
      th[b] = fast_angle(re, im, av[b]);
    });
    a0 = av[0]; a1 = av[1];
    if constexpr (FIRST) {
      // shifts: mean over the wave of the first angle / first step
      const float w00 = wrapped_step(th[1], th[0]);
      Kt = Lanes::sum_all(th[0]) * Lanes::kInvLanes;
      Kw = Lanes::sum_all(w00) * Lanes::kInvLanes;
      Ka = Lanes::sum_all(__builtin_fabsf(th[0])) * Lanes::kInvLanes;
    }
    static_for<2>([&](auto bb) {
      constexpr int b = decltype(bb)::value;
      const float d = th[b] - Kt;
      st1 += d;
      st2 = __builtin_fmaf(d, d, st2);
      const float e = __builtin_fabsf(th[b]) - Ka;
      sab1 += e;
      sab2 = __builtin_fmaf(e, e, sab2);
    });
    const float wa = wrapped_step(th[1], th[0]);          // sample (i,0) -> (i,1), same lane
    step(wa);
    const float rot = Lanes::next_lane(th[0]);            // lane l: angle(i,0) of lane l+1 (63 -> lane 0)
    if constexpr (!FIRST) {
      // right neighbour of (i-1, b=1) is (i-1, b=0) of lane l+1, or (i, b=0) of lane 0 for lane 63
      const float nxt = Lanes::is_last(lane) ? rot : rot_prev;
      const float wb = wrapped_step(nxt, th_b1_prev);
      step(wb);
      wmax = __builtin_fmaxf(__builtin_fmaxf(wmax, __builtin_fabsf(wa)), __builtin_fabsf(wb));   // one v_max3
    } else {
      wmax = __builtin_fabsf(wa);
    }
    if constexpr (LAST) {
      // (last row, b=1): lane 63 holds the frame's last sample, which has no step
      const float w = wrapped_step(rot, th[1]);
      const float wc = Lanes::is_last(lane) ? Kw : w;
      step(wc);
      wmax = __builtin_fmaxf(wmax, __builtin_fabsf(wc));
    }
    th_b1_prev = th[1];
    rot_prev = rot;
  }

  // start a new stretch of the sums; shifts, neighbour angles and the tie tracker carry on
  __device__ __forceinline__ void clear_sums() {
    sA = sBh = sP = sAA = sX4 = sAB = sAP = sBP = 0.f;
    sAAA = sABB = sAAB = sBBB = sAAP = sX4P = sABP = 0.f;
    sa = st1 = st2 = sab1 = sab2 = sw1 = sw2 = sw3 = sw4 = 0.f;
  }

  __device__ __forceinline__ void envelope(float a, float mu) {
    const float d = a - mu, d2 = d * d;
    sad1 += __builtin_fabsf(d);
    sad2 += d2;
    sad4 = __builtin_fmaf(d2, d2, sad4);
  }
};
using Stats = StatsT<WaveLanes>;


// LDS addresses that depend only on the lane (bytes)
struct LaneAddr {
  const char* tw2;    // pass-2 twiddles of this lane's k1 slot: T2 + (lane>>3)*kTw2Stride
  const char* tw3;    // pass-3 twiddles: T3 + lane*8
  char* ex1_w;        // exchange + lane*8
  const char* ex1_r;
  char* ex2_w;
  const char* ex2_r;
};

// (a, b) <- (a + t b, a - t b) in 6 FMAs: the second output is 2a - (a + t b)
__device__ __forceinline__ void bfly6(float& ar, float& ai, float& br, float& bi, const float2 t) {
  const float yr = __builtin_fmaf(-t.y, bi, __builtin_fmaf(t.x, br, ar));
  const float yi = __builtin_fmaf(t.y, br, __builtin_fmaf(t.x, bi, ai));
  br = __builtin_fmaf(2.0f, ar, -yr);
  bi = __builtin_fmaf(2.0f, ai, -yi);
  ar = yr;
  ai = yi;
}

// LEN-point DFT of x[m] * w^m, radix-2 decimation in time with the twist folded into every
// butterfly's factor: Y = E + (w W_LEN^k) O, E / O the (w^2-twisted) transforms of the even / odd
// samples.  In place on natural-order storage; frequency k ends at position bitrev(k).
// tw(i) returns factor i of the flat list [stage s: 2^s factors w^(LEN/2^(s+1)) W_(2^(s+1))^k].
// This is how the inter-pass twiddles of the 16 x 16 x 8 decomposition are applied: 6 FMAs per
// butterfly instead of a 4-op complex multiplication per point plus 4 add/sub per butterfly
// (tools/wave_fft_model.py: model_fused).
template <int LEN, class TW>
__device__ __forceinline__ void twisted_dit(float (&re)[LEN], float (&im)[LEN], TW&& tw) {
  constexpr int LOG = LEN == 16 ? 4 : 3;
  static_assert(LEN == 16 || LEN == 8, "pass lengths");
  static_for<LOG>([&](auto ss) {
    constexpr int sidx = decltype(ss)::value;
    constexpr int half = 1 << sidx;             // factors in this stage = L/2
    constexpr int d = LEN / (2 * half);         // distance between the two inputs
    static_for<half>([&](auto kk) {
      constexpr int k = decltype(kk)::value;
      const float2 t = tw(std::integral_constant<int, half - 1 + k>{});   // one factor serves d butterflies
      static_for<d>([&](auto mm) {
        constexpr int m = decltype(mm)::value;
        constexpr int p = m + 2 * d * bitrev(k, sidx);
        bfly6(re[p], im[p], re[p + d], im[p + d], t);
      });
    });
  });
}

// Register FFT of 128*R points held as xr/xi[2*i+b] (see header); returns this
// lane's max |X|^2 over the 2R bins it ends up with.
// In two pieces -- fft_head (the part of pass 1 that comes before the exchange phases) and fft_tail (the phases) -- so that
// a caller with TWO transforms to run (N = 4096: the sum and the difference branch of the radix-2 split) can put the
// second one's pass 1, which touches registers only, into the shadows of the first one's LDS round trips: fft_tail calls
// hook(phase, which) right behind the reads of exchange 1 (which = 0) and of exchange 2 (which = 1) of every phase, before
// it waits for them.  fft_peak = head + tail without a hook: the one-transform kernels' machine code is what it was.
struct NoFftHook {
  template <class G, class W>
  __device__ __forceinline__ void operator()(G, W) const {}
};

template <int R>
__device__ __forceinline__ void fft_head(const float (&xr)[2 * R], const float (&xi)[2 * R], float (&v0r)[R], float (&v0i)[R],
                                         float (&v1r)[R], float (&v1i)[R]) {
  static_assert(R == 8 || R == 16, "1024 or 2048 points (shorter frames: amcx_short_kernel.h)");
  // pass 1 (both b groups): plain DFT over the rows, no twiddle (it rides pass 2 and pass 3).
  // With two exchange phases (R = 16) phase g takes the k1 of parity g: after the first
  // decimation-in-frequency stage those are the two independent halves of the register set, so
  // the odd half waits as 8 + 8 complex values while the even half goes through exchange 1,
  // pass 2, exchange 2 and pass 3 -- 32 registers fewer in flight than with both phases' pass-2
  // inputs read before either is processed.
  static_for<R>([&](auto ii) {
    constexpr int i = decltype(ii)::value;
    v0r[i] = xr[2 * i]; v0i[i] = xi[2 * i]; v1r[i] = xr[2 * i + 1]; v1i[i] = xi[2 * i + 1];
  });
  if constexpr (R / 8 == 2) {
    dif_stage<R, 0>(v0r, v0i);
    dif_stage<R, 0>(v1r, v1i);
  }
}

// kDifDone: the 8-point transforms of pass 1 (dif<8, 8 g> of both b groups) have been done by the caller already
template <int R, bool kDifDone = false, class Hook = NoFftHook>
__device__ __forceinline__ float fft_tail(float (&v0r)[R], float (&v0i)[R], float (&v1r)[R], float (&v1i)[R], const LaneAddr& la,
                                          Hook&& hook = Hook{}) {
  constexpr int PH = R / 8;                  // exchange phases of 8 k1 values
  float peak = 0.f;
  static_for<PH>([&](auto gg) {
    constexpr int gph = decltype(gg)::value;
    if constexpr (!kDifDone) {
      dif<8, 8 * gph>(v0r, v0i);
      dif<8, 8 * gph>(v1r, v1i);
    }
    float zr[16], zi[16];
    lds_wave_fence();
    static_for<8>([&](auto kk_) {
      constexpr int kk = decltype(kk_)::value;
      constexpr int p = 8 * gph + bitrev(kk, 3);        // R = 16: k1 = 2 kk + gph; R = 8: k1 = kk
      *reinterpret_cast<float2*>(la.ex1_w + (kk * kEx1StrideKK) * 8) = make_float2(v0r[p], v0i[p]);
      *reinterpret_cast<float2*>(la.ex1_w + (kk * kEx1StrideKK + kEx1StrideB) * 8) = make_float2(v1r[p], v1i[p]);
    });
    lds_wave_fence();
    static_for<16>([&](auto nn) {
      constexpr int n2 = decltype(nn)::value;
      const float2 v = *reinterpret_cast<const float2*>(la.ex1_r + n2 * 32);
      zr[n2] = v.x; zi[n2] = v.y;
    });
    hook(gg, std::integral_constant<int, 0>{});
    if constexpr (gph == 0) asm volatile("; MARK fft2");
    if constexpr (gph == 0) __builtin_amdgcn_s_setprio(AMCX_PRIO_OF(4));
    __builtin_amdgcn_sched_barrier(0);
    // pass 2 over n2, twist (W_(NF/8)^k1)^n2; exchange 2; pass 3 over n3, twist (W_NF^(R k2 + k1))^n3
    const char* const tw2 = la.tw2;
    twisted_dit<16>(zr, zi, [&](auto ii) {
      return *reinterpret_cast<const float2*>(tw2 + gph * 8 * kTw2Stride + decltype(ii)::value * 8);
    });
    lds_wave_fence();
    static_for<16>([&](auto kk2) {
      constexpr int k2 = decltype(kk2)::value;
      constexpr int p = bitrev(k2, 4);
      *reinterpret_cast<float2*>(la.ex2_w + (k2 * kEx2StrideK2) * 8) = make_float2(zr[p], zi[p]);
    });
    lds_wave_fence();
    const char* const tw3 = la.tw3;
    static_for<2>([&](auto jj) {
      constexpr int j = decltype(jj)::value;
      float ur[8], ui[8];
      static_for<8>([&](auto nn) {
        constexpr int n3 = decltype(nn)::value;
        const float2 v = *reinterpret_cast<const float2*>(la.ex2_r + (j * 8 * kEx2StrideK2 + n3) * 8);
        ur[n3] = v.x; ui[n3] = v.y;
      });
      if constexpr (j == 0) hook(gg, std::integral_constant<int, 1>{});
      twisted_dit<8>(ur, ui, [&](auto ii) {
        return *reinterpret_cast<const float2*>(tw3 + ((gph * 2 + j) * 7 + decltype(ii)::value) * kTw3Row);
      });
      static_for<4>([&](auto pp) {
        constexpr int p = 2 * decltype(pp)::value;
        peak = __builtin_fmaxf(__builtin_fmaxf(peak, __builtin_fmaf(ur[p], ur[p], ui[p] * ui[p])),
                               __builtin_fmaf(ur[p + 1], ur[p + 1], ui[p + 1] * ui[p + 1]));   // v_max3
      });
    });
    if constexpr (gph + 1 < PH) __builtin_amdgcn_sched_barrier(0);   // the other half starts only now
  });
  lds_wave_fence();
  return peak;
}

template <int R>
__device__ __forceinline__ float fft_peak(const float (&xr)[2 * R], const float (&xi)[2 * R],
                                          const LaneAddr& la) {
  float v0r[R], v0i[R], v1r[R], v1i[R];
  fft_head<R>(xr, xi, v0r, v0i, v1r, v1i);
  return fft_tail<R>(v0r, v0i, v1r, v1i, la);
}

// ---------------------------------------------------------------------------
// Twiddle tables of the NF-point register FFT (NF = 128 R: 128 ... 2048), built once per workgroup by all of its
// threads.  W_NF^e = exp(-2 pi i e / NF) with an integer exponent: exact argument reduction.
//   t2 [k1 slot][15]: pass 2 (16 points over n2, slot k1' = 8 g + kk holding frame-local k1 = k1' mod R): factor i of
//      stage s (L = 2^(s+1), d = 16 / L):  (W_(NF/8)^k1)^d W_L^k  =  W_NF^(8 k1 d + k NF / L)
//   t3 [phase][j][7][lane]: pass 3 (8 points over n3; lane = (kk, k2 low), combination c = 2 g + j: k2 = (lane & 7) + 8 j):
//      (W_NF^(R k2 + k1))^d W_L^k = W_NF^((R k2 + k1) d + k NF / L), L = 2^(s+1), d = 8 / L
// Shared by the wave kernels and the quad kernel (N = 8192).
// ---------------------------------------------------------------------------
template <int NF>
__device__ __forceinline__ void build_fft_tables(char* t2, char* t3, int tid, int n_threads) {
  constexpr int R = NF / 128, PH = R >= 8 ? R / 8 : 1;
  auto w_nf = [](int e) {
    float sn, cs;
    sincospif((float)(e & (NF - 1)) * (2.0f / (float)NF), &sn, &cs);
    return make_float2(cs, -sn);
  };
  for (int e = tid; e < 8 * PH * 15; e += n_threads) {
    const int slot = e / 15, i = e % 15;
    const int k1 = PH == 2 ? 2 * (slot & 7) + (slot >> 3) : slot % R;   // two phases: by parity of k1
    const int sidx = i < 1 ? 0 : i < 3 ? 1 : i < 7 ? 2 : 3;
    const int k = i - ((1 << sidx) - 1), Lp = 2 << sidx, d = 16 / Lp;
    reinterpret_cast<float2*>(t2)[e] = w_nf(8 * k1 * d + k * (NF / Lp));
  }
  for (int e = tid; e < PH * 2 * 7 * 64; e += n_threads) {
    const int ln = e & 63, i = (e >> 6) % 7, c = e / (7 * 64);
    const int k1 = PH == 2 ? 2 * (ln >> 3) + (c >> 1) : (ln >> 3) % R, k2 = (ln & 7) + 8 * (c & 1);
    const int sidx = i < 1 ? 0 : i < 3 ? 1 : 2;
    const int k = i - ((1 << sidx) - 1), Lp = 2 << sidx, d = 8 / Lp;
    reinterpret_cast<float2*>(t3)[e] = w_nf((R * k2 + k1) * d + k * (NF / Lp));
  }
}

// ---------------------------------------------------------------------------
// f5 / f9 of ONE frame with the sign of every phase step within kTieBand of +-pi decided exactly
// (exact_step, amcx_math.h), by the whole wave: the slow path behind the statistics sweep's tie flag.
// It runs inside the finaliser -- on ~0.3 % of frames, after the frame's registers are dead -- as a
// ROLLED loop over the frame re-read from memory (L2 / Infinity Cache: the wave read it microseconds
// ago): lane l takes steps l, l + 64, ...; both samples of a step are loaded by the lane itself, so
// nothing crosses lanes until the fp64 reductions (reference: np.std / scipy kurtosis of the float64
// frequency array, features.py:88-91,110-113).  In the
// hot loop this cost 60-760 spilled VGPRs (round 1); as a separate launch it cost a 4-byte-per-frame scan
// of the result matrix plus 29 us per 639 k frames (round 2).  sc: the power of two a re-run
// multiplies the frame by (1 in the throughput kernel).
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <int N>
__device__ __forceinline__ void wave_exact_frequency(const float2* __restrict__ src, float sc, float Kw_f, int lane,
                                                     float& f5, float& f9) {
  // lane l takes steps l + 64 j, eight of them per trip so that their sixteen loads are in flight together:
  // the frame comes back from L2 / Infinity Cache / HBM under the full streaming load of the chip, ~2 us a
  // round trip, and a rolled loop of dependent round trips held the wave 120 us per frame.  ONE sweep: the
  // sums are taken in fp64 about the shift the statistics sweep used (the mean of the frame's first 64 steps),
  // which frequency_features turns into central moments -- in fp64 a shift within a few standard deviations of
  // the mean costs nothing.  The frame's last sample has no step: lane 63's last one is computed on a clamped
  // neighbour and weighted 0.
  constexpr int kPer = N / 64, U = kPer < 8 ? kPer : 8, kTrips = kPer / U;     // N = 128: two steps per lane
  static_assert(kPer >= 1 && kPer % U == 0, "frame sizes are powers of two >= 128");
  const double Kw = (double)Kw_f;
  double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
#pragma unroll 1
  for (int t = 0; t < kTrips; ++t) {
    float2 p[U], q[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int n = lane + 64 * (U * t + u);
      p[u] = src[n];
      q[u] = src[n + 1 < N ? n + 1 : N - 1];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int n = lane + 64 * (U * t + u);
      const float pr = p[u].x * sc, pi = p[u].y * sc, qr = q[u].x * sc, qi = q[u].y * sc;
      const float ap = __builtin_amdgcn_sqrtf(__builtin_fmaf(pr, pr, __builtin_fmaf(pi, pi, kTinyPower)));
      const float aq = __builtin_amdgcn_sqrtf(__builtin_fmaf(qr, qr, __builtin_fmaf(qi, qi, kTinyPower)));
      const float w = exact_step(fast_angle(qr, qi, aq), fast_angle(pr, pi, ap), pr, pi, qr, qi);
      const double d = n < N - 1 ? (double)w - Kw : 0.0, d2 = d * d;
      c0 += d; c1 += d2; c2 += d2 * d; c3 += d2 * d2;
    }
  }
  frequency_features(Kw, wave_sum_f64(c0), wave_sum_f64(c1), wave_sum_f64(c2), wave_sum_f64(c3), N, f5, f9);
}

// The 15 mixed-moment sums of ONE frame in fp64, by the whole wave: the slow path behind the finaliser's cancellation flag
// (cancellation_suspect, amcx_math.h) -- ~0.5 % of frames.  Same shape as wave_exact_frequency: the frame re-read from
// memory (L2 / Infinity Cache), lane l takes samples l, l + 64, ..., eight loads in flight per trip; every product and
// sum in fp64, so what is left is the rounding of the complex64 samples themselves -- the reference's complex128
// evaluation of the same samples (features.py:46-58) to ~1e-16 of the summands' scale.  t: sA, sBh, sP, sAA, sX4, sAB,
// sAP, sBP, sAAA, sABB, sAAB, sBBB, sAAP, sX4P, sABP (FrameSums' order), the totals in every lane.  sc: the power of two
// a re-run multiplies the frame by (1 in the throughput pass).
template <int N>
__device__ __forceinline__ void wave_exact_moments(const float2* __restrict__ src, float sc, int lane, double (&t)[15]) {
  // lane l takes the sample pairs (2 l, 2 l + 1) + 128 j, one global_load_dwordx4 each, kG of them per trip of a ROLLED
  // loop, the next trip's requested before this one's are used.  Compact on purpose: this code runs for one frame in
  // ~150, cold every time -- what it costs is its instruction-cache misses and the round trips of the re-read, not its
  // ~26 fp64 instructions per sample.
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(1))) v4f* gv4f;   // (a function argument: without this the loads are flat_load)
  constexpr int kAll = N / 128, kWant = N >= 4096 ? 8 : 4, kG = kAll < kWant ? kAll : kWant, kTrips = kAll / kG;
  static_assert(kAll >= 1 && kAll % kG == 0, "frame sizes are powers of two >= 128");
  gv4f const base = (gv4f)reinterpret_cast<const v4f*>(src + 2 * lane);
  double a[15];
#pragma unroll
  for (int k = 0; k < 15; ++k) a[k] = 0.0;
  v4f nxt[kG];
#pragma unroll
  for (int u = 0; u < kG; ++u) nxt[u] = base[64 * u];
#pragma unroll 1
  for (int tr = 0; tr < kTrips; ++tr) {
    v4f p[kG];
#pragma unroll
    for (int u = 0; u < kG; ++u) p[u] = nxt[u];
    if (tr + 1 < kTrips) {
#pragma unroll
      for (int u = 0; u < kG; ++u) nxt[u] = base[64 * (kG * (tr + 1) + u)];
    }
#pragma unroll
    for (int u = 0; u < 2 * kG; ++u) {
      const double re = (double)(((u & 1) ? p[u >> 1].z : p[u >> 1].x) * sc), im = (double)(((u & 1) ? p[u >> 1].w : p[u >> 1].y) * sc);
      const double im2 = im * im;
      const double A = __builtin_fma(re, re, -im2), P = __builtin_fma(re, re, im2), Bh = re * im;
      const double AA = A * A, BB = Bh * Bh, AP = A * P;
      const double X4 = __builtin_fma(-4.0, BB, AA);
      a[0] += A; a[1] += Bh; a[2] += P; a[3] += AA; a[4] += X4;
      a[5] = __builtin_fma(A, Bh, a[5]);
      a[6] += AP;
      a[7] = __builtin_fma(Bh, P, a[7]);
      a[8] = __builtin_fma(AA, A, a[8]);
      a[9] = __builtin_fma(A, BB, a[9]);
      a[10] = __builtin_fma(AA, Bh, a[10]);
      a[11] = __builtin_fma(BB, Bh, a[11]);
      a[12] = __builtin_fma(AA, P, a[12]);
      a[13] = __builtin_fma(X4, P, a[13]);
      a[14] = __builtin_fma(AP, Bh, a[14]);
      if (u & 1) __builtin_amdgcn_sched_barrier(0);       // (or the scheduler converts every sample of the trip to fp64 first)
    }
  }
#pragma unroll 1
  for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
    for (int k = 0; k < 15; ++k) a[k] += __shfl_xor(a[k], off, 64);
  }
#pragma unroll
  for (int k = 0; k < 15; ++k) t[k] = a[k];
}

// ... and |C20| ... |C63| (features 10-18) of that frame from them, stored by the lane with `mine` set over columns 9-17 of
// the frame's row, which the same lane has stored before (one thread, one address: program order) -- what every throughput
// kernel's finaliser does with a frame it flagged.  h: half the exponent of the power of two the frame was multiplied by
// (finalize_features<SCALED>: feature j of the frame itself is the scaled one times 2^(h * twice_order_j)); 0 where the
// frame was not scaled.
// What this path costs (N = 2048, same box, tools/cancel_cost.py, profiles/r6_cancel_cost.txt): the predicate +0.4 %; the code
// merely being there +0.1 % in this form -- a ROLLED loop, ~3 KB -- but +1.0 % with the sweep unrolled over sixteen
// samples per trip and +1.3 % as a noinline function (no frame was flagged in those runs: the hot loop's instructions were
// the same, their layout and registers were not); a flagged frame itself 0.6 frame times.  The rest of what flagged frames
// cost was IMBALANCE between workgroups (one contiguous slice = one (modulation, SNR) cell: 5 % of noiseless QPSK
// frames are flagged, none of BPSK's), gone with the interleaved runs of wave_body.
template <int N>
__device__ __forceinline__ void wave_exact_cumulants(const float2* __restrict__ src, float sc, int h, int lane, bool mine,
                                                     float* __restrict__ dst_row) {
  double t[15];
  wave_exact_moments<N>(src, sc, lane, t);
  if (mine) {
    moment_features(t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8], t[9], t[10], t[11], t[12], t[13], t[14], (double)N,
                    [&](int j, int twice_order, double v) { dst_row[j] = (float)__builtin_ldexp(v, h * twice_order); });
  }
}

// ---------------------------------------------------------------------------
// The throughput kernel.  Frames OUTSIDE the fp32 sums' range (mean power outside [1e-10, 1e10], or a sum that is not
// finite: is_outside_fp32_range) are found by the finaliser and re-run by the same wave right behind its batch, in this
// launch: each multiplied by an exact power of two first -- 2^-ex, ex the even-rounded exponent of its largest
// component, so that every component is below 4 and no sixth-order product can overflow -- and un-scaled in the fp64
// finaliser through the features' scaling laws (finalize_features<true>).  A row is stored once, final.  (Rounds 2-3
// marked such frames in band, f5 = -inf, and re-ran them in a second launch -- of this machine at N = 1024, 2048, 4096,
// amcx_range_wave_kernel, of the block kernel's fp64 routine at the short sizes: 8 us per step when there was nothing
// to do, and a consumer that did not order itself behind the call could see marked rows.  Round 4 moved the re-run
// here: one launch per step; same box, through the library: +1.3 % at N = 2048, +0.6 % at 1024, +-0 at 4096, +3.0 % /
// +0.5 % / +1.8 % at 128 / 256 / 512 -- profiles/r4_redo_in_kernel_ab.txt.  A launch boundary costs more than the 8 us of
// the second kernel: 4 096 persistent waves drain and ramp up twice per step.)
template <int N>
__device__ __forceinline__ void wave_body(
    const float2* __restrict__ iq, long long n_frames, long long row_stride,
    float* __restrict__ out, long long out_stride AMCX_STAMP_ARG) {
  using C = Cfg<N>;
  constexpr int R = C::kFftRows, ROWS = C::kHeldRows;
  constexpr int kWavesPerWG = C::kWavesPerWG, kThreads = C::kThreads, kTailFrames = C::kTailFrames;
  constexpr int kFramesPerWave = C::kFramesPerWave, kTailChunk = C::kTailChunk;
  extern __shared__ float4 amcx_wave_smem[];
  char* smem = reinterpret_cast<char*>(amcx_wave_smem);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* t2 = smem;                                        // pass-2 twiddles [k1 slot][15] complex
  char* t3 = smem + C::kT2Bytes;                          // pass-3 twiddles [phase][j][7][lane] complex
  char* t4 = smem + C::kTableBytes - C::kT4Bytes;         // [64][2] complex (N = 4096 only)
  char* ex = smem + C::kTableBytes + wave * kExchangeBytes;
  float* stash = reinterpret_cast<float*>(smem + C::kTableBytes + kWavesPerWG * kExchangeBytes) +
                 wave * (kFramesPerWave * C::kFlushes * kStashStride);

  // ---- work distribution ------------------------------------------------------
  // The frames are cut into runs of kRunFrames; workgroup w owns runs w, w + G, w + 2 G, ... (G workgroups) and its
  // waves take chunks of kFramesPerWave of them from an LDS counter instead of a fixed share.  With a fixed share the oldest wave
  // of each SIMD (VALU issue is arbitrated by age) finished in 52 % of the kernel's
  // time and the youngest set its length, the SIMD idling with one wave left
  // (tools/wave_stamps.hip: lifetimes 1.22 / 1.77 / 2.36 ms min / mean / max).  The
  // last kTailFrames frames of a workgroup go out in chunks of kTailChunk to level the finish.
  // INTERLEAVED chunks, not one contiguous slice per workgroup (rounds 1-5): what a frame costs depends on its data where
  // a slow path is taken -- the exact f5 / f9 of +-pi ties, the fp64 moment sums of cancelling cumulants: 5 % of
  // noiseless QPSK frames, none of BPSK's -- and a container holds its frames sorted by modulation and SNR, so a
  // contiguous slice is one (modulation, SNR) cell and the launch lasted as long as its unluckiest cell.
  // `v` below is an index into the workgroup's own frames in the order it takes them: frame = ((v / Q) G + w) Q + v % Q.
  unsigned* const counters = reinterpret_cast<unsigned*>(smem + C::kCounterOffset);
  if (tid == 0) { counters[0] = 0; counters[1] = 0; }
  constexpr int kQ = C::kRunFrames;                        // frames per interleaved run
  static_assert(kQ % kFramesPerWave == 0 && kFramesPerWave % kTailChunk == 0, "grabs stay inside a run");
  const long long n_wg = gridDim.x, wg = blockIdx.x;
  const long long full_runs = n_frames / kQ;
  const int rem_frames = (int)(n_frames - full_runs * kQ);                     // the last, short run: index full_runs
  const long long my_full = full_runs > wg ? (full_runs - wg - 1) / n_wg + 1 : 0;
  const long long slice_len = my_full * kQ + (rem_frames > 0 && full_runs % n_wg == wg ? rem_frames : 0);
  // whole chunks in the body, so that a tail grab never straddles two runs
  const long long body_len = slice_len > kTailFrames ? (slice_len - kTailFrames) / kFramesPerWave * kFramesPerWave : 0;
  const long long tail_len = slice_len - body_len;
  auto frame_of = [&](long long v) -> long long {
    const long long k = v / kQ;
    return (k * n_wg + wg) * kQ + (v - k * kQ);
  };

  // ---- twiddle tables, once per workgroup -----------------------------------
  build_fft_tables<C::kFftN>(t2, t3, tid, kThreads);
  if constexpr (C::kSplit) {
    for (int e = tid; e < 128; e += kThreads) {            // T4[l][b] = W_4096^(2l+b)
      float sn, cs;
      sincospif((float)e * (1.0f / 2048.0f), &sn, &cs);
      reinterpret_cast<float2*>(t4)[e] = make_float2(cs, -sn);
    }
  }
  __syncthreads();

  // lane-constant LDS byte addresses
  const int kkL = lane >> 3, n3L = lane & 7;
  LaneAddr la;
  la.tw2 = t2 + kkL * kTw2Stride;
  la.tw3 = t3 + lane * 8;
  la.ex1_w = ex + lane * 8;
  la.ex1_r = ex + (kkL * kEx1StrideKK + (n3L & 1) * kEx1StrideB + (n3L >> 1)) * 8;
  la.ex2_w = ex + lane * 8;
  la.ex2_r = ex + (n3L * kEx2StrideK2 + kkL * 8) * 8;   // (lane&7) is k2lo for the reader
  float* const a_lds = reinterpret_cast<float*>(ex) + lane;   // |x| parked in the wave's LDS: [e][lane]

#ifdef AMCX_WAVE_STAMPS
  unsigned long long stamp_acc[kStampSections] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
  const unsigned long long real0 = __builtin_amdgcn_s_memrealtime();
#endif
  for (;;) {
    // grab the next chunk of this workgroup's slice (lane 0 asks, the wave follows)
    long long f0;
    int n_here;
    {
      unsigned got = 0;
      if (lane == 0) got = __hip_atomic_fetch_add(&counters[0], (unsigned)kFramesPerWave, __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_WORKGROUP);
      got = __builtin_amdgcn_readfirstlane(got);
      if ((long long)got < body_len) {
        f0 = frame_of(got);
        const long long left = body_len - got;
        n_here = left < kFramesPerWave ? (int)left : kFramesPerWave;
      } else {
        unsigned t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(&counters[1], (unsigned)kTailChunk, __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_WORKGROUP);
        t = __builtin_amdgcn_readfirstlane(t);
        if ((long long)t >= tail_len) break;
        f0 = frame_of(body_len + t);
        const long long left = tail_len - t;
        n_here = left < kTailChunk ? (int)left : kTailChunk;
      }
    }

    // lane l gets samples 128 i + 2 l + {0,1} of each row: kRows x global_load_dwordx4; every
    // byte is read once -> non-temporal
    auto load_frame = [&](float (&xr)[2 * ROWS], float (&xi)[2 * ROWS], long long f) {
      const float2* src = iq + f * row_stride + 2 * lane;
      static_for<ROWS>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f* p = reinterpret_cast<const v4f*>(src + 128 * i);
        const v4f v = __builtin_nontemporal_load(p);
        xr[2 * i] = v.x; xi[2 * i] = v.y; xr[2 * i + 1] = v.z; xi[2 * i + 1] = v.w;
      });
    };
    // one frame, registers -> stash row g
    auto frame = [&](const float (&xr)[2 * ROWS], const float (&xi)[2 * ROWS], int g) {
      asm volatile("; MARK load");
      __builtin_amdgcn_s_setprio(AMCX_PRIO_OF(0));
      AMCX_STAMP(7);
      // wave reduction of a frame's 27 per-lane sums into one stash row
      auto reduce_sums = [&](float (&r28)[28], float (&r7)[7]) __attribute__((always_inline)) {
        // two swap levels (lane bits 5, 4) halve the live values each time --
        // v_permlane32_swap / v_permlane16_swap exchange half a register pair in one
        // instruction -- then four DPP steps inside the 16-lane rows.  70 VALU ops
        // against 162 for 27 independent 6-step butterflies.
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
        static_for<7>([&](auto jj) {
          constexpr int j = decltype(jj)::value;
          float v = r14[2 * j] + r14[2 * j + 1];
          v += dpp<kQuadXor1>(v);
          v += dpp<kQuadXor2>(v);
          v += dpp<kRowHalfMirror>(v);
          v += dpp<kRowMirror>(v);
          r7[j] = v;
        });
      };
      auto store_sums = [&](const float (&r7)[7], float* row) __attribute__((always_inline)) {
        // (with TWO copies of this frame code in the kernel -- the batch loop's and rerun_scaled's -- the compiler hoists the lane-derived offset
        //  out of both, spills it in the prologue and reloads it here once per frame behind a wait: the lane index
        //  goes through an empty asm instead, so that the offset is formed where it is used)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        if ((ln & 15) == 0) {                      // one lane per row: rows hold 4j + {0, 2, 1, 3}
          const int rsel = ln >> 4;
          float* dst = row + (((rsel & 1) << 1) | (rsel >> 1));
          static_for<7>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            dst[4 * j] = r7[j];
          });
        }
      };

      // =====================================================================
      // statistics sweep
      // =====================================================================
      Stats S;
      float sa_flushed = 0.f;                    // this lane's sum of |x| over the stretches flushed so far
      {
        static_for<ROWS>([&](auto ii) {
          constexpr int i = decltype(ii)::value;
          float a0, a1;
          S.template row<i == 0, i == ROWS - 1>(xr[2 * i], xi[2 * i], xr[2 * i + 1], xi[2 * i + 1], lane, a0, a1);
          if constexpr (!C::kSplit) {
            a_lds[(2 * i) * 64] = a0;
            a_lds[(2 * i + 1) * 64] = a1;
          }
          if constexpr (C::kFlushes == 2 && i == ROWS / 2 - 1) {
            __builtin_amdgcn_sched_barrier(0);
            sa_flushed += S.sa;
            float r28[28] = {S.sA, S.sBh, S.sP, S.sAA, S.sX4, S.sAB, S.sAP, S.sBP, S.sAAA, S.sABB,
                             S.sAAB, S.sBBB, S.sAAP, S.sX4P, S.sABP, S.sa, 0.f, 0.f, 0.f,
                             S.st1, S.st2, S.sab1, S.sab2, S.sw1, S.sw2, S.sw3, S.sw4, 0.f};
            float q7[7];
            reduce_sums(r28, q7);
            store_sums(q7, stash + (g * C::kFlushes) * kStashStride);
            S.clear_sums();
            __builtin_amdgcn_sched_barrier(0);
          }
        });
      }
      asm volatile("; MARK envelope");
      __builtin_amdgcn_s_setprio(AMCX_PRIO_OF(1));
      AMCX_STAMP(0);
      __builtin_amdgcn_sched_barrier(0);
      // envelope second sweep about the exact mean
      const float mu = bcast_l63(wave_sum_l63(S.sa + sa_flushed)) * (1.0f / (float)N);
      {
        static_for<2 * ROWS>([&](auto ee) {
          constexpr int e = decltype(ee)::value;
          if constexpr (C::kSplit) {
            S.envelope(__builtin_amdgcn_sqrtf(__builtin_fmaf(xr[e], xr[e], __builtin_fmaf(xi[e], xi[e], kTinyPower))), mu);
          } else {
            S.envelope(a_lds[e * 64], mu);
          }
        });
      }

      // =====================================================================
      // wave reduction of the 27 sums -> stash row g, before the FFT so that the sums'
      // registers are free while it runs
      // =====================================================================
      asm volatile("; MARK reduce");
      __builtin_amdgcn_s_setprio(AMCX_PRIO_OF(2));
      AMCX_STAMP(3);
      __builtin_amdgcn_sched_barrier(0);
      float* const row = stash + (g * C::kFlushes + (C::kFlushes - 1)) * kStashStride;   // the frame's last row
      float r7[7];
      {
        float r28[28] = {S.sA, S.sBh, S.sP, S.sAA, S.sX4, S.sAB, S.sAP, S.sBP, S.sAAA, S.sABB,
                         S.sAAB, S.sBBB, S.sAAP, S.sX4P, S.sABP, S.sa, S.sad1, S.sad2, S.sad4,
                         S.st1, S.st2, S.sab1, S.sab2, S.sw1, S.sw2, S.sw3, S.sw4, 0.f};
        reduce_sums(r28, r7);
      }
      const unsigned long long tie = __builtin_amdgcn_ballot_w64(S.wmax > kPi - kTieBand);
      store_sums(r7, row);                       // (after the ballot: the other order costs 30 VGPRs)
      if (lane == 63) {
        row[kNumSums + 1] = S.Kt;
        row[kNumSums + 2] = S.Kw;
        row[kNumSums + 3] = S.Ka;
        row[kNumSums + 4] = tie != 0 ? 1.0f : 0.0f;
      }

      // =====================================================================
      // spectral peak
      // =====================================================================
      asm volatile("; MARK fft1");
      AMCX_STAMP(1);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(AMCX_PRIO_OF(3));
      float peak;
      if constexpr (!C::kSplit) {
        peak = fft_peak<R>(xr, xi, la);
      } else {
        // 4096 points in registers: radix-2 DIF split, s = y[n] + y[n+2048],
        // d = (y[n] - y[n+2048]) * W_4096^n,  n = 128 i + 2 l + b,  W_4096^n = W_32^i * W_4096^(2l+b)
        // row_of(ic) yields row i of the 4096-point sequence as (re, im) of b = 0 and b = 1; each
        // row is asked for exactly once, so a provider may load and accumulate on the way
        auto fft4096 = [&](auto&& row_of) -> float {
          const float4 w4 = *reinterpret_cast<const float4*>(t4 + lane * 16);
          float sr[2 * R], si[2 * R], dr[2 * R], di[2 * R];
          if constexpr (AMCX_PRIO_OF(6) != AMCX_PRIO_OF(3)) __builtin_amdgcn_s_setprio(AMCX_PRIO_OF(6));
          static_for<R>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            constexpr int lo = 2 * i;
            const float4 p = row_of(std::integral_constant<int, i>{});
            const float4 q = row_of(std::integral_constant<int, R + i>{});
            sr[lo] = p.x + q.x;     si[lo] = p.y + q.y;
            sr[lo + 1] = p.z + q.z; si[lo + 1] = p.w + q.w;
            float d0r = p.x - q.x, d0i = p.y - q.y;
            float d1r = p.z - q.z, d1i = p.w - q.w;
            mul_w32<i>(d0r, d0i);
            mul_w32<i>(d1r, d1i);
            dr[lo] = __builtin_fmaf(d0r, w4.x, -(d0i * w4.y));
            di[lo] = __builtin_fmaf(d0r, w4.y, d0i * w4.x);
            dr[lo + 1] = __builtin_fmaf(d1r, w4.z, -(d1i * w4.w));
            di[lo + 1] = __builtin_fmaf(d1r, w4.w, d1i * w4.z);
          });
          if constexpr (AMCX_PRIO_OF(6) != AMCX_PRIO_OF(3)) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(AMCX_PRIO_OF(3));
          }
          // Two independent transforms, one wave (and only one more on the SIMD): the difference branch's pass 1 --
          // registers only -- is issued in the shadows of the sum branch's LDS round trips (fft_tail's hook), in three
          // pieces: the first stage behind the reads of exchange 1 of phase 0, the 8-point transforms of parity 0
          // behind those of exchange 2, those of parity 1 behind exchange 1 of phase 1.
          float s0r[R], s0i[R], s1r[R], s1i[R], d0r[R], d0i[R], d1r[R], d1i[R];
          fft_head<R>(sr, si, s0r, s0i, s1r, s1i);
          const float pk4 = fft_tail<R>(s0r, s0i, s1r, s1i, la, [&](auto gg, auto ww) {
            constexpr int g = decltype(gg)::value, w = decltype(ww)::value;
            if constexpr (g == 0 && w == 0) fft_head<R>(dr, di, d0r, d0i, d1r, d1i);
            if constexpr (g == 0 && w == 1) { dif<8, 0>(d0r, d0i); dif<8, 0>(d1r, d1i); }
            if constexpr (g == 1 && w == 0) { dif<8, 8>(d0r, d0i); dif<8, 8>(d1r, d1i); }
          });
          __builtin_amdgcn_sched_barrier(0);
          return __builtin_fmaxf(pk4, fft_tail<R, true>(d0r, d0i, d1r, d1i, la));
        };
        peak = fft4096([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          return make_float4(xr[2 * i], xi[2 * i], xr[2 * i + 1], xi[2 * i + 1]);
        });
      }

      const float pk = wave_max_l63(peak);
      if (lane == 63) row[kNumSums] = pk;        // overwrites the zero pad slot 27 (LDS ops of a wave are in order)
    };

    // ---- batch finalisation: lane g turns the sums in stash row g into 18 features ----
    // range_tag true: the rows are those of frames re-run on a pre-scaled copy (rerun_scaled below).
    // Returns the lanes whose frame is outside the fp32 sums' range and has NOT been stored.
    auto finalise = [&](auto range_tag, int count) -> unsigned long long {
      constexpr bool RG = decltype(range_tag)::value;
      bool redo = false;
      lds_wave_fence();
      float feat[18];
      long long f = 0;
      [[maybe_unused]] float sc = 1.0f;
      [[maybe_unused]] int ex_half = 0;
      float kw_shift = 0.f;
      bool tie = false, cancel = false;
      if (lane < count) {
        const float* row = stash + (lane * C::kFlushes + (C::kFlushes - 1)) * kStashStride;   // the frame's last row
        auto sm = [&](int k) -> double {           // sum k of the frame: its stash rows added in fp64
          double t = row[k];
          if constexpr (C::kFlushes > 1) {
#pragma unroll
            for (int h = 1; h < C::kFlushes; ++h) t += (double)row[k - h * kStashStride];
          }
          return t;
        };
        {                                          // fp32, on the stash values, ahead of the fp64 algebra (amcx_math.h)
          float s15[15];
#pragma unroll
          for (int k = 0; k < 15; ++k) {
            s15[k] = row[k];
            if constexpr (C::kFlushes > 1) {
#pragma unroll
              for (int h = 1; h < C::kFlushes; ++h) s15[k] += row[k - h * kStashStride];
            }
          }
          cancel = cancellation_suspect(s15, (float)N, (float)cancel_kappa(N));
        }
        FrameSums F;
        F.sA = sm(0); F.sBh = sm(1); F.sP = sm(2); F.sAA = sm(3); F.sX4 = sm(4); F.sAB = sm(5);
        F.sAP = sm(6); F.sBP = sm(7); F.sAAA = sm(8); F.sABB = sm(9); F.sAAB = sm(10);
        F.sBBB = sm(11); F.sAAP = sm(12); F.sX4P = sm(13); F.sABP = sm(14);
        F.sa = sm(15); F.sad1 = sm(16); F.sad2 = sm(17); F.sad4 = sm(18);
        F.std1 = sm(19); F.std2 = sm(20); F.sab1 = sm(21); F.sab2 = sm(22);
        F.swd1 = sm(23); F.swd2 = sm(24); F.swd3 = sm(25); F.swd4 = sm(26);
        F.gmax_raw = row[27]; F.Kt = row[28]; F.Kw = row[29]; F.Ka = row[30];
        F.pi_tie = row[31] != 0.0f;
        kw_shift = row[29];
        if constexpr (RG) {
          const int code = (int)row[kNumSums + 5];              // (ex + 128) * 64 + index within the block
          const int ex = (code >> 6) - 128;
          cancel = finalize_features<true>(F, N, feat, ex) && cancel;
          sc = __builtin_bit_cast(float, (127 - ex) << 23);     // the 2^-ex the frame was multiplied by
          ex_half = ex / 2;
          f = f0 + (code & 63);
        } else {
          cancel = finalize_features(F, N, feat) && cancel;
          if (is_outside_fp32_range(F, N)) redo = true;         // re-run below, in this kernel; the row is not stored
          f = f0 + lane;
        }
        cancel = cancel && !redo;
        // flagged by the sweep (f5 came back negated) and neither NaN nor on its way to a re-run
        tie = __builtin_signbitf(feat[4]) && feat[4] == feat[4] && feat[4] != -__builtin_inff() && !redo;
      }
      // frames with a phase step within an fp32 rounding of +-pi: f5 and f9 again, the wave on one frame at a time
      unsigned long long ties = __builtin_amdgcn_ballot_w64(tie);
      while (ties != 0) {
        const int idx = __builtin_ctzll(ties);
        ties &= ties - 1;
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(f & 0xffffffffLL), idx);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)f >> 32), idx);
        const long long ft = (long long)(((unsigned long long)hi << 32) | lo);
        float sct = 1.0f;
        if constexpr (RG) sct = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sc), idx));
        const float kwt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, kw_shift), idx));
        float f5x, f9x;
        wave_exact_frequency<N>(iq + ft * row_stride, sct, kwt, lane, f5x, f9x);
        if (lane == idx) { feat[4] = f5x; feat[8] = f9x; }
      }
      if (lane < count && !redo) {
        float* dst = out + f * out_stride;
#pragma unroll
        for (int j = 0; j < 18; ++j) dst[j] = feat[j];
      }
      // frames one of whose cumulants cancels below what fp32 sums resolve (cancellation_suspect): ids 10-18 again from
      // fp64 sums, the wave on one frame at a time, over the row stored above
      unsigned long long cz = __builtin_amdgcn_ballot_w64(cancel);
      while (cz != 0) {
        const int idx = __builtin_ctzll(cz);
        cz &= cz - 1;
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(f & 0xffffffffLL), idx);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)f >> 32), idx);
        const long long ft = (long long)(((unsigned long long)hi << 32) | lo);
        float sct = 1.0f;
        int hx = 0;
        if constexpr (RG) {
          sct = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sc), idx));
          hx = __builtin_amdgcn_readlane(ex_half, idx);
        }
        wave_exact_cumulants<N>(iq + ft * row_stride, sct, hx, lane, lane == idx, out + ft * out_stride);
      }
      lds_wave_fence();
      return __builtin_amdgcn_ballot_w64(redo);
    };

    // frames `todo` (bit i: frame f0 + i) again, each multiplied by 2^-ex first (see the comment above wave_body)
    [[maybe_unused]] const LaneAddr& la_redo = la;
    auto rerun_scaled = [&](unsigned long long todo) {
      while (todo != 0) {
        int cnt = 0;
        for (; cnt < kFramesPerWave && todo != 0; ++cnt) {
          const int idx = __builtin_ctzll(todo);
          todo &= todo - 1;
          float xr[2 * ROWS], xi[2 * ROWS];
          load_frame(xr, xi, f0 + idx);
          float m = 0.f;                                       // largest |component|: NaNs drop out of the maximum
          static_for<2 * ROWS>([&](auto ee) {
            constexpr int e = decltype(ee)::value;
            m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(xr[e])), __builtin_fabsf(xi[e]));
          });
          m = bcast_l63(wave_max_l63(m));
          int ex = 0;                                          // an infinite component keeps 0: the sums go NaN
          if (m >= 0x1p-125f && m <= 3.4028235e38f) ex = (((__builtin_bit_cast(int, m) >> 23) & 0xff) - 127) & ~1;
          const float sc = __builtin_bit_cast(float, (127 - ex) << 23);      // 2^-ex, exact
          static_for<2 * ROWS>([&](auto ee) {
            constexpr int e = decltype(ee)::value;
            xr[e] *= sc; xi[e] *= sc;
          });
          frame(xr, xi, cnt);
          if (lane == 63)
            stash[(cnt * C::kFlushes + (C::kFlushes - 1)) * kStashStride + kNumSums + 5] = (float)((ex + 128) * 64 + idx);
        }
        finalise(std::true_type{}, cnt);
      }
    };
    for (int g = 0; g < n_here; ++g) {
      float xr[2 * ROWS], xi[2 * ROWS];
      load_frame(xr, xi, f0 + g);
      frame(xr, xi, g);
    }

    AMCX_STAMP(4);
    asm volatile("; MARK finalize");
    __builtin_amdgcn_s_setprio(AMCX_PRIO_OF(5));
    const unsigned long long left_over = finalise(std::false_type{}, n_here);
    asm volatile("; MARK redo");
    if (left_over != 0) rerun_scaled(left_over);              // frames outside the fp32 sums' range: never on ordinary data
    AMCX_STAMP(5);
  }
#ifdef AMCX_WAVE_STAMPS
  if (lane == 0) {
    const long long w = (long long)blockIdx.x * kWavesPerWG + wave;
    stamp_acc[6] = __builtin_amdgcn_s_memrealtime() - real0;   // wave lifetime, 100 MHz ticks
#if AMCX_WAVE_STAMPS != 1
    stamp_acc[0] = __builtin_amdgcn_s_memtime() - stamp_last;  // the same span in shader cycles
#endif
    for (int k = 0; k < kStampSections; ++k) stamp_out[w * kStampSections + k] = stamp_acc[k];
  }
#endif
}

template <int N>
__global__ __launch_bounds__(Cfg<N>::kThreads, (Cfg<N>::kWavesPerWG + 3) / 4) void amcx_features18_wave_kernel(
    const float2* __restrict__ iq, long long n_frames, long long row_stride,
    float* __restrict__ out, long long out_stride AMCX_STAMP_ARG) {
#ifdef AMCX_WAVE_STAMPS
  wave_body<N>(iq, n_frames, row_stride, out, out_stride, stamp_out);
#else
  wave_body<N>(iq, n_frames, row_stride, out, out_stride);
#endif
}

}  // namespace wave

inline bool wave_supports(int frame_size) {
  return frame_size >= 128 && frame_size <= 32768 && (frame_size & (frame_size - 1)) == 0;
}

inline const char* wave_kernel_name(int frame_size) {
  switch (frame_size) {
    case 128: return "amcx_features18_short_kernel<128>";      // four frames per wave: amcx_short_kernel.h
    case 256: return "amcx_features18_short_kernel<256>";
    case 512: return "amcx_features18_short_kernel<512>";
    case 1024: return "amcx_features18_wave_kernel<1024>";
    case 2048: return "amcx_features18_wave_kernel<2048>";
    case 4096: return "amcx_features18_wave_kernel<4096>";
    case 8192: return "amcx_features18_quad_kernel";
    case 16384: return "amcx_features18_group_kernel<8>";
    case 32768: return "amcx_features18_group_kernel<16>";
    default: return "";
  }
}

#ifndef AMCX_WAVE_STAMPS
template <int N>
inline hipError_t launch_wave_n(const float2* iq, int64_t n_frames, int64_t row_stride, float* out,
                                int64_t out_stride, hipStream_t stream, int cus) {
  auto kern = wave::amcx_features18_wave_kernel<N>;
  constexpr int lds = wave::Cfg<N>::kLdsBytes;
  // > 64 KiB of dynamic LDS needs the attribute; it is per device, so once per (kernel, device)
  static bool lds_attr_set[64] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64 || !lds_attr_set[dev]) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) lds_attr_set[dev] = true;   // benign race: idempotent
  }
  int64_t grid = (int64_t)cus;                        // persistent: one resident workgroup per CU
  const int64_t min_slice = wave::Cfg<N>::kWavesPerWG;   // at least a frame per wave
  if (grid * min_slice > n_frames) grid = (n_frames + min_slice - 1) / min_slice;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(wave::Cfg<N>::kThreads), lds, stream, iq,
                     (long long)n_frames, (long long)row_stride, out, (long long)out_stride);
  return hipGetLastError();
}

inline hipError_t launch_wave(const float2* iq, int64_t n_frames, int32_t frame_size,
                              int64_t row_stride, float* out, int64_t out_stride,
                              hipStream_t stream, int cus) {
  switch (frame_size) {
    // (128, 256 and 512 ran here until late in round 5 -- 8 / 4 / 2 frames sharing one run of FFT passes 2-3 -- and have a
    //  kernel of their own now, amcx_short_kernel.h: +34 % / +13 % / +2.5 ... 5 %)
    case 1024: return launch_wave_n<1024>(iq, n_frames, row_stride, out, out_stride, stream, cus);
    case 2048: return launch_wave_n<2048>(iq, n_frames, row_stride, out, out_stride, stream, cus);
    case 4096: return launch_wave_n<4096>(iq, n_frames, row_stride, out, out_stride, stream, cus);
    default: return hipErrorNotSupported;       // 128, 256, 512, 8192, 16384, 32768 have kernels of their own (amcx.hip)
  }
}

#endif

}  // namespace amcx
