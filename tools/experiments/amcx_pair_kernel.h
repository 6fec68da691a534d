// EXPERIMENT (-DAMCX_EXP_PAIR4096; not in the product library): N = 4096 (BASELINE configs[2]) with TWO wavefronts per
// frame, each holding one half of it in registers.
//
// RESULT under wave priority (later in round 4, the product's sweeps at s_setprio 1): 85.8 M frames/s with AMCX_PAIR_PRIO_MASK = 3
// against 83.5 M without and 92.5 M for the one-wave kernel -- 7.3 % SLOWER (profiles/r4_pair_vs_wave4096_ab.txt, second part).
// RESULT before it (round 4, same box, alternating runs through the library, tools/ab_lib.sh "-DAMCX_EXP_PAIR4096" 2 4096;
// profiles/r4_pair_vs_wave4096_ab.txt): 85.2-85.8 M frames/s against 86.7-87.2 M for the one-wave kernel -- 1.7 % SLOWER,
// where the bar for adopting it was +3 %.  Parity is green (the N = 4096 golden, ragged, variant and full-size tests
// pass through it).  Why it does not pay: what four waves per SIMD buy over two on this arithmetic is ~8 % (measured on
// the N = 2048 kernel), and the pair pays for it with what the one-wave kernel does not have -- 2 x 16 KiB of LDS
// exchange per frame, the envelope's 32 square roots taken twice per wave (the exchange area is where |x| would be
// parked), each wave running the fp64 finaliser for its own two frames of a chunk, five meetings per frame, a few
// per cent more instructions for the (a - b) W branch's second copy of the code -- and with the board's power cap
// (the one-wave kernel runs at 2.09 GHz, not 2.4: DESIGN.md section 4.3) handing part of every cycle saved back as
// clock.  Kept, fenced, because the machinery it proves is reusable: a pair / quad that synchronises through LDS
// epoch words instead of s_barrier, and the three compiler traps at 128 VGPRs listed below.
//
// Rounds 1-3 gave a 32 KiB frame to ONE wave (amcx_features18_wave_kernel<4096>, the product): the whole frame in
// 128 VGPR pairs, so 256 VGPRs per wave and only 2 waves per SIMD -- 3.4 SIMD cycles per VALU instruction against 2.85
// at the four waves per SIMD of the N = 2048 kernel, 0.344 of the HBM roofline against 0.39
// (profiles/r3_n4096_summary.json).  Here a frame belongs to a PAIR of waves at the N = 2048 kernel's register budget
// (128 VGPRs, 16 waves per CU):
//
//   * wave h of the pair loads half h of the frame (samples [2048 h, 2048 h + 2048), the N = 2048 register layout:
//     16 x global_load_dwordx4, every byte read from HBM once) and runs the N = 2048 statistics sweep on it with its
//     own shifts -- the finaliser re-centres the two halves' shifted sums in fp64 -- plus, for half 0, the one phase
//     step that crosses into half 1 (reference: np.diff(np.unwrap(np.angle(x))) over the whole frame,
//     features.py:27-31);
//   * radix-2 decimation in frequency across the pair:
//       X[2k] = FFT_2048(a + b),   X[2k+1] = FFT_2048((a - b) W_4096^n),   a / b = first / second half
//     the halves cross through LDS in two rounds of eight rows (one ds_write_b128 and one ds_read_b128 per row and
//     wave, linear, conflict-free) INSIDE each wave's own FFT exchange buffer, so the LDS budget is the N = 2048
//     kernel's; the wave with half 0 forms a + b in place, the other (a - b) W (row factor W_32^i an immediate, lane
//     factor W_4096^(2l+b) four registers), and each runs ONE 2048-point register FFT (fft_peak<16>); the frame's
//     spectral peak is the larger of the two.  Which wave takes which half alternates from frame to frame, so that
//     the (a - b) W branch's extra multiplications even out over the SIMDs;
//   * the mean envelope crosses the pair through two floats of LDS (the second envelope sweep is about the exact
//     mean, features.py:82-85);
//   * the pair synchronises WITHOUT s_barrier: sixteen waves (eight pairs) share one workgroup -- one copy of the
//     16 KB twiddle tables per CU is what lets 16 waves fit its LDS -- and a workgroup barrier would march all
//     eight pairs in step (the first quad kernel did that: waves parked 32 % of their cycles,
//     amcx_quad_kernel.h).  Each wave owns an epoch word in LDS: `arrive` is a release store of the next epoch,
//     `wait` an acquire spin (s_sleep between polls) on the partner's word; LDS operations of one wave complete in
//     order, so the partner's rows are there when its epoch is.  Five meetings per frame (rows 0-7 published / read,
//     rows 8-15 published / read, chunk hand-over folded into the first) and one per batch.  Both waves of a pair
//     belong to one workgroup, hence are resident together: a wait always ends.  As insurance against a hang of
//     the whole board should that reasoning ever be broken by a later edit, the spin is BOUNDED: after ~2^22 polls
//     the wave gives up, stores NaN features for its chunk and leaves, and so does its partner;
//   * pairs take chunks of four frames from the workgroup's LDS counter (two-frame chunks over the tail of its
//     slice), the even wave asking for both; after a chunk's last FFT each wave of the pair finalises every other
//     frame of it in fp64 (one frame per lane; a frame with a phase step within an fp32 ulp of +-pi gets its f5 / f9
//     from wave_exact_frequency, as everywhere).
//
// Three things the compiler does to this kernel at 128 VGPRs, each of which cost 50-150 spilled registers until it was
// stopped (they apply to any kernel that recomputes values behind a synchronisation point):
//   1. it recognises the envelope sweep's sqrt(re^2 + im^2) as the statistics sweep's and keeps all 32 alive across
//      the meeting instead of recomputing them -> the samples pass through an empty asm behind the meeting;
//   2. nothing of the statistics sweep is NEEDED before the meeting, so most of it sinks behind it, into the envelope
//      sweep and the reduction -> pin_sums(): every running sum is an opaque value at the end of the sweep;
//   3. a dozen lane-dependent LDS addresses are loop invariants: hoisted out of the frame loop, spilled in the sweep,
//      each reloaded once per frame behind an exposed s_waitcnt vmcnt(0) -> lane_here(): addresses are formed where
//      they are used from a copy of the lane index the compiler cannot see through.
// With them: no scratch instruction between a frame's load and its FFT peak; 50 spilled dwords in the prologue and
// the per-chunk fp64 finaliser (tools/resource_usage.py on an experiment build).
//
// Frames outside the fp32 sums' range are flagged (f5 = -inf) and redone by amcx_range_fixup_kernel (the block
// kernel's fp64-sum routine) in a second launch; the one-wave kernel re-runs them itself (amcx_wave_kernel.h).
// LDS per workgroup: 16 256 B tables + 16 x (8 672 B region + 528 B stash) + 168 B of sync words = 163 624 B.
// Algorithmic HBM bytes per frame: 8 * 4096 read + 72 written.
#pragma once

#include "amcx_quad_kernel.h"

namespace amcx {
namespace pair {

using namespace wave;
using quad::Recentred;
using quad::reduce_store;

constexpr int kN = 4096, kHalf = 2048, kRowsH = 16;
constexpr int kWavesPerWG = 16, kPairs = kWavesPerWG / 2, kThreads = 64 * kWavesPerWG;
constexpr int kBatch = 4;                                    // frames a pair takes and finalises together
constexpr int kTailChunk = 2;                                // ... over the last stretch of a workgroup's slice
constexpr int kTailFrames = kPairs * kBatch;
using C2 = Cfg<2048>;                                        // the register FFT every wave runs
constexpr int kTabBytes = C2::kT2Bytes + C2::kT3Bytes;
constexpr int kRoundRows = kRowsH / 2;                       // rows of a half exchanged per round
constexpr int kRegionBytes = kExchangeBytes;                 // a wave's region: 8 rows of 1 KiB in a round, its FFT exchange buffer afterwards
constexpr int kStashRow = kStashStride;                      // floats per (frame, half)
constexpr int kStashFloats = kBatch * 2 * kStashRow;         // one pair's stash
constexpr int kOffRegions = kTabBytes;
constexpr int kOffStash = kOffRegions + kWavesPerWG * kRegionBytes;
constexpr int kOffEpoch = kOffStash + kPairs * kStashFloats * 4;   // [wave] epoch word
constexpr int kOffSa = kOffEpoch + kWavesPerWG * 4;          // [wave] this frame's sum of |x| over the wave's half
constexpr int kOffChunk = kOffSa + kWavesPerWG * 4;          // [pair] (offset into the slice) << 3 | frames in the chunk
constexpr int kOffCounters = kOffChunk + kPairs * 4;
constexpr int kLdsBytes = kOffCounters + 8;
static_assert(kLdsBytes <= 163840, "one workgroup per CU must fit in 160 KiB of LDS");
static_assert(kRoundRows * 1024 <= kRegionBytes, "a wave's region holds a round's rows");
constexpr int kSpinLimit = 1 << 22;

// Wave priority as in amcx_wave_kernel.h (AMCX_PRIO_MASK): which sections of a frame run at s_setprio 1 -- bit 0 the statistics
// sweep (phase A), bit 1 envelope + wave reduction + round 1 of the radix-2 stage (phase B), bit 2 round 2 of the radix-2
// stage.  A wave drops to 0 while it waits at a meeting and for its FFT.
#ifndef AMCX_PAIR_PRIO_MASK
#define AMCX_PAIR_PRIO_MASK 3
#endif
#define AMCX_PAIR_PRIO(b) __builtin_amdgcn_s_setprio((AMCX_PAIR_PRIO_MASK >> (b)) & 1)

// W_4096^(2 l + b) = (cos, -sin)(2 pi (2 l + b) / 4096) for lane l, b = 0, 1: the lane factors of the radix-2 stage, correctly
// rounded (generated with numpy in float64).  1 KiB of constant global memory that every wave re-reads from the caches
// behind meeting (1) -- the CU's LDS has 216 bytes left, and four registers held through the statistics sweep were four
// too many at 128.
__device__ const float4 kLaneTwiddle4096[64] = {
    {1.0f, -0.0f, 0.9999988079071045f, -0.0015339801320806146f},
    {0.9999952912330627f, -0.0030679567717015743f, 0.99998939037323f, -0.004601926077157259f},
    {0.999981164932251f, -0.006135884672403336f, 0.9999706149101257f, -0.007669828832149506f},
    {0.9999576210975647f, -0.009203754365444183f, 0.9999423623085022f, -0.01073765940964222f},
    {0.9999247193336487f, -0.012271538376808167f, 0.9999046921730042f, -0.0138053884729743f},
    {0.9998823404312134f, -0.015339205972850323f, 0.9998576641082764f, -0.016872987151145935f},
    {0.9998306035995483f, -0.018406730145215988f, 0.9998011589050293f, -0.01994042843580246f},
    {0.999769389629364f, -0.0214740801602602f, 0.9997352957725525f, -0.023007681593298912f},
    {0.99969881772995f, -0.024541229009628296f, 0.9996600151062012f, -0.026074718683958054f},
    {0.9996188282966614f, -0.027608145028352737f, 0.9995753169059753f, -0.029141508042812347f},
    {0.9995294213294983f, -0.030674804002046585f, 0.999481201171875f, -0.032208025455474854f},
    {0.9994305968284607f, -0.03374117240309715f, 0.9993776679039001f, -0.035274237394332886f},
    {0.9993223547935486f, -0.03680722415447235f, 0.9992647767066956f, -0.03834012150764465f},
    {0.9992047548294067f, -0.039872925728559494f, 0.9991424083709717f, -0.04140564054250717f},
    {0.9990777373313904f, -0.04293825849890709f, 0.9990106821060181f, -0.04447077214717865f},
    {0.9989413022994995f, -0.046003181487321854f, 0.9988695383071899f, -0.0475354827940464f},
    {0.9987954497337341f, -0.049067676067352295f, 0.9987190365791321f, -0.05059975013136864f},
    {0.998640239238739f, -0.05213170498609543f, 0.9985590577125549f, -0.05366353690624237f},
    {0.9984755516052246f, -0.055195245891809464f, 0.998389720916748f, -0.05672682076692581f},
    {0.9983015656471252f, -0.058258265256881714f, 0.9982110261917114f, -0.05978957191109657f},
    {0.9981181025505066f, -0.06132073700428009f, 0.9980228543281555f, -0.06285175681114197f},
    {0.9979252815246582f, -0.0643826276063919f, 0.9978253245353699f, -0.06591334939002991f},
    {0.9977230429649353f, -0.06744392216205597f, 0.9976184368133545f, -0.0689743310213089f},
    {0.9975114464759827f, -0.0705045759677887f, 0.9974021315574646f, -0.07203464955091476f},
    {0.9972904324531555f, -0.0735645666718483f, 0.9971764087677002f, -0.0750942975282669f},
    {0.9970600605010986f, -0.07662386447191238f, 0.996941328048706f, -0.07815324515104294f},
    {0.9968202710151672f, -0.07968243956565857f, 0.9966968894004822f, -0.08121144771575928f},
    {0.9965711236000061f, -0.08274026215076447f, 0.9964430332183838f, -0.08426889032125473f},
    {0.9963126182556152f, -0.08579730987548828f, 0.9961798191070557f, -0.08732553571462631f},
    {0.9960446953773499f, -0.08885355293750763f, 0.9959072470664978f, -0.09038136154413223f},
    {0.9957674145698547f, -0.09190895408391953f, 0.9956252574920654f, -0.0934363380074501f},
    {0.9954807758331299f, -0.09496349841356277f, 0.9953339099884033f, -0.09649042785167694f},
    {0.9951847195625305f, -0.0980171412229538f, 0.9950332045555115f, -0.09954361617565155f},
    {0.9948793053627014f, -0.1010698601603508f, 0.9947231411933899f, -0.10259586572647095f},
    {0.9945645928382874f, -0.104121632874012f, 0.9944036602973938f, -0.10564715415239334f},
    {0.9942404627799988f, -0.1071724221110344f, 0.9940748810768127f, -0.10869744420051575f},
    {0.9939069747924805f, -0.11022220551967621f, 0.993736743927002f, -0.11174671351909637f},
    {0.9935641288757324f, -0.11327095329761505f, 0.9933891892433167f, -0.11479492485523224f},
    {0.9932119250297546f, -0.11631862819194794f, 0.9930323362350464f, -0.11784206330776215f},
    {0.9928504228591919f, -0.11936521530151367f, 0.9926661252975464f, -0.12088808417320251f},
    {0.9924795627593994f, -0.12241067737340927f, 0.9922906160354614f, -0.12393297255039215f},
    {0.9920992851257324f, -0.12545497715473175f, 0.991905689239502f, -0.12697669863700867f},
    {0.9917097687721252f, -0.1284981071949005f, 0.9915114641189575f, -0.13001921772956848f},
    {0.9913108348846436f, -0.13154003024101257f, 0.9911079406738281f, -0.1330605298280716f},
    {0.9909026622772217f, -0.13458070158958435f, 0.9906949996948242f, -0.13610057532787323f},
    {0.9904850721359253f, -0.13762012124061584f, 0.9902728199958801f, -0.1391393393278122f},
    {0.990058183670044f, -0.14065824449062347f, 0.9898412823677063f, -0.1421768069267273f},
    {0.9896219968795776f, -0.14369502663612366f, 0.9894004464149475f, -0.14521291851997375f},
    {0.9891765117645264f, -0.1467304676771164f, 0.988950252532959f, -0.14824767410755157f},
    {0.9887216687202454f, -0.1497645378112793f, 0.9884908199310303f, -0.15128104388713837f},
    {0.9882575869560242f, -0.15279719233512878f, 0.9880220293998718f, -0.15431296825408936f},
    {0.9877841472625732f, -0.15582840144634247f, 0.9875439405441284f, -0.15734346210956573f},
    {0.9873014092445374f, -0.15885815024375916f, 0.9870565533638f, -0.16037245094776154f},
    {0.9868093729019165f, -0.16188639402389526f, 0.9865599274635315f, -0.16339994966983795f},
    {0.9863080978393555f, -0.1649131178855896f, 0.9860539436340332f, -0.1664258986711502f},
    {0.9857975244522095f, -0.16793829202651978f, 0.9855387210845947f, -0.1694502979516983f},
    {0.9852776527404785f, -0.1709618866443634f, 0.9850142598152161f, -0.17247308790683746f},
    {0.9847484827041626f, -0.1739838719367981f, 0.9844804406166077f, -0.1754942536354065f},
    {0.9842100739479065f, -0.17700421810150146f, 0.9839374423027039f, -0.178513765335083f},
    {0.9836624264717102f, -0.18002289533615112f, 0.9833850860595703f, -0.1815316081047058f},
    {0.983105480670929f, -0.18303988873958588f, 0.9828235507011414f, -0.18454773724079132f},
    {0.9825392961502075f, -0.18605515360832214f, 0.9822527170181274f, -0.18756212294101715f},
    {0.9819638729095459f, -0.18906866014003754f, 0.9816727042198181f, -0.1905747503042221f},
    {0.9813792109489441f, -0.19208039343357086f, 0.9810833930969238f, -0.1935855895280838f},
};

// the half in xr / xi (this wave's own: half H) -> the pair's branch H in place, rows [ROW0, ROW0 + 8); the other
// half's rows from `theirs`
template <int H, int ROW0>
__device__ __forceinline__ void radix2_stage(float (&xr)[2 * kRowsH], float (&xi)[2 * kRowsH], const char* theirs, int lane,
                                             const float4 lw) {
  static_for<kRoundRows>([&](auto ii) {
    constexpr int i = ROW0 + decltype(ii)::value;
    const float4 v = *reinterpret_cast<const float4*>(theirs + (i - ROW0) * 1024 + lane * 16);
    if constexpr (H == 0) {                                  // a + b
      xr[2 * i] += v.x; xi[2 * i] += v.y; xr[2 * i + 1] += v.z; xi[2 * i + 1] += v.w;
    } else {                                                 // (a - b) W_4096^n, n = 128 i + 2 l + b: W_32^i W_4096^(2l+b)
      float d0r = v.x - xr[2 * i], d0i = v.y - xi[2 * i], d1r = v.z - xr[2 * i + 1], d1i = v.w - xi[2 * i + 1];
      mul_w32<i>(d0r, d0i);
      mul_w32<i>(d1r, d1i);
      xr[2 * i] = __builtin_fmaf(d0r, lw.x, -(d0i * lw.y));
      xi[2 * i] = __builtin_fmaf(d0r, lw.y, d0i * lw.x);
      xr[2 * i + 1] = __builtin_fmaf(d1r, lw.z, -(d1i * lw.w));
      xi[2 * i + 1] = __builtin_fmaf(d1r, lw.w, d1i * lw.z);
    }
  });
}

// every running sum of the sweep becomes an opaque value at this point of the program: computed before, not re-derived after
__device__ __forceinline__ void pin_sums(Stats& S) {
  asm volatile("" : "+v"(S.sA), "+v"(S.sBh), "+v"(S.sP), "+v"(S.sAA), "+v"(S.sX4), "+v"(S.sAB), "+v"(S.sAP), "+v"(S.sBP));
  asm volatile("" : "+v"(S.sAAA), "+v"(S.sABB), "+v"(S.sAAB), "+v"(S.sBBB), "+v"(S.sAAP), "+v"(S.sX4P), "+v"(S.sABP));
  asm volatile("" : "+v"(S.sa), "+v"(S.st1), "+v"(S.st2), "+v"(S.sab1), "+v"(S.sab2), "+v"(S.sw1), "+v"(S.sw2), "+v"(S.sw3), "+v"(S.sw4));
  asm volatile("" : "+v"(S.Kt), "+v"(S.Kw), "+v"(S.Ka), "+v"(S.th_b1_prev), "+v"(S.wmax));
}

__global__ __launch_bounds__(kThreads, kWavesPerWG / 4) void amcx_features18_pair_kernel(
    const float2* __restrict__ iq, long long n_frames, long long row_stride,
    float* __restrict__ out, long long out_stride) {
  extern __shared__ float4 amcx_pair_smem[];
  char* smem = reinterpret_cast<char*>(amcx_pair_smem);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pr = wave >> 1, me = wave & 1;                   // neighbouring waves form a pair
  char* t2 = smem;
  char* t3 = smem + C2::kT2Bytes;
  char* const ex = smem + kOffRegions + wave * kRegionBytes;            // my region
  const char* const ex_other = smem + kOffRegions + (wave ^ 1) * kRegionBytes;
  float* const stash = reinterpret_cast<float*>(smem + kOffStash) + pr * kStashFloats;
  unsigned* const epoch_mine = reinterpret_cast<unsigned*>(smem + kOffEpoch) + wave;
  unsigned* const epoch_theirs = reinterpret_cast<unsigned*>(smem + kOffEpoch) + (wave ^ 1);
  float* const sa_words = reinterpret_cast<float*>(smem + kOffSa) + 2 * pr;
  unsigned* const chunk_word = reinterpret_cast<unsigned*>(smem + kOffChunk) + pr;
  unsigned* const counters = reinterpret_cast<unsigned*>(smem + kOffCounters);

  build_fft_tables<C2::kFftN>(t2, t3, tid, kThreads);
  if (tid < kWavesPerWG) reinterpret_cast<unsigned*>(smem + kOffEpoch)[tid] = 0;
  if (tid == 0) { counters[0] = 0; counters[1] = 0; }
  __syncthreads();                                           // the only workgroup barrier of the kernel

  constexpr int R = C2::kFftRows;                            // 16
  // Lane-dependent LDS addresses are formed WHERE THEY ARE USED, from a copy of the lane index the compiler cannot see
  // through: as loop invariants it hoists all of them (a dozen registers) out of the frame loop, runs out of registers in
  // the statistics sweep and spills them -- reloading each behind an exposed s_waitcnt vmcnt(0) once per frame.  A shift
  // and an add per address cost less.
  auto lane_here = [&]() {
    int l = lane;
    asm volatile("" : "+v"(l));
    return l;
  };
  auto lane_addr = [&](int l) {
    const int kkL = l >> 3, n3L = l & 7;
    LaneAddr la;
    la.tw2 = t2 + kkL * kTw2Stride;
    la.tw3 = t3 + l * 8;
    la.ex1_w = ex + l * 8;
    la.ex1_r = ex + (kkL * kEx1StrideKK + (n3L & 1) * kEx1StrideB + (n3L >> 1)) * 8;
    la.ex2_w = ex + l * 8;
    la.ex2_r = ex + (n3L * kEx2StrideK2 + kkL * 8) * 8;
    return la;
  };

  // ---- the pair's meetings ----------------------------------------------------------------------
  unsigned epoch = 0;
  bool alive = true;                                         // false: a wait ran into its bound (never, see header)
  auto arrive = [&]() {
    ++epoch;
    if (lane == 0) __hip_atomic_store(epoch_mine, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  auto wait = [&]() {
    for (int spin = 0; spin < kSpinLimit; ++spin) {
      if (__hip_atomic_load(epoch_theirs, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= epoch) return;
      __builtin_amdgcn_s_sleep(1);
    }
    alive = false;
  };
  auto meet = [&]() {
    if (alive) {
      arrive();
      if (AMCX_PAIR_PRIO_MASK != 0) __builtin_amdgcn_s_setprio(0);
      wait();
    }
  };

  // ---- work: the workgroup owns a contiguous slice of frames; its pairs take chunks of it ---------
  const long long per_wg = (n_frames + gridDim.x - 1) / gridDim.x;
  const long long slice0 = (long long)blockIdx.x * per_wg;
  long long slice1 = slice0 + per_wg;
  if (slice1 > n_frames) slice1 = n_frames;
  const long long slice_len = slice1 > slice0 ? slice1 - slice0 : 0;
  const long long tail_len = slice_len < kTailFrames ? slice_len : kTailFrames;
  const long long body_len = slice_len - tail_len;

  typedef float v4f __attribute__((ext_vector_type(4)));

  while (alive) {
    // the even wave asks the workgroup's counters for the pair's next chunk and hands it over through LDS
    if (me == 0) {
      unsigned word = 0;
      if (lane == 0) {
        unsigned got = __hip_atomic_fetch_add(&counters[0], (unsigned)kBatch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((long long)got < body_len) {
          const long long left = body_len - got;
          word = (got << 3) | (unsigned)(left < kBatch ? left : kBatch);
        } else {
          const unsigned t = __hip_atomic_fetch_add(&counters[1], (unsigned)kTailChunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if ((long long)t < tail_len) {
            const long long left = tail_len - t;
            word = ((unsigned)(body_len + t) << 3) | (unsigned)(left < kTailChunk ? left : kTailChunk);
          }
        }
        *chunk_word = word;
      }
    }
    meet();                                                  // (0) the chunk word is there
    if (!alive) break;
    const unsigned word = __builtin_amdgcn_readfirstlane(*chunk_word);
    const int n_here = (int)(word & 7u);
    if (n_here == 0) break;                                  // both waves read the same word: both leave
    const long long f0 = slice0 + (long long)(word >> 3);

    for (int g = 0; g < n_here && alive; ++g) {
      const int h = me ^ (g & 1);                            // the half this wave takes of frame g
      const long long f = f0 + g;
      float xr[2 * kRowsH], xi[2 * kRowsH];
      // ---- phase A: my half from HBM, statistics sweep, rows 0-7 published ----
      {
        const float2* src = iq + f * row_stride + h * kHalf + 2 * lane;
        v4f v[kRowsH];
        static_for<kRowsH>([&](auto ii) {
          constexpr int i = decltype(ii)::value;
          v[i] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(src + 128 * i));
        });
        static_for<kRowsH>([&](auto ii) {
          constexpr int i = decltype(ii)::value;
          xr[2 * i] = v[i].x; xi[2 * i] = v[i].y; xr[2 * i + 1] = v[i].z; xi[2 * i + 1] = v[i].w;
        });
      }
      AMCX_PAIR_PRIO(0);
      Stats S;
      static_for<kRowsH>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        float a0, a1;                                        // |x| is taken again in phase B: the region is the exchange area
        S.template row<i == 0, i == kRowsH - 1>(xr[2 * i], xi[2 * i], xr[2 * i + 1], xi[2 * i + 1], lane, a0, a1);
      });
      // the sweep's sums are pinned HERE: nothing of them is needed before the meeting, and left to itself the compiler
      // sinks most of the sweep behind it, into the envelope sweep and the reduction -- 150 spilled registers
      pin_sums(S);
      {
        char* const mine = ex + lane_here() * 16;
        static_for<kRoundRows>([&](auto ii) {
          constexpr int i = decltype(ii)::value;
          *reinterpret_cast<float4*>(mine + i * 1024) = make_float4(xr[2 * i], xi[2 * i], xr[2 * i + 1], xi[2 * i + 1]);
        });
      }
      {
        const float sa_w = wave_sum_l63(S.sa);
        if (lane == 63) sa_words[me] = sa_w;
      }
      __builtin_amdgcn_sched_barrier(0);
      meet();                                                // (1) rows 0-7 of both halves and both envelope sums are in LDS
      if (!alive) break;
      AMCX_PAIR_PRIO(1);
      if (h == 0) {
        // the phase step that crosses the middle of the frame: half 1's first sample is row 0, lane 0, b = 0 of what the
        // partner has just published (a broadcast read); lane 63 holds this half's last sample, to which row<.., LAST>
        // gave a null step
        const float2 nx = *reinterpret_cast<const float2*>(ex_other);
        const float an = __builtin_amdgcn_sqrtf(__builtin_fmaf(nx.x, nx.x, __builtin_fmaf(nx.y, nx.y, kTinyPower)));
        const float w = wrapped_step(fast_angle(nx.x, nx.y, an), S.th_b1_prev);
        if (lane == 63) {
          S.step(w);
          S.wmax = __builtin_fmaxf(S.wmax, __builtin_fabsf(w));
        }
      }
      // ---- phase B: envelope about the exact mean, sums -> stash, radix-2 stage in two rounds ----
      // (|x| is taken AGAIN from the samples here.  Left to itself the compiler recognises the sweep's square roots
      //  and keeps all 32 of them alive across the meeting instead -- 32 registers the wave does not have: 149 spilled
      //  VGPRs.  The empty asm makes the samples opaque to that.)
#pragma unroll
      for (int e = 0; e < 2 * kRowsH; ++e) asm volatile("" : "+v"(xr[e]), "+v"(xi[e]));
      {
        // half 0's sum first, in both waves: the same float in both
        const float mu = (sa_words[g & 1] + sa_words[1 - (g & 1)]) * (1.0f / (float)kN);
        static_for<2 * kRowsH>([&](auto ee) {
          constexpr int e = decltype(ee)::value;
          S.envelope(__builtin_amdgcn_sqrtf(__builtin_fmaf(xr[e], xr[e], __builtin_fmaf(xi[e], xi[e], kTinyPower))), mu);
        });
        float* const row = stash + (g * 2 + h) * kStashRow;
        float r28[28] = {S.sA, S.sBh, S.sP, S.sAA, S.sX4, S.sAB, S.sAP, S.sBP, S.sAAA, S.sABB,
                         S.sAAB, S.sBBB, S.sAAP, S.sX4P, S.sABP, S.sa, S.sad1, S.sad2, S.sad4,
                         S.st1, S.st2, S.sab1, S.sab2, S.sw1, S.sw2, S.sw3, S.sw4, 0.f};
        const unsigned long long tie = __builtin_amdgcn_ballot_w64(S.wmax > kPi - kTieBand);
        reduce_store(r28, row, lane_here());
        if (lane == 63) {
          row[kNumSums + 1] = S.Kt;
          row[kNumSums + 2] = S.Kw;
          row[kNumSums + 3] = S.Ka;
          row[kNumSums + 4] = tie != 0 ? 1.0f : 0.0f;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      float4 lw = make_float4(1.f, 0.f, 1.f, 0.f);
      if (h == 1) lw = kLaneTwiddle4096[lane_here()];
      if (h == 0) radix2_stage<0, 0>(xr, xi, ex_other, lane_here(), lw); else radix2_stage<1, 0>(xr, xi, ex_other, lane_here(), lw);
      __builtin_amdgcn_sched_barrier(0);
      meet();                                                // (2) round 1 has been read
      if (!alive) break;
      AMCX_PAIR_PRIO(2);
      {
        char* const mine = ex + lane_here() * 16;
        static_for<kRoundRows>([&](auto ii) {
          constexpr int i = decltype(ii)::value;
          // (rows 8-15 of the ORIGINAL half: the stage above rewrote rows 0-7 only)
          *reinterpret_cast<float4*>(mine + i * 1024) =
              make_float4(xr[2 * (kRoundRows + i)], xi[2 * (kRoundRows + i)], xr[2 * (kRoundRows + i) + 1], xi[2 * (kRoundRows + i) + 1]);
        });
      }
      meet();                                                // (3) rows 8-15 of both halves are in LDS
      if (!alive) break;
      AMCX_PAIR_PRIO(2);
      if (h == 0) radix2_stage<0, kRoundRows>(xr, xi, ex_other, lane_here(), lw); else radix2_stage<1, kRoundRows>(xr, xi, ex_other, lane_here(), lw);
      __builtin_amdgcn_sched_barrier(0);
      meet();                                                // (4) round 2 has been read: my region is my FFT scratch now
      if (!alive) break;
      if (AMCX_PAIR_PRIO_MASK != 0) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      // ---- phase C: 2048-point register FFT of my branch, its peak into my stash row ----
      {
        const LaneAddr la = lane_addr(lane_here());
        const float peak = fft_peak<R>(xr, xi, la);
        const float pk = wave_max_l63(peak);
        if (lane == 63) stash[(g * 2 + h) * kStashRow + kNumSums] = pk;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    meet();                                                  // (5) every peak of the chunk is in the stash
    if (!alive) break;

    // ---- finalisation: this wave takes frames me, me + 2 of the chunk, one per lane ----
    {
      float feat[18];
      bool tie = false;
      float kw0 = 0.f;
      const int g_mine = 2 * lane + me;                      // lane j <-> frame 2 j + me
      const bool have = g_mine < n_here;
      if (have) {
        const float* rows = stash + g_mine * 2 * kStashRow;  // half 0's row, then half 1's
        auto sm = [&](int k) -> double { return (double)rows[k] + (double)rows[kStashRow + k]; };
        FrameSums F;
        F.sA = sm(0); F.sBh = sm(1); F.sP = sm(2); F.sAA = sm(3); F.sX4 = sm(4); F.sAB = sm(5);
        F.sAP = sm(6); F.sBP = sm(7); F.sAAA = sm(8); F.sABB = sm(9); F.sAAB = sm(10);
        F.sBBB = sm(11); F.sAAP = sm(12); F.sX4P = sm(13); F.sABP = sm(14);
        F.sa = sm(15); F.sad1 = sm(16); F.sad2 = sm(17); F.sad4 = sm(18);
        // shifted sums: re-centred about half 0's shifts
        F.Kt = rows[28]; F.Kw = rows[29]; F.Ka = rows[30];
        kw0 = rows[29];
        Recentred th, ab, ws;
        float pk = 0.f;
        bool flagged = false;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const float* r = rows + hh * kStashRow;
          th.add((double)kHalf, (double)r[28] - F.Kt, r[19], r[20]);
          ab.add((double)kHalf, (double)r[30] - F.Ka, r[21], r[22]);
          ws.add(hh == 1 ? (double)(kHalf - 1) : (double)kHalf, (double)r[29] - F.Kw, r[23], r[24], r[25], r[26]);
          pk = __builtin_fmaxf(pk, r[27]);
          if (!(r[27] == r[27])) pk = r[27];                 // a NaN peak (non-finite sample) must survive the maximum
          flagged = flagged || r[31] != 0.0f;
        }
        F.std1 = th.s1; F.std2 = th.s2; F.sab1 = ab.s1; F.sab2 = ab.s2;
        F.swd1 = ws.s1; F.swd2 = ws.s2; F.swd3 = ws.s3; F.swd4 = ws.s4;
        F.gmax_raw = pk;
        F.pi_tie = flagged;
        finalize_features(F, kN, feat);
        if (is_outside_fp32_range(F, kN)) feat[4] = -__builtin_inff();       // redone by amcx_range_fixup_kernel
        tie = __builtin_signbitf(feat[4]) && feat[4] == feat[4] && feat[4] != -__builtin_inff();
      }
      unsigned long long ties = __builtin_amdgcn_ballot_w64(tie);
      while (ties != 0) {                                    // phase steps within an fp32 ulp of +-pi: exact f5 / f9
        const int idx = __builtin_ctzll(ties);
        ties &= ties - 1;
        const float kwt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, kw0), idx));
        float f5x, f9x;
        wave_exact_frequency<kN>(iq + (f0 + 2 * idx + me) * row_stride, 1.0f, kwt, lane, f5x, f9x);
        if (lane == idx) { feat[4] = f5x; feat[8] = f9x; }
      }
      if (have) {
        float* dst = out + (f0 + g_mine) * out_stride;
#pragma unroll
        for (int j = 0; j < 18; ++j) dst[j] = feat[j];
      }
    }
    // (the next chunk's stash rows are written behind meeting (1) of its first frame, which this wave joins only
    //  after the stores above: the partner cannot overwrite a row that is still being read here)
  }
  if (!alive) {
    // a wait ran into its bound: leave a mark nobody can take for features.  The frames of the chunk in flight keep
    // whatever the caller's buffer held; the first frame of this workgroup's slice says so.
    if (lane == 0 && slice_len > 0) out[slice0 * out_stride] = __builtin_nanf("");
  }
}

inline hipError_t launch_pair(const float2* iq, int64_t n_frames, int64_t row_stride, float* out,
                              int64_t out_stride, hipStream_t stream, int cus) {
  static bool lds_attr_set[64] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64 || !lds_attr_set[dev]) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(amcx_features18_pair_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) lds_attr_set[dev] = true;      // benign race: idempotent
  }
  int64_t grid = (int64_t)cus;                               // persistent: one resident workgroup per CU
  if (grid * kPairs > n_frames) grid = (n_frames + kPairs - 1) / kPairs;     // at least a frame per pair
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(amcx_features18_pair_kernel, dim3((unsigned)grid), dim3(kThreads), kLdsBytes, stream, iq,
                     (long long)n_frames, (long long)row_stride, out, (long long)out_stride);
  return hipGetLastError();
}

}  // namespace pair
}  // namespace amcx
