// stft_kernel.hpp -- the STFT hot path as one HIP kernel template for gfx950.
//
// What it replaces: the platform code behind analyser.getByteFrequencyData()
// (reference call site 3D/visualizer.js:358; configuration UI/player.js:10-11),
// in the form BASELINE.json's north_star prescribes: window -> hop-overlapped
// framing -> FFT -> |X|^2 / log-magnitude per hop.
//
// Shape of the computation:
//   * A real N-point frame is packed as M = N/2 complex points z[n] = x[2n] + i x[2n+1].
//   * T = M/8 threads own one frame; thread t holds the 8 points n = t + T*j in
//     registers.  With hop = S * 2T samples the next frame's points j = 0..7-S
//     are this frame's j = S..7 IN THE SAME THREAD, so a thread that walks a run
//     of consecutive frames loads each PCM sample from memory exactly once
//     (register-resident sliding window: no overlap re-reads at all).  The frame
//     loop is unrolled over the rotation period 8/S so the window never moves
//     between registers; the prefetch of the next hop lands in the slots the
//     current frame has just finished with.
//   * The M-point FFT is a Stockham autosort: radix-8 passes done in registers,
//     each followed by an exchange through LDS; every pass reads in[t + T*r], so
//     reads are always lane-contiguous.  After the last pass thread t holds
//     Z[t + T*r] in natural order.  Window and inter-pass twiddles are fused
//     into the first butterfly stage of each pass as FMAs.
//   * Real-input post-process pairs bin k = t + T*i with M-k: one fetch of the
//     partner through LDS gives both X[k] and X[M-k].
//   * Epilogue fused: |X|^2, |X|, dB or the AnalyserNode byte, stored coalesced.
// Not a dense contraction: no MFMA.  VALU work is ~270 scalar-f32 operations per
// lane per frame (packed f32 ops cost two issue cycles each on gfx950, so they
// are not cheaper than two scalar ops); LDS is used only for the transposes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

#include "pk_blocks.hpp"
#include "xlane.hpp"

namespace sg {

enum : int { OUT_POWER = 0, OUT_MAG = 1, OUT_DB = 2, OUT_DB_U8 = 3, OUT_LOGFREQ = 4, OUT_ANALYSER = 8, OUT_SMOOTH = 16, OUT_PEND = 32, OUT_LFFAST = 64 };
// OUT_LOGFREQ | OUT_LFFAST: the row is exactly the 256 columns of the consecutive-column path (N = 1024: four per lane, N = 2048:
// two) and the caller takes all of them -- the reference's row (sonogram3DWidth = 256, 3D/visualizer.js:54).  Decided at launch, so
// that the kernel carries one column path instead of both behind a run-time test (136 -> 124 VGPRs at N = 1024: four waves per SIMD).
// OUT_ANALYSER | OUT_PEND: the fused push + poll launch (sg_push_poll_*): the newest samples of the window are still in the caller's
// block.  An instantiation of its own, so that the plain poll kernels do not carry its loads and stores (round 5: they cost the
// N = 4096 poll kernel its second wave per SIMD).
// kind | OUT_SMOOTH: batch-mode "smoothing over time" fused behind the transform -- a run is a whole channel and the
// smoothed magnitudes of a lane's bins ride in registers from frame to frame (no |X| scratch, no second launch).
// kind | OUT_SMOOTH | OUT_LOGFREQ: ... and the row that leaves is the log-frequency columns of the SMOOTHED bins (the order
// the reference's display applies them: the analyser smooths, 3D/visualizer.js:351, the shader samples the row it hands out).
// kind | OUT_LOGFREQ: log-frequency columns.  OUT_ANALYSER | OUT_DB / OUT_DB_U8: the poll-mode epilogue
// (AnalyserNode smoothing state, 20 log10, float dB or byte, history-ring row) fused behind the transform;
// | OUT_LOGFREQ on top of that: the poll hands out (and the ring keeps) the log-frequency columns of that row.

// packed little-endian 24-bit PCM sample (WAV "PCM 24"): three bytes, no alignment
struct s24 {
  unsigned char b[3];
  __device__ __forceinline__ operator float() const {
    return (float)(((int)(signed char)b[2] << 16) | ((int)b[1] << 8) | (int)b[0]);
  }
};
static_assert(sizeof(s24) == 3, "packed 24-bit sample");

// ring / load element: one complex point z[n] = (x[2n], x[2n+1]).  FMT: 0 float32, 1 int16, 2 packed int24.
template <int FMT> struct PcmT;
template <> struct PcmT<0> {
  typedef f2 point;      // two float32 samples
  typedef float sample;
  static __device__ __forceinline__ f2 cvt(f2 v) { return v; }
};
template <> struct PcmT<1> {
  typedef int point;     // two int16 samples packed in one dword (1/32768 is folded into the window)
  typedef short sample;
  static __device__ __forceinline__ f2 cvt(int v) { return f2{(float)(short)(v & 0xffff), (float)(v >> 16)}; }
};
template <> struct PcmT<2> {
  typedef f2 point;      // generic loads only (a 6-byte point has no aligned ring element): converted while loading,
  typedef s24 sample;    // 1/8388608 folded into the window
  static __device__ __forceinline__ f2 cvt(f2 v) { return v; }
};

struct StftParams {
  const void* pcm;
  void* out;
  const f2* tw;   // tw[k] = (cos, -sin)(2 pi k / N), k in [0, N); N >= 4096: the chosen (W^1, W^2, W^4) triplets of every pass behind them
  const f2* win;  // win[n] = c * (w[2n], w[2n+1]), n in [0, M); c folds 1/2 and the scale
  long long n_channels;
  long long frames;          // per channel
  long long channel_stride;  // floats
  long long sample_stride;   // floats
  long long out_channel_stride;  // output elements
  long long runs_per_channel;
  long long total_runs;
  int hop;
  int run_len;   // frames per run
  int bins;      // M or M + 1
  int nyquist;
  float db_mul;  // 10 / log2(10)
  float floor_db;
  float byte_a;      // db_mul * 255 / (max_db - min_db)
  float byte_b;      // -min_db * 255 / (max_db - min_db)
  float byte_floor;  // max((floor_db - min_db) * 255 / (max_db - min_db), 0): lower clamp of the byte value
  int downmix;       // K interleaved input samples averaged per analysed sample (generic load path), 0/1 = off
  float downmix_scale;  // 1 / K
  const float* downmix_w;  // null: the mean.  Else K weights (SG_DOWNMIX_SPEAKERS where it is not the mean: 5.1 -> mono, or
                           // channel 0 alone for layouts the W3C rules leave to "discrete")
  const uint2* lf_tab;  // log-frequency lookup: {4*i0 | 4*i1 << 16 (byte offsets), frac as float bits} per column
  int lf_cols;
  // OUT_ANALYSER (poll mode: one frame per channel, N/2 bins)
  float* an_state;         // [channels][N/2] smoothed magnitudes, read-modify-written
  unsigned char* an_ring;  // this poll's row of the byte history (channel 0), or null
  long long an_ring_stride;  // bytes between channels in the ring
  float an_tau;            // smoothingTimeConstant
  float an_min_db;         // minDecibels
  float an_byte_mul;       // 255 / (maxDecibels - minDecibels)
  int an_len;              // bins the caller's buffer takes per channel (<= N/2)
  int prio_rem;            // 1: waves rank themselves by the part of their run still ahead (launches of a single round)
  long long in_extent;     // samples of `pcm` the caller vouches for, counted from pcm[0] (LDS-ring loads never read past it)
  int host_extra_lds;      // host side only (sg_debug_set "extra_lds"): extra dynamic LDS per workgroup, an occupancy throttle for A/B runs
  // OUT_ANALYSER, fused push + poll (sg_push_poll_*): window samples [pend_split, N) of every channel are not in the history
  // yet -- the kernel takes them from the caller's block (page-locked host memory read in place, or device memory) and
  // puts them where they belong in the history row on its way
  const void* pend_src;    // null: everything is in the history already
  long long pend_stride;   // samples between the channels' blocks
  int pend_split;          // N - n_new (even)
  int pend_s16;            // the block holds int16 samples (stored in the history as the float of the integer)
  int pend_wstart;         // position of the window's first sample in its 2 H-float history row (a double-written ring)
  int pend_h;              // H: a sample at row position q has its other copy at q - H (q >= H) or q + H
};

#define SG_FMA(a, b, c) __builtin_fmaf((a), (b), (c))

// ---- 8-point DFT, forward sign, natural order in and out, split in two so that the
// first butterfly stage can absorb either the window (below) or the inter-pass twiddles
// (dft8_stage1_tw_blk); stages 2+3 are dft8_stage23_blk; a twiddled pass is both in one block (dft8_tw_pass_blk).  All complex arithmetic is packed
// (pk_math.hpp / pk_blocks.hpp): 12 + 18 instructions for the first pass, 18 + 18 for a twiddled pass.

// stage 1 with the (real) window fused: a_r = x_r w_r + x_{r+4} w_{r+4}, a_{r+4} = x_r w_r - x_{r+4} w_{r+4}.
// `rot` is the physical rotation of the sliding register window (compile-time after inlining).
// Virtual thread i of a lane reads ring slots (i + VW r + rot) % RING, r = 0..7.
template <int RING, int FMT, int VW>
__device__ __forceinline__ void dft8_stage1_win(f2 (&v)[8], const typename PcmT<FMT>::point (&x)[RING],
                                                const f2 (&w)[8], int rot, int i) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const f2 xa = PcmT<FMT>::cvt(x[(i + VW * r + rot) % RING]);
    const f2 xb = PcmT<FMT>::cvt(x[(i + VW * (r + 4) + rot) % RING]);
    const f2 t = xa * w[r];
    const f2 pb = xb * w[r + 4];  // contracted into the two FMAs below
    v[r] = t + pb;
    v[r + 4] = t - pb;
  }
}

__device__ __forceinline__ void dft4(f2& x0, f2& x1, f2& x2, f2& x3) {
  const f2 a0 = x0 + x2, a2 = x0 - x2;
  const f2 a1 = x1 + x3, d = x1 - x3;
  x0 = a0 + a1;
  x2 = a0 - a1;
  x1 = pk_add_mi(a2, d);
  x3 = pk_add_pi(a2, d);
}

// LDS index padding: one spare element per 8 keeps the radix-8 exchange writes (stride-8 element bursts per
// lane) off each other's banks.  Kept for the small sizes (fewer than 16 virtual threads per frame); everything
// else uses the layouts gfx950's 16-lane / 32-bank service of 64-bit accesses wants: SG_PAD16 for the first
// exchange, SG_SWIZZLE for the later ones, plain linear for the partner exchange (no conflicts at all).
__device__ __forceinline__ int pad(int i) { return i + (i >> 3); }
// pad(x + d) == pad(x) + padd(d) whenever d is a multiple of 8 (or x is and 0 <= d < 8): lets every
// exchange address be one per-lane base plus a compile-time offset (the ds_* immediate).
constexpr int padd(int d) { return d + d / 8; }

#ifndef SG_WG_SMALL
#define SG_WG_SMALL 256
#endif

#ifndef SG_XLANE_MW
#define SG_XLANE_MW 1  // bit 0: frames of 2 waves (N = 2048), bit 1: of 4 waves (N = 4096): both in-wave exchanges are cross-lane
                       // transpositions and only the wave bits go through LDS (one exchange + one barrier instead of three
                       // of each).  Measured: N = 2048 +2 % (f32) / +5 % (bytes), N = 4096 +-0 against the swizzled LDS
                       // exchanges: on for 2048 only.
#endif

// Geometry.  A frame's M = N/2 complex points are owned by TV = M/8 *virtual* threads of 8 points
// each (point n of virtual thread j: n = j + TV r).  A lane carries VW virtual threads
// (j = tp + T i, i < VW), so a frame occupies T = TV / VW lanes.  VW = 2 brings N = 32768
// (TV = 2048), the AnalyserNode maximum, within the 1024-thread workgroup limit.
template <int LOG2N>
struct Geo {
  static constexpr int N = 1 << LOG2N;
  static constexpr int M = N / 2;
  static constexpr int LOG2M = LOG2N - 1;
  static constexpr int TV = M / 8;                   // virtual threads per frame
  static constexpr int VW = LOG2N == 15 ? 2 : 1;
  static constexpr int T = TV / VW;                  // lanes per frame
  static constexpr int WG = T > 64 ? T : SG_WG_SMALL;  // workgroup: one multi-wave frame, or 256 lanes of
                                                     // single-wave (or smaller) frames
  static constexpr int SLOTS = WG / T;               // frames in flight per workgroup
  static constexpr bool MULTIWAVE = T > 64;          // a frame spans several waves
  static constexpr int P8 = LOG2M / 3;               // radix-8 passes
  static constexpr int REM = LOG2M % 3;              // final radix 2^REM pass if != 0
  static constexpr int BUF = M + M / 8;              // padded elements per exchange buffer
  // ping-pong buffers where barriers are needed and LDS allows; otherwise one buffer, two barriers
  static constexpr int NBUF = (MULTIWAVE && 2 * SLOTS * BUF * 8 <= 160 * 1024) ? 2 : 1;
  // frames of 2 / 4 waves whose in-wave exchanges are cross-lane transpositions (XLM in the kernel): buffer 1 only ever
  // carries the partner exchange (M / 2 points), so it is that small -- LDS, not registers, is what limits the
  // occupancy of these sizes once a sample ring sits beside the buffers
  static constexpr bool XLM = MULTIWAVE && VW == 1 && P8 == 3 && REM != 0 &&
                              ((T == 128 && (SG_XLANE_MW & 1)) || (T == 256 && (SG_XLANE_MW & 2)));
  static constexpr int BUF1 = XLM ? M / 2 : BUF;             // elements of the second buffer
  static constexpr int SLOT_ELEMS = BUF + (NBUF == 2 ? BUF1 : 0);
  static constexpr int LDS_BYTES = SLOTS * SLOT_ELEMS * 8;
};

template <bool MULTIWAVE>
__device__ __forceinline__ void frame_sync() {
  if constexpr (MULTIWAVE) {
    __syncthreads();
  } else {
    // All threads of a frame live in one wave; a wave's DS operations execute
    // in order, so only the compiler has to be kept from reordering them.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

template <int V>
struct IC {
  static constexpr int value = V;
};
// f(IC<0>{}), f(IC<1>{}), ... f(IC<N-1>{}): a loop whose index is a constant expression (asm immediates)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(IC<I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// Diagnostic ablations (tools/ only; never defined in the product build):
//   bit 0: drop the global stores   bit 1: drop the LDS exchanges   bit 2: drop the butterflies
//   bit 3: drop the per-frame prefetch loads
#ifndef SG_ABLATE
#define SG_ABLATE 0
#endif
__device__ __forceinline__ void keep(float x) { asm volatile("" ::"v"(x)); }
#ifndef SG_MIN_WAVES
#define SG_MIN_WAVES 1
#endif
#ifndef SG_PREFETCH_DEPTH
#define SG_PREFETCH_DEPTH 2  // frames of cover for the PCM loads (HBM latency under load exceeds one frame)
#endif

#ifndef SG_DERIVE_TW
#define SG_DERIVE_TW 1  // frames of more than two waves keep W^1, W^2, W^4 per pass and derive the other four (8 packed
                        // instructions per pass) to stay at 4 waves/SIMD: N=4096 +3 %, N=8192 +24 % over holding all
                        // seven.  N=2048 holds all seven (126 VGPRs either way): +1-4 % (16 instructions fewer a frame-wave)
#endif
#ifndef SG_XLANE
#define SG_XLANE 2  // one-wave frames (T == 64), bit p: the exchange after radix-8 pass p is a register <-> lane
                    // transposition on DPP / v_permlane*_swap (xlane.hpp) instead of an LDS round trip
#endif
#ifndef SG_PRIO_REM
#define SG_PRIO_REM 1  // the remaining-work issue priority is compiled in; StftParams::prio_rem turns it on per launch
                       // (launches of one dispatch round; large batches run without: -3 % there)
#endif
#ifndef SG_STAMP
#define SG_STAMP 0
#endif
#ifndef SG_STALL
#define SG_STALL 0  // diagnostic builds only (tools/stalls.py): which wait of the two-wave frame of stft_kernel<11,2,0> sits between
                    // two s_memtime stamps -- 0 off; 1 the head's vmcnt wait, 2 the in-wave transpose, 3 / 4 / 5 writes / barrier /
                    // reads of the wave-bit exchange, 6 / 7 / 8 of the partner exchange, 9 the whole frame, 10 nothing (the
                    // stamp pair's own cost)
#endif
#ifndef SG_POLL12_ALL7
#define SG_POLL12_ALL7 1  // the poll-mode kernels of N = 4096 (config 4's size: one frame per channel, bound by the hop's latency)
                          // hold all seven table twiddles of a pass (round 4).  150 VGPRs, 228 for the fused push + poll
                          // instantiation: two waves per SIMD, which the bank's 512 workgroups need to be resident at once
#endif
#ifndef SG_SWIZZLE
#define SG_SWIZZLE 1  // exchanges after the first: XOR-swizzled layout instead of one spare element per 8: conflict-free
                      // reads as well as writes (N=4096 +5 %, N=8192 +12 %, N=512 +3.5 %)
#endif
#ifndef SG_PAD16
#define SG_PAD16 1  // first-exchange LDS layout: one spare element per 16 (0: per 8, which leaves each 16-lane read group
                    // one 2-way conflict: 40 % of the N=1024 kernel's LDS cycles)
#endif
#ifndef SG_KEEP_SPLATS
#define SG_KEEP_SPLATS 1  // the (2, 2) and (1/sqrt 2, 1/sqrt 2) operands stay in VGPR pairs instead of being rematerialised
#endif
#ifndef SG_XLM_EX1_LDS
#define SG_XLM_EX1_LDS 1  // multi-wave cross-lane frames: exchange 1 (lane bits 0..2) as an in-wave LDS transpose
                          // (8 LDS instructions, no barrier) instead of 48 DPP selects: N=2048 +10 % with the layouts above
#endif
#ifndef SG_NT
#define SG_NT 1  // bit 0: non-temporal output stores (+3-4 % on the 112k-frame workload, +1 % on large
                 // batches), bit 1: non-temporal PCM loads (no gain: the run halo wants its L2 hits); bits 2 / 3:
                 // output stores written through the XCD's L2 at device / system scope (sc1, sc0 sc1) -- 15-25 % slower,
                 // the L2's write combining is worth more than a clean L2 at the end of the kernel (docs/EXPERIMENTS.md section 5.5)
#endif
template <typename Ptr, typename Tp>
__device__ __forceinline__ void st_out(Ptr ptr, Tp v) {
  if constexpr (SG_NT & 4) __hip_atomic_store(ptr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // sc1: written through the XCD's L2
  else if constexpr (SG_NT & 8) __hip_atomic_store(ptr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // sc0 sc1
  else if constexpr (SG_NT & 1) __builtin_nontemporal_store(v, ptr);
  else *ptr = v;
}
template <typename Ptr>
__device__ __forceinline__ auto ld_pcm(Ptr ptr) {
  if constexpr (SG_NT & 2) return __builtin_nontemporal_load(ptr);
  else return *ptr;
}

constexpr int sg_gcd(int a, int b) { return b == 0 ? a : sg_gcd(b, a % b); }

// Global-memory pointers with an explicit address space: a pointer rebuilt from integers would
// otherwise be generic and turn every access into flat_load / flat_store.
template <typename Tp> using gptr = __attribute__((address_space(1))) Tp*;

// Pins a wave-uniform pointer in an SGPR pair and hides where it came from, so that address
// arithmetic at the use stays "SGPR base + 32-bit VGPR offset + immediate" (the saddr form of
// global_load / global_store) instead of being strength-reduced into per-lane 64-bit VGPR pointers
// (two VGPRs and a v_lshl_add_u64 per pointer per unrolled frame).  Not uniform: passed through.
template <bool UNIFORM, typename Tp>
__device__ __forceinline__ gptr<Tp> sgpr_base(Tp* ptr) {
  if constexpr (UNIFORM) {
    unsigned long long v = reinterpret_cast<unsigned long long>(ptr);
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    asm volatile("" : "+s"(lo), "+s"(hi));
    return (gptr<Tp>)(((unsigned long long)hi << 32) | lo);
  } else {
    return (gptr<Tp>)ptr;
  }
}

// The same for "uniform base pointer + uniform byte offset": only the OFFSET goes through the opaque SGPR pair, the
// pointer arithmetic stays a GEP on the real (kernel-argument) pointer, which is what instruction selection needs to
// pick the saddr form  global_load/store v_off, s[base:base+1] offset:imm  -- no VALU address math, no 64-bit
// per-lane offsets in VGPR pairs.
template <bool UNIFORM, typename Tp>
__device__ __forceinline__ gptr<Tp> sgpr_at(Tp* base, long long byte_off) {
  if constexpr (UNIFORM) {
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)byte_off);
    unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)byte_off >> 32));
    asm volatile("" : "+s"(lo), "+s"(hi));
    const long long o = (long long)(((unsigned long long)hi << 32) | lo);
    return (gptr<Tp>)((gptr<unsigned char>)reinterpret_cast<unsigned char*>(const_cast<typename std::remove_const<Tp>::type*>(base)) + o);
  } else {
    return (gptr<Tp>)(reinterpret_cast<unsigned char*>(const_cast<typename std::remove_const<Tp>::type*>(base)) + byte_off);
  }
}

// ... and the 32-bit VGPR offset has to stay 32-bit at the use: left alone, the loop-invariant zero-extension is
// hoisted as a 64-bit VGPR pair and the access falls back to a VGPR-pair address (v_lshl_add_u64 per base).
// In place on the kernel-lifetime variable (a by-value copy would cost a v_mov per use).
__device__ __forceinline__ unsigned opaque_off(unsigned& x) {
  asm volatile("" : "+v"(x));
  return x;
}

// Per-lane constants of a frame slot (window pairs, inter-pass twiddles, final-pass twiddles, post-process twiddles): how many
// of each a lane holds in registers for its run.
template <int LOG2N>
struct Consts {
  using G = Geo<LOG2N>;
  static constexpr bool DERIVE = SG_DERIVE_TW && (G::MULTIWAVE || G::VW > 1) && LOG2N != 11;
  static constexpr int NTW = DERIVE ? 3 : 7;             // stored twiddles per pass: all seven, or W^1, W^2, W^4
  static constexpr int NTWF = G::REM == 2 ? 6 : (G::REM == 1 ? 4 : 0);
};

// Launch geometry that depends on the load variant as well as on the size.
template <int LOG2N, int SHIFT>
struct Plan {
  using G = Geo<LOG2N>;
  static constexpr bool SLIDE = SHIFT == 2 || SHIFT == 4;  // register-resident sliding window
  static constexpr bool LRING = SHIFT == 1 || SHIFT == 3;  // per-slot LDS sample ring (any hop <= N - 4, unit stride)
  static constexpr int KR = SHIFT == 1 ? 2 : 4;            // LRING: rows of T 4-sample granules a frame may bring in
  static constexpr int PP = 8 * G::VW;                     // points a lane holds per frame
  static constexpr int SV = SLIDE ? SHIFT * G::VW : 0;     // ... of which a hop retires this many
  static constexpr int RING = SLIDE ? PP + SV * (SG_PREFETCH_DEPTH - 1) : PP;
#ifndef SG_LR_ASM
#define SG_LR_ASM 1  // window reads of the LDS ring as hand-placed ds_read instructions (one per point, the point's two
                     // samples landing in an aligned VGPR pair).  Left to the compiler the reads are paired by ROW
                     // (ds_read2st64 of rows m, m + 1) and 2-8 v_mov per frame put the points back together.
#endif
#ifndef SG_LR_LATE
#define SG_LR_LATE 1
#endif
  static constexpr int PERIOD = SLIDE ? RING / sg_gcd(RING, SV) : 1;  // frames per unrolled trip
  // LDS sample ring of a frame slot: C >= N + 4 samples (the window plus one granule of slack); the plain layout adds
  // the mirror of sample 0 (a complex point that starts on the last sample reads its second half there)
  // Two ring layouts.  Plain: the samples as they come, C = N + 4, one spare float.  Split (the two-wave frames whose
  // lanes own bit-transposed point columns, N = 2048): sample s at region s & 3, index phi(s >> 2), C = N + 256 so that
  // a region's length is a multiple of the 64 indices phi permutes -- conflict-free for that lane map.
  static constexpr bool LR_SPLIT = G::XLM && G::T == 128;
  static constexpr int LR_C = LR_SPLIT ? G::N + 256 : G::N + 4;
  static constexpr int LR_ALLOC = LR_SPLIT ? LR_C : LR_C + 4;
  static constexpr int LR_BYTES = LRING ? G::SLOTS * LR_ALLOC * 4 : 0;
  // exchange buffers, then the sample rings
  static constexpr int LR_OFFSET = G::LDS_BYTES;
  static constexpr int LDS_BYTES = LR_OFFSET + LR_BYTES;
};

// SHIFT: 0 = generic loads (any hop, any stride); S in {2,4} = sliding window, hop == S * 2 TV
// (hop = N/4 or N/2), unit stride, rows aligned to one complex point.
//
// Register ring: a lane holds the PP = 8 VW points n = tp + T m (m < PP) of the current frame;
// logical point m of the frame at unrolled position u sits in slot (m + u SV) % RING.  After the
// first butterfly stage has read the window, the SV oldest slots are dead and receive the loads of
// logical points RING .. RING + SV - 1 (frame f + DEPTH), so nothing ever moves between registers
// and every PCM sample is loaded from HBM exactly once.
// Waves per SIMD the register allocator must leave room for.  Poll mode at N = 4096 (config 4: 512 workgroups of four waves on
// 256 CUs) needs TWO -- both workgroups of a CU resident at once.  (Round 5: with the fused push + poll loads and stores behind
// a RUN-TIME test in the one poll kernel, the allocator settled at 278 registers and the bank ran in two rounds, +3.5 us per
// hop; as an instantiation of its own -- OUT_PEND -- the plain poll kernel takes 150 and the fused one 228.)
constexpr int min_waves_per_simd(int log2n, int out_full) {
  return ((out_full & OUT_ANALYSER) && log2n == 12) ? 2 : SG_MIN_WAVES;
}

template <int LOG2N, int SHIFT_FULL, int OUT_FULL>
__global__ __launch_bounds__(Geo<LOG2N>::WG, min_waves_per_simd(LOG2N, OUT_FULL)) void stft_kernel(const StftParams p) {
  constexpr int SHIFT = SHIFT_FULL & 7;         // load variant
  constexpr int FMT = (SHIFT_FULL >> 3) & 3;    // sample format: 0 float32, 1 int16, 2 packed int24
  static_assert(FMT != 2 || SHIFT == 0 || SHIFT == 1 || SHIFT == 3, "int24 input runs on the generic loads or through the LDS ring");
  using PT = PcmT<FMT>;
  typedef typename PT::point point_t;
  typedef typename PT::sample sample_t;
  constexpr int OUT = OUT_FULL & 3;             // value kind
  constexpr bool LOGF = (OUT_FULL & OUT_LOGFREQ) != 0;  // log-frequency columns instead of bins
  constexpr bool LFFAST = (OUT_FULL & OUT_LFFAST) != 0;  // ... and the launcher vouches for the consecutive-column shape
  static_assert(!LFFAST || LOGF, "OUT_LFFAST qualifies OUT_LOGFREQ");
  constexpr bool ANALYSER = (OUT_FULL & OUT_ANALYSER) != 0;  // poll mode: smoothing + dB / byte + ring row
  constexpr bool PEND = (OUT_FULL & OUT_PEND) != 0;          // ... fused with the push: window samples [pend_split, N) come from the caller's block
  static_assert(!PEND || ANALYSER, "the fused push belongs to poll mode");
  constexpr bool SMOOTH = (OUT_FULL & OUT_SMOOTH) != 0;      // X^[f] = tau X^[f-1] + (1 - tau) |X[f]| along the run
  static_assert(!(SMOOTH && ANALYSER), "fused batch smoothing is a batch-mode epilogue (poll mode smooths on the handle's state)");
  static_assert(!(SMOOTH && LFFAST), "the 256-column instantiations carry no smoothing state");
  using G = Geo<LOG2N>;
  using PL = Plan<LOG2N, SHIFT>;
  constexpr int N = G::N, M = G::M, T = G::T, TV = G::TV, VW = G::VW, SLOTS = G::SLOTS;
  constexpr int PP = PL::PP, SV = PL::SV, RING = PL::RING, PERIOD = PL::PERIOD;
  constexpr bool SLIDE = PL::SLIDE, LRING = PL::LRING;
  // load variant 5: the generic loads with unit stride, float32 samples and no down-mix known at LAUNCH -- the instantiation
  // carries the one 8-byte load per point alone (round 5: with the strided / down-mix loop beside it behind a run-time test the
  // generic kernels took 148-154 VGPRs, three waves per SIMD; alone 100-112, four -- and N = 16384 no longer spills)
  constexpr bool GEN_UNIT = SHIFT == 5;
  static_assert(!GEN_UNIT || FMT == 0, "load variant 5 is float32");
  constexpr int KR = PL::KR, LR_C = PL::LR_C, LR_Q = PL::LR_C / 4;
  constexpr bool LR_SPLIT = PL::LR_SPLIT;
  static_assert(!LRING || (G::T >= 64 && G::VW == 1), "the LDS sample ring serves whole-wave frame slots");
  constexpr bool TWO_SYNC = G::MULTIWAVE && G::NBUF == 1;  // single buffer: barrier after the reads too
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

  // XL: the frame is exactly one wave and the passes trade index bits between registers and lanes
  // with cross-lane VALU ops.  Pass 0 then wants the lane's low / high 3 bits swapped in the point
  // index (lane l owns points j0 + T r, j0 = (l & 7) << 3 | l >> 3), so that after the two
  // transpositions (REG <-> lane bits 0..2, then REG <-> lane bits 3..5) lane l holds Z[l + T r]
  // exactly as the LDS path leaves it.  Twiddle indices k = l & (Ns - 1) are the same in both.
  constexpr bool XL_OK = !G::MULTIWAVE && T == 64 && VW == 1 && G::P8 == 3 && G::REM == 0;
  // XLM: a frame of 2 or 4 waves (T = 128, 256; M = 8^3 * 2 or 8^3 * 4).  The point index of a thread has 6 lane
  // bits and NW wave bits; passes 1 and 2 take their three bits from the lanes of a wave (cross-lane, no LDS, no
  // barrier), and only the NW bits the final radix-2 / radix-4 pass needs cross waves, through LDS once.
  constexpr bool XLM = G::XLM;
  constexpr int NW = XLM ? (T == 128 ? 1 : 2) : 0;  // wave bits of the thread index
  constexpr int XLM_REGION = 576;                   // elements of an exchange buffer owned by one wave (BUF = 576 waves)
  static_assert(!XLM || G::BUF == XLM_REGION * (T / 64), "one region per wave");
  constexpr bool XL1 = (XL_OK && (SG_XLANE & 1)) || XLM;  // exchange after pass 0 on lane bits 0..2 (DPP)
  constexpr bool XL2 = (XL_OK && (SG_XLANE & 2)) || XLM;  // exchange after pass 1 on lane bits 3..5 (DPP + v_permlane*_swap)
  constexpr bool XL = XL1;                       // pass 0 runs on the transposed lane map only when exchange 1 is cross-lane
  const int tid = threadIdx.x;
  // A slot of T >= 64 lanes is whole waves: its index, and with it the run, the channel, the frame
  // counters and every base pointer, are wave-uniform and live in SGPRs (scalar branches, SGPR-base
  // addressing: global_load/store v_off, s[base:base+1]).
  constexpr bool UNI = T >= 64;
  const int slot = UNI ? __builtin_amdgcn_readfirstlane(tid / T) : tid / T;
  const int tp = tid % T;                        // lane within the frame
  // point column this thread loads and windows.  Cross-lane order: the bits pass 1 will pull into the register
  // index sit in lane bits 0..2, pass 2's in lane bits 3..5, the final pass's in the wave bits, so that after the
  // transpositions thread t holds Z[t + T r] exactly as the LDS path leaves it (tools/xlane model, tests).
  const int j0 = XL ? ((((tp & 63) & 7) << (3 + NW)) | (((tp & 63) >> 3) << NW) | (tp >> 6)) : tp;
  int jv[VW];                                    // virtual thread ids
#pragma unroll
  for (int i = 0; i < VW; ++i) jv[i] = tp + T * i;
  f2* buf0 = reinterpret_cast<f2*>(smem_raw) + slot * G::SLOT_ELEMS;
  auto lds_buf = [&](int which) -> f2* { return buf0 + (G::NBUF == 2 ? which : 0) * G::BUF; };

  // log-frequency lookup table: staged once per workgroup behind the exchange buffers
  uint2* lf_lds = reinterpret_cast<uint2*>(smem_raw + PL::LDS_BYTES);
  if constexpr (LOGF) {
    for (int i = tid; i < p.lf_cols; i += G::WG) lf_lds[i] = p.lf_tab[i];
    __syncthreads();
  }

  // XCD-aware block -> run mapping: blocks are dealt round-robin over the 8 XCDs,
  // so give each XCD a contiguous range of runs (neighbouring runs share a
  // (N - hop)-sample halo, which then hits in that XCD's L2).  Speed only.
  long long bid = blockIdx.x;
  {
    const long long nb = gridDim.x, per = nb >> 3;
    if (bid < (per << 3)) bid = (bid & 7) * per + (bid >> 3);
  }
  const long long run = bid * SLOTS + slot;
  const bool run_ok = run < p.total_runs;
#if SG_STAMP  // diagnostic build only (tools/stamps.py, tools/inkernel_clock.py): when each run starts and ends, on the
              // 100 MHz constant clock (s_memrealtime) and on the shader clock (s_memtime): their ratio is the clock the
              // chip really holds under this kernel (MI355X_MICROARCH.md, DVFS give-back item 6)
  const unsigned long long stamp_t0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long stamp_c0 = __builtin_amdgcn_s_memtime();
#endif
  const long long ch = run_ok ? run / p.runs_per_channel : 0;
  const long long f0 = run_ok ? (run % p.runs_per_channel) * (long long)p.run_len : 0;
  long long nf_ll = p.frames - f0;
  if (nf_ll > p.run_len) nf_ll = p.run_len;
  const int nf_v = run_ok ? (int)nf_ll : 0;
  const int nf = UNI ? __builtin_amdgcn_readfirstlane(nf_v) : nf_v;  // frames of this run (an SGPR for whole-wave slots)

  // ---- per-thread constants: window pairs and twiddles, held in registers for the run
  f2 win[VW][8];
#pragma unroll
  for (int i = 0; i < VW; ++i)
#pragma unroll
    for (int r = 0; r < 8; ++r) win[i][r] = p.win[(XL ? j0 : jv[i]) + TV * r];

  // inter-pass twiddles W_{8 Ns}^{k r}, k = j & (Ns - 1); identical for a lane's virtual threads
  // while Ns <= T (their ids differ by multiples of T)
  using CS = Consts<LOG2N>;
  // Poll mode transforms one frame per channel and is bound by the hop's latency, not by registers: at N = 4096 (config 4's
  // size: 512 workgroups of 256 lanes on 256 CUs) its kernels hold all seven twiddles of a pass, table-exact (DESIGN.md
  // section 6), in 226-244 VGPRs, no scratch.  The larger sizes would spill
  // (1024-lane workgroups are capped at 128 VGPRs) and keep the products.
  constexpr bool DERIVE = CS::DERIVE && !(SG_POLL12_ALL7 && ANALYSER && LOG2N == 12);
  constexpr int NTW = DERIVE ? 3 : 7;
  f2 tw8[G::P8 > 1 ? G::P8 - 1 : 1][VW][NTW];
#pragma unroll
  for (int ps = 1; ps < G::P8; ++ps) {
    const int Ns = 1 << (3 * ps);
    [[maybe_unused]] const int step = N / (8 * Ns);  // W_{8Ns}^{k r} = W_N^{k r step}
#pragma unroll
    for (int i = 0; i < VW; ++i) {
      const int k = jv[(Ns <= T) ? 0 : i] & (Ns - 1);
#pragma unroll
      for (int r = 0; r < NTW; ++r) {
        // DERIVE: (W^1, W^2, W^4) of this k come from the table of triplets the host chose so that the four products
        // formed from them land nearest the true twiddles (spectro_api.hip, derived_triplets: behind the N entries of
        // W_N^k, pass ps at offset (8^ps - 8) / 7, three entries per k) -- plain table entries otherwise
        if constexpr (DERIVE) tw8[ps - 1][i][r] = p.tw[N + 3 * ((Ns - 8) / 7 + k) + r];
        else tw8[ps - 1][i][r] = p.tw[(k * (r + 1) * step) & (N - 1)];
      }
    }
  }
  f2 twf[VW][G::REM == 2 ? 6 : 4];  // final radix-4 / radix-2 pass twiddles
#pragma unroll
  for (int i = 0; i < VW; ++i) {
    if constexpr (G::REM == 2) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 1; r < 4; ++r) twf[i][a * 3 + r - 1] = p.tw[(2 * (jv[i] + TV * a) * r) & (N - 1)];
    } else if constexpr (G::REM == 1) {
#pragma unroll
      for (int a = 0; a < 4; ++a) twf[i][a] = p.tw[2 * (jv[i] + TV * a)];
    }
  }
  f2 two = f2{2.f, 2.f};
  f2 cc = f2{0.70710678118654752440f, 0.70710678118654752440f};
  // keep the two splat constants in VGPR pairs for the run: left alone they are rematerialised from SGPRs in front of
  // every block that takes them (s_mov + v_mov_b64, four instructions a frame)
  // (Unpinning them where it would buy the fourth wave per SIMD -- the two-wave LDS-ring kernels with log-frequency rows, 130 ->
  // 128 VGPRs -- was measured in round 5: 5.2 against 5.6e8 frames/s at the reference's cadence, the rematerialisation costs more.)
  if constexpr (SG_KEEP_SPLATS) asm volatile("" : "+v"(two), "+v"(cc));
  f2 twp[VW][4];  // post-process W_N^k, k = j + TV a
#pragma unroll
  for (int i = 0; i < VW; ++i)
#pragma unroll
    for (int a = 0; a < 4; ++a) twp[i][a] = p.tw[jv[i] + TV * a];


  // ---- input addressing
  const long long frame_step = (long long)p.hop * p.sample_stride * (p.downmix > 1 ? p.downmix : 1);
  const sample_t* src = reinterpret_cast<const sample_t*>(p.pcm) + ch * p.channel_stride + f0 * frame_step;
  // logical point m (m >= PP) is first needed by frame ceil((m - PP + 1) / SV) of the step
  constexpr auto first_user = [](int m) constexpr { return m < PP ? 0 : (m - PP + (SV > 0 ? SV : 1)) / (SV > 0 ? SV : 1); };
  point_t raw[RING];
#pragma unroll
  for (int j = 0; j < RING; ++j) raw[j] = point_t{};
  if constexpr (SLIDE) {
    const point_t* s2 = reinterpret_cast<const point_t*>(src);
#pragma unroll
    for (int m = 0; m < RING; ++m)
      if (first_user(m) < nf) raw[m] = s2[j0 + T * m];  // the whole ring: frame 0 and its look-ahead
    // Land the ring before the frame loop.  Frame 0 needs it anyway, and with nothing pending on
    // the entry edge the compiler's wait at the head of every unrolled trip is the back edge's
    // counted s_waitcnt vmcnt(2 S + 1) instead of vmcnt(0) -- which also drained the previous
    // frame's stores once per trip.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt / lgkmcnt untouched
  }
  // the current frame's window start as a byte offset from the run's first sample (uniform when UNI) and this
  // lane's byte offset into the window
  gptr<const unsigned char> sp = sgpr_at<UNI>(reinterpret_cast<const unsigned char*>(src), 0);
  const gptr<const unsigned char> sp0 = sp;
  unsigned in_off = (unsigned)j0 * (unsigned)sizeof(point_t);

  // ---- LDS sample ring (LRING): any hop <= N - 4 at unit stride, every PCM sample fetched from memory once per run.
  // The slot keeps the current window in a ring of C = N + 4 floats.  The run is cut into granules of four samples
  // counted from its first sample (16-byte loads, 4-byte aligned; 8-byte for int16, converted on the way in); frame f
  // reads its window from the ring, and the granules frame f + 1 adds are requested at the top of frame f, ride in
  // KR registers rows through the butterflies and replace the samples frame f has retired.  Everything but the lane's
  // own offsets is wave-uniform.
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  typedef unsigned u3 __attribute__((ext_vector_type(3)));
  // one granule as loaded: four float32, four int16 in two dwords, or four packed int24 in three
  typedef typename std::conditional<FMT == 1, u2, typename std::conditional<FMT == 2, u3, f4>::type>::type stage_t;
  constexpr int GB = FMT == 1 ? 8 : (FMT == 2 ? 12 : 16);                 // its bytes in memory
  [[maybe_unused]] stage_t stage[KR];                                     // granules on their way: requested at the top of a frame, in the ring at its end
  [[maybe_unused]] float* const lring = reinterpret_cast<float*>(smem_raw + PL::LR_OFFSET) + slot * PL::LR_ALLOC;
  [[maybe_unused]] int lr_gloaded = 0;  // granules of the run in the ring so far
  [[maybe_unused]] int lr_gissued = 0;  // ... requested so far (>= lr_gloaded: the difference rides in `stage`)
  [[maybe_unused]] int lr_wq = 0;       // ring slot (ring position / 4) of the next granule, kept in (0, C/4]: C/4 stands for slot 0
  [[maybe_unused]] int lr_ppos = 0;     // ring position of the current window's first sample
  [[maybe_unused]] int lr_gfull = 0;    // granules of the run that lie wholly inside the caller's buffer
  [[maybe_unused]] int lr_lim = 0;      // samples readable from the run's first sample on
  [[maybe_unused]] auto ld_gran = [&](int g) -> stage_t {
    typedef stage_t stage_ua __attribute__((aligned(FMT == 1 ? 2 : (FMT == 2 ? 1 : 4))));  // sample alignment only
    return *(gptr<const stage_ua>)(sp0 + (unsigned)g * (unsigned)GB);
  };
  [[maybe_unused]] auto gran_f4 = [](stage_t v) -> f4 {
    if constexpr (FMT == 1) {
      return f4{(float)(short)(v.x & 0xffffu), (float)((int)v.x >> 16), (float)(short)(v.y & 0xffffu), (float)((int)v.y >> 16)};
    } else if constexpr (FMT == 2) {
      // twelve bytes = four little-endian 24-bit samples: byte-aligned funnel shifts (v_alignbit), then the sign
      const unsigned s1 = __builtin_amdgcn_alignbit(v.y, v.x, 24), s2 = __builtin_amdgcn_alignbit(v.z, v.y, 16);
      return f4{(float)((int)(v.x << 8) >> 8), (float)((int)(s1 << 8) >> 8), (float)((int)(s2 << 8) >> 8), (float)((int)v.z >> 8)};
    } else {
      return v;
    }
  };
  // one sample of the caller's buffer as a float (the end-of-buffer granule is fetched this way)
  [[maybe_unused]] auto ld_sample = [&](long long idx) -> float {
    if constexpr (FMT == 2) {
      const gptr<const unsigned char> b = (gptr<const unsigned char>)sp0 + 3 * idx;
      return (float)(((int)(signed char)b[2] << 16) | ((int)b[1] << 8) | (int)b[0]);
    } else {
      return (float)((gptr<const sample_t>)sp0)[idx];
    }
  };
  // granules the run holds once frame f can be transformed (frames past the run add nothing)
  [[maybe_unused]] auto lr_gneed = [&](int f) -> int {
    if (f > nf - 1) f = nf - 1;
    return (f * p.hop + N + 3) >> 2;
  };
  // request granules [from, to) of the run into one staging set: lane tp + T r takes granule from + tp + T r
  [[maybe_unused]] auto lr_request = [&](stage_t (&st)[KR], int from, int to) {
   if constexpr (LRING) {
    const int cnt_full = min(to, lr_gfull) - from;  // whole granules inside the caller's buffer
    // a row is skipped by a scalar test; inside a row the lanes past the count re-read its last granule (same
    // cache line, value never written to the ring) instead of being masked off: no exec juggling around the loads
#pragma unroll
    for (int r = 0; r < KR; ++r)
      if (cnt_full > T * r) st[r] = ld_gran(from + min(tp + T * r, cnt_full - 1));
    if (cnt_full < to - from) {
      // the caller's buffer ends inside the run's last granule (the last run of the last channel only): that
      // granule sample by sample, never touching a byte past the extent
#pragma unroll
      for (int r = 0; r < KR; ++r)
        if (tp + T * r == cnt_full) {
          const long long sq = 4LL * (from + cnt_full);
          const int left = lr_lim - 4 * (from + cnt_full);  // 1 .. 3 samples
          const float a0 = ld_sample(sq), a1 = left > 1 ? ld_sample(sq + 1) : 0.f, a2 = left > 2 ? ld_sample(sq + 2) : 0.f;
          if constexpr (FMT == 1) {
            st[r] = u2{((unsigned)(int)a0 & 0xffffu) | ((unsigned)(int)a1 << 16), (unsigned)(int)a2 & 0xffffu};
          } else if constexpr (FMT == 2) {
            const unsigned q0 = (unsigned)(int)a0 & 0xffffffu, q1 = (unsigned)(int)a1 & 0xffffffu, q2 = (unsigned)(int)a2 & 0xffffffu;
            st[r] = u3{q0 | (q1 << 24), (q1 >> 8) | (q2 << 16), q2 >> 16};
          } else {
            st[r] = f4{a0, a1, a2, 0.f};
          }
        }
    }
   }
  };
  // granule g into ring slot wq (< C / 4)
  [[maybe_unused]] auto lr_put = [&](f4 g, unsigned wq) {
    if constexpr (LR_SPLIT) {
      float* const d = lring + (wq ^ ((wq >> 3) & 4u));  // phi: index bit 5 into bit 2
      d[0] = g.x;
      d[LR_Q] = g.y;
      d[2 * LR_Q] = g.z;
      d[3 * LR_Q] = g.w;
    } else {
      *reinterpret_cast<f4*>(lring + 4 * wq) = g;
    }
  };
  // plain layout: whoever wrote ring slot 0 repeats its first sample behind the ring's end (the mirror)
  [[maybe_unused]] auto lr_mirror = [&](f4 g, unsigned wq) {
    if constexpr (!LR_SPLIT) {
      if (wq == 0) lring[LR_C] = g.x;
    }
  };
  // The current frame's window, read from the ring one frame ahead: the reads for frame f + 1 are issued in the
  // middle of frame f (behind its ring writes), so that their latency lies under the rest of that frame.
  [[maybe_unused]] f2 xw[PP];
  // hand-placed LDS reads (SG_LR_ASM): the compiler does not see the data in flight, so the consumer waits explicitly
  // (lr_window_landed) and "memory" keeps the reads behind the ring writes and the barrier in front of them
  [[maybe_unused]] auto lds_rd_b64 = [](f2& d, unsigned addr, auto off) {
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(decltype(off)::value) : "memory");
  };
  [[maybe_unused]] auto lds_rd2st64 = [](f2& d, unsigned addr, auto o0, auto o1) {
    asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(d) : "v"(addr), "n"(decltype(o0)::value), "n"(decltype(o1)::value) : "memory");
  };
  [[maybe_unused]] auto lr_window_landed = [&]() {
    if constexpr (LRING && SG_LR_ASM)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xw[0]), "+v"(xw[1]), "+v"(xw[2]), "+v"(xw[3]), "+v"(xw[4]), "+v"(xw[5]), "+v"(xw[6]), "+v"(xw[7]));
  };
  [[maybe_unused]] const unsigned lring_abs =
      (unsigned)(unsigned long long)(__attribute__((address_space(3))) const unsigned char*)reinterpret_cast<const unsigned char*>(lring);
  [[maybe_unused]] auto lr_window = [&]() {
    // complex point j0 + T m of this frame = ring samples lr_ppos + 2 (j0 + T m) and the one after it.  Row m (one
    // point per lane) lies in front of the ring's end for m < c, behind it for m > c; row c is decided per lane.
    // c is wave-uniform: one scalar branch per frame picks the copy of the reads whose addresses are three lane
    // bases plus immediates.
    const unsigned char* const rb = reinterpret_cast<const unsigned char*>(lring);
    if constexpr (!LR_SPLIT) {
      const unsigned blo = (unsigned)j0 * 8u + 4u * (unsigned)lr_ppos;
      int c = (int)((unsigned)(LR_C + 1 - lr_ppos) / (unsigned)(2 * T));
      if (c > 8) c = 8;
      const unsigned tmix = blo + 8u * T * (unsigned)c;
      const unsigned bmix = min(tmix, tmix - 4u * LR_C);  // t - 4 C wraps to a huge value while t < 4 C
      const unsigned bhi = tmix + 8u * T - 4u * LR_C;     // rows behind the wrap (exist only while c < 7)
      auto rd = [&](auto cc, auto par) {
        constexpr int CC = decltype(cc)::value;
        constexpr bool ODD = decltype(par)::value != 0;
        if constexpr (!ODD && SG_LR_ASM) {
          // eight ds_read_b64: three lane bases, byte immediates
          static_for<8>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            if constexpr (m < CC) lds_rd_b64(xw[m], lring_abs + blo, IC<8 * T * m>{});
            else if constexpr (m == CC) lds_rd_b64(xw[m], lring_abs + bmix, IC<0>{});
            else lds_rd_b64(xw[m], lring_abs + bhi, IC<8 * T * (m - CC - 1)>{});
          });
          return;
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const unsigned char* const a = m < CC ? rb + blo + 8u * T * m : (m == CC ? rb + bmix : rb + bhi + 8u * T * (m - CC - 1));
          if constexpr (ODD) {  // the window starts on an odd sample: the point straddles two 8-byte slots
            xw[m].x = *reinterpret_cast<const float*>(a);
            xw[m].y = *reinterpret_cast<const float*>(a + 4);
          } else {
            xw[m] = *reinterpret_cast<const f2*>(a);
          }
        }
        // keeps the copies apart: left alone, the optimiser sinks the eighteen sets of reads into one block fed by
        // eight per-case address registers (13 VALU per frame and ds_read2_b32 for both parities)
        asm volatile("; window reads, wrap row %0, parity %1" ::"n"(CC), "n"((int)ODD));
      };
#define SG_LR_CASES(PAR)                                                                                                    \
  switch (c) {                                                                                                               \
    case 0: rd(IC<0>{}, IC<PAR>{}); break;                                                                                   \
    case 1: rd(IC<1>{}, IC<PAR>{}); break;                                                                                   \
    case 2: rd(IC<2>{}, IC<PAR>{}); break;                                                                                   \
    case 3: rd(IC<3>{}, IC<PAR>{}); break;                                                                                   \
    case 4: rd(IC<4>{}, IC<PAR>{}); break;                                                                                   \
    case 5: rd(IC<5>{}, IC<PAR>{}); break;                                                                                   \
    case 6: rd(IC<6>{}, IC<PAR>{}); break;                                                                                   \
    case 7: rd(IC<7>{}, IC<PAR>{}); break;                                                                                   \
    default: rd(IC<8>{}, IC<PAR>{}); break;                                                                                  \
  }
      if (lr_ppos & 1) { SG_LR_CASES(1) } else { SG_LR_CASES(0) }
    } else {
      // split layout: sample s = u + 32 a + 4 b' + 256 m (+1), u = lr_ppos + 2 wave: region u & 3 (wave-uniform),
      // index (u >> 2) + e + 64 m with the lane's e = 8 a + b' = j0 >> 1
      const int wv = __builtin_amdgcn_readfirstlane(tp >> 6);
      const unsigned e = (unsigned)j0 >> 1;
      const int u0 = lr_ppos + 2 * wv;
      // the three lane bases (rows in front of the region's end, the row that straddles it, rows behind it) of the
      // samples at region R4 / (4 Q), index U + e + 64 m, for a wrap row c
      auto bases = [&](unsigned U, unsigned R4, int c, unsigned& blo, unsigned& bmix, unsigned& bhi) {
        const unsigned t = U + e;
        const unsigned ph4 = 4u * (t ^ ((t >> 3) & 4u));
        const unsigned xm = ph4 + 256u * (unsigned)c;
        bmix = R4 + min(xm, xm - 4u * LR_Q);
        blo = R4 + ph4;
        bhi = R4 + xm + 256u - 4u * LR_Q;
      };
      const unsigned U = (unsigned)u0 >> 2;
      int c = (LR_Q - (int)U) >> 6;  // rows (64 indices each) entirely in front of the region's end
      if (c > 8) c = 8;
      if ((u0 & 3) != 3) {
        // both samples of a point share the index and sit in neighbouring regions
        unsigned blo, bmix, bhi;
        bases(U, (unsigned)(u0 & 3) * (4u * LR_Q), c, blo, bmix, bhi);
        auto rd = [&](auto cc, auto) {
          constexpr int CC = decltype(cc)::value;
          if constexpr (SG_LR_ASM) {
            // one ds_read2st64_b32 per point: its two samples sit 4 Q bytes = LR_Q / 64 units of 256 bytes apart
            static_assert(LR_Q % 64 == 0 && LR_Q / 64 + 7 < 256, "st64 offsets");
            static_for<8>([&](auto mc) {
              constexpr int m = decltype(mc)::value;
              if constexpr (m < CC) lds_rd2st64(xw[m], lring_abs + blo, IC<m>{}, IC<m + LR_Q / 64>{});
              else if constexpr (m == CC) lds_rd2st64(xw[m], lring_abs + bmix, IC<0>{}, IC<LR_Q / 64>{});
              else lds_rd2st64(xw[m], lring_abs + bhi, IC<m - CC - 1>{}, IC<m - CC - 1 + LR_Q / 64>{});
            });
            return;
          }
#pragma unroll
          for (int m = 0; m < 8; ++m) {
            const unsigned char* const a = m < CC ? rb + blo + 256u * m : (m == CC ? rb + bmix : rb + bhi + 256u * (m - CC - 1));
            xw[m].x = *reinterpret_cast<const float*>(a);
            xw[m].y = *reinterpret_cast<const float*>(a + 4 * LR_Q);
          }
          asm volatile("; window reads (split ring), wrap row %0" ::"n"(CC));
        };
        SG_LR_CASES(0)
      } else {
        // the point's second sample is in region 0 of the NEXT index (one frame-wave in four when the hop is odd, never
        // when it is even): its own three bases, and the same wrap row unless the region's end falls between the two
        int cy = (LR_Q - (int)U - 1) >> 6;
        if (cy > 8) cy = 8;
        if (cy < 0) cy = 0;
        if (cy == c) {
          unsigned blo, bmix, bhi, ylo, ymix, yhi;
          bases(U, 3u * (4u * LR_Q), c, blo, bmix, bhi);
          bases(U + 1, 0u, c, ylo, ymix, yhi);
          auto rd = [&](auto cc, auto) {
            constexpr int CC = decltype(cc)::value;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
              xw[m].x = *reinterpret_cast<const float*>(m < CC ? rb + blo + 256u * m : (m == CC ? rb + bmix : rb + bhi + 256u * (m - CC - 1)));
              xw[m].y = *reinterpret_cast<const float*>(m < CC ? rb + ylo + 256u * m : (m == CC ? rb + ymix : rb + yhi + 256u * (m - CC - 1)));
            }
            asm volatile("; window reads (split ring, second sample in the next index), wrap row %0" ::"n"(CC));
          };
          SG_LR_CASES(0)
        } else {
          // ... which happens once in 64 such frames: every address worked out per lane
#pragma unroll
          for (int m = 0; m < 8; ++m) {
            unsigned t0 = U + e + 64u * m, t1 = t0 + 1;
            t0 = min(t0, t0 - (unsigned)LR_Q);
            t1 = min(t1, t1 - (unsigned)LR_Q);
            xw[m].x = lring[3 * LR_Q + (t0 ^ ((t0 >> 3) & 4u))];
            xw[m].y = lring[t1 ^ ((t1 >> 3) & 4u)];
          }
        }
      }
    }
#undef SG_LR_CASES
  };
  if constexpr (LRING) {
    const long long run_abs = ch * p.channel_stride + f0 * (long long)p.hop;  // the run's first sample, counted from pcm[0]
    long long lim_ll = p.in_extent - run_abs;
    if (lim_ll > 0x7fffffffLL) lim_ll = 0x7fffffffLL;
    lr_lim = __builtin_amdgcn_readfirstlane(run_ok && lim_ll > 0 ? (int)lim_ll : 0);
    const int g_end = nf > 0 ? ((nf - 1) * p.hop + N + 3) >> 2 : 0;            // granules the run needs
    lr_gfull = min(g_end, lr_lim >> 2);
    if (nf > 0) {
      // the first window: N / 4 = 4 T whole granules, all of them inside the run
      stage_t pre[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) pre[r] = ld_gran(tp + T * r);
#pragma unroll
      for (int r = 0; r < 4; ++r) lr_put(gran_f4(pre[r]), (unsigned)(tp + T * r));
      lr_mirror(gran_f4(pre[0]), (unsigned)tp);
      lr_gloaded = N / 4;
      lr_wq = N / 4;
      lr_gissued = N / 4;
    }
    frame_sync<G::MULTIWAVE>();
    lr_window();
  }

  const int bins = p.bins;
  unsigned char* out_base = reinterpret_cast<unsigned char*>(p.out);
  const long long out_row0 = ch * p.out_channel_stride + f0 * (long long)bins;
  // running output pointers (float kinds): bin tp and bin M - tp of the current row
  constexpr int ESZ = OUT == OUT_DB_U8 ? 1 : 4;                      // bytes per output element
  gptr<unsigned char> op = sgpr_at<UNI>(out_base, out_row0 * (long long)ESZ);  // the current output row (uniform when UNI)
  const long long row_bytes = (long long)bins * (long long)ESZ;
  // lane byte offsets into the row: bin tp, bin M - tp, and where the first mirror store goes -- virtual thread 0
  // has no mirror bin for k = 0, so that store carries bin M/2 instead (or the Nyquist bin M when that is requested)
  unsigned lo_off = (unsigned)tp * (unsigned)ESZ, hi_off = (unsigned)(M - tp) * (unsigned)ESZ;
  unsigned hi0_off = (tp == 0 && !p.nyquist) ? (unsigned)(M / 2) * (unsigned)ESZ : hi_off;

  // fused batch smoothing: the AnalyserNode recurrence (3D/visualizer.js:351 sets 0.75) on the lane's own bins; the
  // launcher makes the run the whole channel, so the state starts at zero exactly as the serial scan does
  [[maybe_unused]] float sm_k[VW][4], sm_m[VW][4], sm_mid = 0.f;
#pragma unroll
  for (int i = 0; i < VW; ++i)
#pragma unroll
    for (int a = 0; a < 4; ++a) sm_k[i][a] = sm_m[i][a] = 0.f;
  [[maybe_unused]] const float sm_gain = 1.0f - p.an_tau;
  [[maybe_unused]] auto smooth = [&](float& st, float pw) -> float {  // power in, smoothed magnitude squared out
    float sm = SG_FMA(p.an_tau, st, sm_gain * __builtin_amdgcn_sqrtf(pw));
    if (!(__builtin_fabsf(sm) < __builtin_inff())) sm = 0.f;  // NaN / inf resets the smoother
    st = sm;
    return sm * sm;
  };
  auto finish = [&](float pw) -> float {
    if constexpr (OUT == OUT_POWER) return pw;
    else if constexpr (OUT == OUT_MAG) return __builtin_amdgcn_sqrtf(pw);
    else {
      const float l2 = __builtin_amdgcn_logf(pw);  // v_log_f32 = log2
      if constexpr (OUT == OUT_DB) return fmaxf(l2 * p.db_mul, p.floor_db);
      else {
        // 255 (dB - min_db) / (max_db - min_db) as one FMA of log2 |X|^2, clamped to [max(dB floor in byte
        // units, 0), 255] by one v_med3; the floor() of the byte map is the truncation of the float -> integer
        // conversion at the store (the value is non-negative by then), so neither costs an instruction
        return __builtin_amdgcn_fmed3f(SG_FMA(l2, p.byte_a, p.byte_b), p.byte_floor, 255.f);
      }
    }
  };

  // thresholds of the remaining-work priority (quarters of the run length; SGPRs)
  const int prio_t1 = (p.run_len + 3) >> 2, prio_t2 = 2 * prio_t1, prio_t3 = 3 * prio_t1;
  // ... as a countdown: the level the run starts at, and the frame at which it next drops (-1: never)
  [[maybe_unused]] int prio_lvl = 0, prio_next = -1;
  if constexpr (SG_PRIO_REM && UNI && !LRING) {
    if (p.prio_rem) {
      prio_lvl = nf > prio_t3 ? 3 : (nf > prio_t2 ? 2 : (nf > prio_t1 ? 1 : 0));
      if (prio_lvl == 3) __builtin_amdgcn_s_setprio(3);
      else if (prio_lvl == 2) __builtin_amdgcn_s_setprio(2);
      else if (prio_lvl == 1) __builtin_amdgcn_s_setprio(1);
      prio_next = prio_lvl == 3 ? nf - prio_t3 : (prio_lvl == 2 ? nf - prio_t2 : (prio_lvl == 1 ? nf - prio_t1 : -1));
    }
  }
#if SG_STALL
  constexpr bool STALL_K = LOG2N == 11 && SHIFT_FULL == 2 && OUT_FULL == OUT_POWER;
  unsigned long long stall_acc = 0, stall_t0 = 0;
  const unsigned long long stall_run0 = __builtin_amdgcn_s_memtime();
#endif
  // stamps around one wait (SG_STALL == K picks which): s_memtime comes back through lgkmcnt, so each stamp drains it first --
  // the opening one so that nothing older is billed to the interval, the closing one because that drain IS the wait being
  // measured for the LDS probes
  [[maybe_unused]] auto stall_a = [&](auto pc) {
#if SG_STALL
    if constexpr (STALL_K && decltype(pc)::value == SG_STALL) {
      unsigned long long t0;
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) : : "memory");
      stall_t0 = t0;
    }
#endif
  };
  [[maybe_unused]] auto stall_b = [&](auto pc) {
#if SG_STALL
    if constexpr (STALL_K && decltype(pc)::value == SG_STALL) {
      unsigned long long t1;
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : : "memory");
      stall_acc += t1 - stall_t0;
    }
#endif
  };
  // One frame.  U = position inside the unrolled rotation period (compile time).
  // Returns false when this wave has nothing left to do.
  auto step = [&](auto uc, const int fi) -> bool {
    constexpr int U = decltype(uc)::value;
    constexpr int ROT = (U * SV) % RING;
    // (whole-wave slots only ever step through frames of their run -- the loop below stops at nf)
    const bool live = UNI ? true : fi < nf;
    if constexpr (!G::MULTIWAVE && !UNI) {
      // frames never span waves here, so a wave whose slots are all done can leave
      if (!__any(live)) return false;
    }

    if constexpr (SG_PRIO_REM && UNI && !LRING) {
      // Issue priority by remaining work.  The SIMD's arbiter serves the oldest ready wave first, so the four
      // workgroups sharing a CU drain one after the other (identical 16-frame runs take 18 ... 41 us) and a launch
      // ends with a long tail of CUs running one wave per SIMD.  A wave with more of its run ahead of it outranks
      // one with less: the waves of a CU advance together and finish together.  By countdown: one scalar compare per
      // frame; the level drops by one at each threshold -- prio_t1 < prio_t2 < prio_t3 -- so three frames of a run do
      // more than compare.
      if (__builtin_expect(fi == prio_next, 0)) {
        prio_lvl -= 1;
        if (prio_lvl == 2) __builtin_amdgcn_s_setprio(2);
        else if (prio_lvl == 1) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
        prio_next = prio_lvl == 2 ? nf - prio_t2 : (prio_lvl == 1 ? nf - prio_t1 : -1);
      }
    }
    f2 v[VW][8];
#if SG_STALL
    stall_a(IC<9>{});
    stall_a(IC<10>{});
    stall_b(IC<10>{});
    stall_a(IC<1>{});
    if constexpr (STALL_K && SG_STALL == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // the frame's two loads: all but the previous frame's 8 stores
    stall_b(IC<1>{});
#endif
    if constexpr (LRING) {
      // request what frame fi + 1 adds to the run
      {
        const int to = lr_gneed(fi + 1);
        lr_request(stage, lr_gissued, to);
        lr_gissued = to;
      }
      lr_window_landed();
#pragma unroll
      for (int i = 0; i < VW; ++i) dft8_stage1_win<PP, 0, VW>(v[i], xw, win[i], 0, i);
    } else if constexpr (SLIDE) {
      // window multiply fused into the first butterfly stage
#pragma unroll
      for (int i = 0; i < VW; ++i) {
        dft8_stage1_win<RING, FMT, VW>(v[i], raw, win[i], ROT, i);
      }
      // Request logical points RING .. RING + SV - 1 (frame fi + DEPTH) straight into the slots this
      // frame has just finished with (its logical 0 .. SV-1).
      if constexpr (!(SG_ABLATE & 8)) {
        if constexpr (UNI) {
          // every look-ahead point of a frame has the same first user (SV points per hop), so one uniform test
          // covers the group: past the end of the run the loads are aimed at the run's first points instead
          // (valid memory, values never used) -- no branch around them, no merge of old and new ring values,
          // and the counted vmcnt waits hold on every path
          static_assert(first_user(RING) == first_user(RING + SV - 1), "one test per hop");
          const gptr<const unsigned char> lbs = (fi + first_user(RING) < nf) ? sp + T * RING * sizeof(point_t) : sp0;
          const unsigned io = opaque_off(in_off);
#pragma unroll
          for (int j = 0; j < SV; ++j) raw[(j + ROT) % RING] = ld_pcm((gptr<const point_t>)(lbs + io) + T * j);
        } else {
#pragma unroll
          for (int j = 0; j < SV; ++j)
            if (fi + first_user(RING + j) < nf)
              raw[(j + ROT) % RING] = ld_pcm((gptr<const point_t>)(sp + in_off) + T * (RING + j));
        }
        sp += SV * T * sizeof(point_t);  // hop == 2 S TV samples: one frame is S TV = SV T points further on
      }
    } else {
      f2 x[PP];  // the generic path converts while loading, so its window input is always float
      bool pend_done = false;
      if constexpr (PEND && FMT == 0) {
        {
          pend_done = true;
          typedef f2 f2_a4 __attribute__((aligned(4)));
          typedef int int_a2 __attribute__((aligned(2)));
          float* const hrow = const_cast<float*>(reinterpret_cast<const float*>(src));  // this channel's window in the history
          // the history row is a double-written ring of H samples (aux_kernels.hip): a new sample goes to its place in
          // the window and to the other copy, H floats below or above it.  Every address below is the row's start (a
          // uniform pointer, pinned in SGPRs) plus a 32-bit byte offset: position q's other copy is (q + H) mod 2 H.
          const gptr<unsigned char> hrow0 = sgpr_base<UNI>(reinterpret_cast<unsigned char*>(hrow - p.pend_wstart));
          const unsigned h4 = 4u * (unsigned)p.pend_h, h8 = 2u * h4;
          const unsigned qb = 4u * (unsigned)p.pend_wstart + 8u * (unsigned)j0;  // byte offset of this lane's point m = 0 in the row
          const unsigned char* const prow =
              reinterpret_cast<const unsigned char*>(p.pend_src) + ch * p.pend_stride * (p.pend_s16 ? 2 : 4);
          // phase 1: every load of the window is issued (new samples from the caller's block -- page-locked memory is a
          // PCIe round trip away -- older ones from the history)
#pragma unroll
          for (int m = 0; m < PP; ++m) {
            const int s2 = 2 * (j0 + T * m);
            f2 xv = f2{0.f, 0.f};
            if (live) {
              if (s2 >= p.pend_split) {
                const long long q = s2 - p.pend_split;
                if (p.pend_s16) {
                  const int r = *reinterpret_cast<const int_a2*>(prow + 2 * q);
                  xv = f2{(float)(short)(r & 0xffff), (float)(r >> 16)};
                } else {
                  xv = *reinterpret_cast<const f2_a4*>(prow + 4 * q);
                }
              } else {
                xv = *(gptr<const f2_a4>)(hrow0 + (qb + 8u * (unsigned)(T * m)));
              }
            }
            x[m] = xv;
          }
          // phase 2: the new samples go to their place in the history and to the ring's other copy, point by point (the
          // scheduling barrier keeps one point's addresses live at a time: left to cluster, the sixteen mirror stores cost
          // the poll kernel of N = 4096 thirty registers and its second wave per SIMD)
#pragma unroll
          for (int m = 0; m < PP; ++m) {
            const int s2 = 2 * (j0 + T * m);
            if (live && s2 >= p.pend_split) {
              const unsigned q0 = qb + 8u * (unsigned)(T * m);
              *(gptr<f2_a4>)(hrow0 + q0) = x[m];
              const unsigned t0 = q0 + h4, t1 = t0 + 4u;
              *(gptr<float>)(hrow0 + min(t0, t0 - h8)) = x[m].x;   // (t - 2 H wraps to a huge value while t < 2 H)
              *(gptr<float>)(hrow0 + min(t1, t1 - h8)) = x[m].y;
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      if (pend_done) {
      } else if (live && FMT == 0 && (ANALYSER || GEN_UNIT || (p.sample_stride == 1 && p.downmix <= 1))) {  // (poll mode reads its own mono history: always unit stride)
        // unit stride, any hop (the reference's cadence is ~735 samples between polls): complex points are
        // adjacent float pairs, one 8-byte load each (4-byte aligned: odd hops put every other frame on an odd
        // sample), addressed as SGPR frame base + lane offset + immediate
        if constexpr (FMT == 0) {
          typedef f2 f2_a4 __attribute__((aligned(4)));
          const gptr<const unsigned char> sb =
              sgpr_base<UNI>(reinterpret_cast<const unsigned char*>(src + (long long)fi * frame_step));
#pragma unroll
          for (int m = 0; m < PP; ++m) x[m] = *(gptr<const f2_a4>)(sb + in_off + (unsigned)(T * m * sizeof(f2)));
        }
      } else if (!((ANALYSER || GEN_UNIT) && FMT == 0) && live) {
        const sample_t* s = src + (long long)fi * frame_step;
#pragma unroll
        for (int m = 0; m < PP; ++m) {
          const long long n2 = 2LL * (j0 + T * m);
          f2 xv;
          if (p.downmix > 1 && p.downmix_w) {  // weighted mono down-mix (uniform branches; the weights are scalar loads)
            float a0 = 0.f, a1 = 0.f;
            for (int c = 0; c < p.downmix; ++c) {
              const float w = p.downmix_w[c];
              a0 = SG_FMA(w, (float)s[(n2 * p.downmix + c) * p.sample_stride], a0);
              a1 = SG_FMA(w, (float)s[((n2 + 1) * p.downmix + c) * p.sample_stride], a1);
            }
            xv = f2{a0, a1};
          } else if (p.downmix > 1) {  // mono down-mix of K interleaved samples: the mean
            float a0 = 0.f, a1 = 0.f;
            for (int c = 0; c < p.downmix; ++c) {
              a0 += (float)s[(n2 * p.downmix + c) * p.sample_stride];
              a1 += (float)s[((n2 + 1) * p.downmix + c) * p.sample_stride];
            }
            xv = f2{a0 * p.downmix_scale, a1 * p.downmix_scale};
          } else {
            xv = f2{(float)s[n2 * p.sample_stride], (float)s[(n2 + 1) * p.sample_stride]};
          }
          x[m] = xv;
        }
      } else {
#pragma unroll
        for (int m = 0; m < PP; ++m) x[m] = f2{0.f, 0.f};
      }
#pragma unroll
      for (int i = 0; i < VW; ++i) {
        dft8_stage1_win<PP, 0, VW>(v[i], x, win[i], 0, i);
      }
    }

    // ---- Stockham passes
    int which = 0;  // ping-pong buffer index (compile-time after unrolling)
#pragma unroll
    for (int ps = 0; ps < G::P8; ++ps) {
      const int Ns = 1 << (3 * ps);
      if constexpr (!(SG_ABLATE & 4)) {
#pragma unroll
        for (int i = 0; i < VW; ++i) {
          if (ps > 0) {
            const f2* tl = tw8[ps > 0 ? ps - 1 : 0][i];
            if constexpr (DERIVE) {
              // W^3, W^5, W^6, W^7 as products of the stored triplet (W^1, W^2, W^4), which the host chose for them
              const f2* b = tl;
              f2 w[7];
              w[0] = b[0]; w[1] = b[1]; w[3] = b[2];
              w[2] = pk_cmul(b[0], b[1]);
              w[4] = pk_cmul(b[0], b[2]);
              w[5] = pk_cmul(b[1], b[2]);
              w[6] = pk_cmul(w[2], b[2]);
              dft8_tw_pass_blk(v[i], w, two, cc);
            } else {
              dft8_tw_pass_blk(v[i], tl, two, cc);
            }
          } else {
            dft8_stage23_blk(v[i], cc);
          }
        }
      }
      const bool last = (ps == G::P8 - 1) && (G::REM == 0);
      if (XLM && SG_XLM_EX1_LDS && ps == 0) {
        if constexpr (XLM) {
          // The same transposition as xl_swap3<0> (REG bits <-> lane bits 0..2: an 8 x 8 transpose inside every
          // group of 8 lanes), through the wave's private 576-element part of the buffer instead of 48 DPP selects:
          // 4 ds_write2_b64 + 4 ds_read2_b64 at immediate offsets, rows padded 8 -> 9 (conflict-free both ways),
          // no barrier -- a wave's DS operations execute in order.
          f2* b = lds_buf(which) + (tp >> 6) * XLM_REGION;
          const int L = tp & 63;
          // (Round 6 measured the reads returning in butterfly order with the next pass split at its waits: -0.6 %,
          // profiles/r06_n2048_stalls.txt section C.)
          f2* bw = b + (L >> 3) * 72 + (L & 7);
          stall_a(IC<2>{});
#pragma unroll
          for (int q = 0; q < 8; ++q) bw[9 * q] = v[0][q];
          frame_sync<false>();
          const f2* br = b + (L >> 3) * 72 + (L & 7) * 9;
#pragma unroll
          for (int r = 0; r < 8; ++r) v[0][r] = br[r];
          stall_b(IC<2>{});
        }
      } else if ((XL1 && ps == 0) || (XL2 && ps == 1)) {
        if constexpr (XL1 || XL2) {
          if (!(SG_ABLATE & 2)) {
            if (ps == 0) xl_swap3<0>(v[0]);
            else xl_swap3<3>(v[0]);
          }
        }
      } else if (XLM && ps == 2) {
        if constexpr (XLM) {
          // the NW low bits of the register index trade places with the wave bits: element (wave w, lane l, reg q)
          // goes to (wave q & (2^NW - 1), lane l, reg w << (3 - NW) | q >> NW).  Rows of T lanes: conflict-free.
          // A wave writes only its own XLM_REGION (512 elements used of 576: the part exchange 1 also uses, so a
          // wave that runs ahead never writes where a slower one is still transposing); the reads cross regions.
          f2* b = lds_buf(which);
          const int wv = tp >> 6;
          f2* bw = b + wv * XLM_REGION + (tp & 63);
          stall_a(IC<3>{});
#pragma unroll
          for (int q = 0; q < 8; ++q) bw[(q >> NW) * T + (q & ((1 << NW) - 1)) * 64] = v[0][q];
          stall_b(IC<3>{});
          stall_a(IC<4>{});
          frame_sync<true>();
          stall_b(IC<4>{});
          const f2* br = b + tp;
          stall_a(IC<5>{});
#pragma unroll
          for (int r = 0; r < 8; ++r) v[0][r] = br[(r >> (3 - NW)) * XLM_REGION + (r & ((8 >> NW) - 1)) * T];
          stall_b(IC<5>{});
          if constexpr (TWO_SYNC) __syncthreads();
          which ^= 1;
        }
      } else if (!last && !(SG_ABLATE & 2)) {
        f2* b = lds_buf(which);
#pragma unroll
        for (int i = 0; i < VW; ++i) {
          const int k = jv[i] & (Ns - 1);
          const int o = ((jv[i] - k) << 3) + k;   // a multiple of 8 plus k < Ns
          if constexpr (SG_PAD16 && TV % 16 == 0) {
            if (Ns == 1) {
              // First exchange: a lane writes 8 consecutive elements, a 16-lane group (what one LDS cycle of a
              // 64-bit access serves, 32 banks) must land on 32 different banks: one spare element per SIXTEEN
              // does it for the writes (8 j + (j >> 1)) and keeps the reads of an aligned 16-lane group contiguous,
              // where one per eight leaves every read group one 2-way conflict.
              f2* bw16 = b + (o + (jv[i] >> 1));
#pragma unroll
              for (int q = 0; q < 8; ++q) bw16[q] = v[i][q];
              continue;
            }
          }
          if constexpr (SG_SWIZZLE && TV % 16 == 0) {
            if (Ns >= 8) {
              // Later exchanges: element i lives at i ^ 8 when bit 6 of i is set, no padding.  A 16-lane group of a
              // write (two runs of 8 elements 64 apart for Ns = 8, one run of 16 otherwise) and of a read (16
              // consecutive elements) then covers every bank once.  The flip depends on the lane and on the parity
              // of q only, so it is two base registers and the usual immediates:
              //   Ns = 8:   i = 64 m + k + 8 q, bit 6 = m & 1: even q sit 8 higher, odd q 8 lower, for odd m
              //   Ns = 64:  i = 512 m' + k + 64 q, bit 6 = q & 1: odd q use k ^ 8
              //   Ns >= 512: bit 6 of i is bit 6 of k for every q
              const int flip8 = Ns == 8 ? ((jv[i] >> 3) & 1) * 8 : 0;
              f2* be = b + (Ns == 8 ? o + flip8 : (Ns == 64 ? o : (o ^ (((o >> 6) & 1) << 3))));
              f2* bo = b + (Ns == 8 ? o - flip8 : (Ns == 64 ? (o ^ 8) : (o ^ (((o >> 6) & 1) << 3))));
#pragma unroll
              for (int q = 0; q < 8; ++q) ((q & 1) ? bo : be)[q * Ns] = v[i][q];
              continue;
            }
          }
          f2* bw = b + pad(o);
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            if constexpr (TV % 8 == 0) bw[Ns >= 8 ? padd(q * Ns) : q] = v[i][q];
            else b[pad(o + q * Ns)] = v[i][q];
          }
        }
        frame_sync<G::MULTIWAVE>();
#pragma unroll
        for (int i = 0; i < VW; ++i) {
          if constexpr (SG_PAD16 && TV % 16 == 0) {
            if (Ns == 1) {
              const f2* br16 = b + (jv[i] + (jv[i] >> 4));
#pragma unroll
              for (int r = 0; r < 8; ++r) v[i][r] = br16[TV * r + TV * r / 16];
              continue;
            }
          }
          if constexpr (SG_SWIZZLE && TV % 16 == 0) {
            if (Ns >= 8) {
              if constexpr (TV % 128 == 0) {
                const f2* brs = b + (jv[i] ^ (((jv[i] >> 6) & 1) << 3));  // TV r is a multiple of 128: bit 6 is the lane's
#pragma unroll
                for (int r = 0; r < 8; ++r) v[i][r] = brs[TV * r];
              } else {
                const f2* br0 = b + jv[i];        // j < TV <= 64: bit 6 of j + TV r is a compile-time bit of r
                const f2* br1 = b + (jv[i] ^ 8);
#pragma unroll
                for (int r = 0; r < 8; ++r) v[i][r] = ((((TV * r) >> 6) & 1) ? br1 : br0)[TV * r];
              }
              continue;
            }
          }
          const f2* br = b + pad(jv[i]);
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            if constexpr (TV % 8 == 0) v[i][r] = br[padd(TV * r)];
            else v[i][r] = b[pad(jv[i] + TV * r)];
          }
        }
        if constexpr (TWO_SYNC) __syncthreads();
        which ^= 1;
      }
    }
#pragma unroll
    for (int i = 0; i < VW; ++i) {
      const f2* fl = twf[i];
      if constexpr (G::REM == 2) {  // two radix-4 butterflies: (v[a], v[a+2], v[a+4], v[a+6])
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          v[i][a + 2] = pk_cmul(v[i][a + 2], fl[a * 3 + 0]);
          v[i][a + 4] = pk_cmul(v[i][a + 4], fl[a * 3 + 1]);
          v[i][a + 6] = pk_cmul(v[i][a + 6], fl[a * 3 + 2]);
          dft4(v[i][a], v[i][a + 2], v[i][a + 4], v[i][a + 6]);
        }
      } else if constexpr (G::REM == 1) {  // four radix-2 butterflies: (v[a], v[a+4])
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const f2 s = pk_cfma(v[i][a + 4], fl[a], v[i][a]);
          v[i][a + 4] = pk_fms(v[i][a], two, s);
          v[i][a] = s;
        }
      }
    }
    // now v[i][r] = Z[j_i + TV r] (half-scaled), natural order.

    // LRING: the staged granules replace the samples this frame has retired, then the next frame's window is read.
    // One-wave frames do both at the END of the frame (SG_LR_LATE): the loads requested at its top get the whole frame to
    // land instead of the 70 % up to the partner exchange.  Larger frames need a barrier on either side of the ring
    // writes and keep them between the last pass's barrier and the partner exchange's.
    [[maybe_unused]] auto lr_advance = [&]() {
      // The staged granules replace the samples this frame has retired.  Every wave of the frame is past its window
      // reads (one-wave frames: a wave's LDS operations execute in order; larger frames: at least one exchange
      // barrier lies behind), and the barrier of the partner exchange below orders these writes in front of the next
      // frame's reads.
      // They are the ones frame fi + 1 adds, requested at the top of this frame.
      const int lr_ge = lr_gneed(fi + 1);
      const int lr_cnt = lr_ge - lr_gloaded;  // <= KR T (the launcher picks KR by the hop)
#pragma unroll
      for (int r = 0; r < KR; ++r)
        if (tp + T * r < lr_cnt) {
          const unsigned w = (unsigned)lr_wq + (unsigned)(tp + T * r);
          lr_put(gran_f4(stage[r]), min(w, w - (unsigned)LR_Q));
        }
      if (!LR_SPLIT && lr_wq + lr_cnt > LR_Q) {  // scalar: these rows write slot 0 (once per trip around the ring)
#pragma unroll
        for (int r = 0; r < KR; ++r)
          if (tp + T * r < lr_cnt) {
            const unsigned w = (unsigned)lr_wq + (unsigned)(tp + T * r);
            lr_mirror(gran_f4(stage[r]), min(w, w - (unsigned)LR_Q));
          }
      }
      lr_gloaded = lr_ge;
      lr_wq += lr_cnt;
      if (lr_wq > LR_Q) lr_wq -= LR_Q;
      lr_ppos += p.hop;
      if (lr_ppos >= LR_C) lr_ppos -= LR_C;
    };
    constexpr bool LR_LATE = LRING && !G::MULTIWAVE && SG_LR_LATE;
    if constexpr (LRING && !LR_LATE) lr_advance();

    // ---- partner exchange: publish the upper half, fetch Z[M - k] for k = j + TV a
    f2 zp[VW][4];
    if constexpr (SG_ABLATE & 2) {
#pragma unroll
      for (int i = 0; i < VW; ++i)
#pragma unroll
        for (int a = 0; a < 4; ++a) zp[i][a] = v[i][7 - a];
    } else {
      f2* b = lds_buf(which);
#pragma unroll
      for (int i = 0; i < VW; ++i) {
        // no padding here: lanes write and read consecutive elements (forwards, then backwards), so every lane
        // group of an access covers one contiguous span of banks
        f2* bw = b + jv[i];
        stall_a(IC<6>{});
#pragma unroll
        for (int r = 4; r < 8; ++r) bw[TV * (r - 4)] = v[i][r];
        stall_b(IC<6>{});
      }
      stall_a(IC<7>{});
      frame_sync<G::MULTIWAVE && !(SG_ABLATE & 16)>();  // bit 4 (timing only, wrong results): no barrier at the partner exchange
      stall_b(IC<7>{});
#pragma unroll
      for (int i = 0; i < VW; ++i) {
        const f2* br = b + (TV - jv[i]);  // a = 3; the others sit TV (3 - a) above it: (M - k) - 4 TV, == 4 TV only for k == 0
        stall_a(IC<8>{});
#pragma unroll
        for (int a = 0; a < 4; ++a) zp[i][a] = br[TV * (3 - a)];
        stall_b(IC<8>{});
        if (i == 0 && tp == 0) zp[0][0] = v[0][0];  // k = 0 pairs with itself: yields DC and Nyquist
      }
      if constexpr (LRING && !LR_LATE) lr_window();  // the next frame's window, behind this frame's ring writes (and their barrier)
      if constexpr (TWO_SYNC) __syncthreads();
      which ^= 1;
    }
    if constexpr (!G::MULTIWAVE && LOGF) frame_sync<false>();

    // ---- real-input post-process: with E = Z_k + conj Z_{M-k}, D = Z_k - conj Z_{M-k},
    // X_k = E - i W D and conj X_{M-k} = E + i W D = 2E - X_k.
    float pk[VW][4], pm[VW][4];
    float pmid = 4.f * SG_FMA(v[0][4].x, v[0][4].x, v[0][4].y * v[0][4].y);  // bin M/2 (tp == 0 only)
#pragma unroll
    for (int i = 0; i < VW; ++i) {
      real_post_blk(v[i], zp[i], twp[i], two);  // v[i][a] = X_k, zp[i][a] = conj X_{M-k}
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        pk[i][a] = SG_FMA(v[i][a].x, v[i][a].x, v[i][a].y * v[i][a].y);
        pm[i][a] = SG_FMA(zp[i][a].x, zp[i][a].x, zp[i][a].y * zp[i][a].y);
      }
    }
    if constexpr (SMOOTH) {
#pragma unroll
      for (int i = 0; i < VW; ++i)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          pk[i][a] = smooth(sm_k[i][a], pk[i][a]);
          pm[i][a] = smooth(sm_m[i][a], pm[i][a]);
        }
      pmid = smooth(sm_mid, pmid);
    }

    if constexpr (LOGF) {
      // whole row of values into LDS, then every lane interpolates its columns; byte rows hold whole bytes
      // (finish() leaves the floor to the integer conversion, which never happens on this path)
      // Poll mode (ANALYSER): the texel is the AnalyserNode value of the bin -- smoothing state read-modify-written,
      // 20 log10, then the byte (floored, like the Uint8Array the shader samples: sonogram-vertex.shader:51-52 over
      // 3D/visualizer.js:399-409) or the float dB clamped at floor_db (a -inf texel would poison its neighbours' blend).
      [[maybe_unused]] float* const an_st = ANALYSER ? p.an_state + ch * (long long)M : nullptr;
      auto texel = [&](int k, float pw) -> float {
        if constexpr (ANALYSER) {
          float sm = SG_FMA(p.an_tau, an_st[k], (1.0f - p.an_tau) * __builtin_amdgcn_sqrtf(pw));
          if (!(__builtin_fabsf(sm) < __builtin_inff())) sm = 0.f;  // NaN / inf resets the smoother
          if (live) an_st[k] = sm;
          const float y = 6.02059991327962f * __builtin_amdgcn_logf(sm);  // 20 log10 = 20 log10(2) * log2
          if constexpr (OUT == OUT_DB_U8) return __builtin_floorf(__builtin_amdgcn_fmed3f((y - p.an_min_db) * p.an_byte_mul, 0.f, 255.f));
          else return fmaxf(y, p.floor_db);
        } else {
          if constexpr (OUT == OUT_DB_U8) return __builtin_floorf(finish(pw));
          else return finish(pw);
        }
      };
      float* lrow = reinterpret_cast<float*>(lds_buf(which));
#pragma unroll
      for (int i = 0; i < VW; ++i)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int k = jv[i] + TV * a;
          lrow[k] = texel(k, pk[i][a]);
          if (k != 0) lrow[M - k] = texel(M - k, pm[i][a]);
        }
      if (tp == 0) lrow[M / 2] = texel(M / 2, pmid);
      frame_sync<G::MULTIWAVE>();
      // poll mode: one frame per channel, rows `out_channel_stride` apart, the caller's array may be shorter than the
      // row (an_len), the history-ring row always takes all columns
      unsigned char* o = ANALYSER ? out_base + ch * p.out_channel_stride * (long long)ESZ
                                  : out_base + (out_row0 + (long long)fi * bins) * (long long)ESZ;
      [[maybe_unused]] unsigned char* const rg = (ANALYSER && p.an_ring) ? p.an_ring + ch * p.an_ring_stride : nullptr;
      [[maybe_unused]] const int o_len = ANALYSER ? p.an_len : p.lf_cols;
      // Fast path (round 4): the row's columns are dealt CPL = 2 or 4 CONSECUTIVE columns to a lane (256 columns: N = 1024
      // and 2048), so that a lane's table entries are one 16- / 32-byte LDS read, a column's two texels one ds_read2_b32
      // (the table keeps i1 = i0 + 1 at the clamped ends too, with a blend fraction of 0 or 1), and its columns leave as ONE
      // store per lane -- a dword of four bytes, or 16 bytes of floats -- instead of one per column.
      constexpr int CPL_FAST = T <= 64 ? 4 : (T == 128 ? 2 : 0);   // one fast shape per size: 256 columns at N = 1024 / 2048
      static_assert(!LFFAST || CPL_FAST > 0, "the consecutive-column path exists for frames of up to two waves");
      const bool lf_fast = LFFAST || (CPL_FAST > 0 && p.lf_cols == CPL_FAST * T && o_len == p.lf_cols);
      auto fast_cols = [&](auto cplc) {
        constexpr int CPL = decltype(cplc)::value;
        const unsigned char* lrb = reinterpret_cast<const unsigned char*>(lrow);
        float val[CPL];
        // two columns at a time (one 16-byte table read, two texel pairs): the scheduling barrier between the halves keeps
        // the live set of this block at four registers of table entries, not eight (the kernel sits at the 128-VGPR edge)
#pragma unroll
        for (int h = 0; h < CPL; h += 2) {
          uint2 e[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) e[u] = lf_lds[CPL * tp + h + u];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const float* t = reinterpret_cast<const float*>(lrb + (e[u].x & 0xffffu));
            const float f = __uint_as_float(e[u].y);
            val[h + u] = SG_FMA(f, t[1], SG_FMA(-f, t[0], t[0]));   // a (1 - f) + b f: exact at f = 0 and f = 1
          }
          if (h + 2 < CPL) asm volatile("" : "+v"(val[h]), "+v"(val[h + 1]));
        }
        if constexpr (OUT == OUT_DB_U8) {
          // bytes: the GL_LINEAR blend of two byte texels, rounded to the nearest byte; CPL of them packed into one store
          unsigned pack = 0;
#pragma unroll
          for (int u = 0; u < CPL; ++u) pack |= (unsigned)(unsigned char)(val[u] + 0.5f) << (8 * u);
          if constexpr (CPL == 4) {
            typedef unsigned u32_a1 __attribute__((aligned(1)));
            if constexpr (ANALYSER) {
              if (rg) *reinterpret_cast<u32_a1*>(rg + 4 * tp) = pack;
            }
            *reinterpret_cast<u32_a1*>(o + 4 * tp) = pack;
          } else {
            typedef unsigned short u16_a1 __attribute__((aligned(1)));
            if constexpr (ANALYSER) {
              if (rg) *reinterpret_cast<u16_a1*>(rg + 2 * tp) = (unsigned short)pack;
            }
            *reinterpret_cast<u16_a1*>(o + 2 * tp) = (unsigned short)pack;
          }
        } else {
          typedef float fv __attribute__((ext_vector_type(CPL)));
          typedef fv fv_a4 __attribute__((aligned(4)));
          fv v;
#pragma unroll
          for (int u = 0; u < CPL; ++u) v[u] = val[u];
          __builtin_nontemporal_store(v, reinterpret_cast<fv_a4*>(o) + tp);
        }
      };
      if (live && lf_fast) {
        if constexpr (CPL_FAST > 0) fast_cols(IC<CPL_FAST>{});
      } else if (!LFFAST && live) {
        // four columns per lane per trip: all table reads, then all row reads, then the stores, so the
        // LDS latencies overlap; the table holds byte offsets into the row
        const unsigned char* lrb = reinterpret_cast<const unsigned char*>(lrow);
        for (int c0 = tp; c0 < p.lf_cols; c0 += 4 * T) {
          uint2 e[4];
          float a[4], b[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) e[u] = lf_lds[min(c0 + u * T, p.lf_cols - 1)];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            a[u] = *reinterpret_cast<const float*>(lrb + (e[u].x & 0xffffu));
            b[u] = *reinterpret_cast<const float*>(lrb + (e[u].x >> 16));
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (c0 + u * T < p.lf_cols) {
              const float fq = __uint_as_float(e[u].y);
              const float val = SG_FMA(fq, b[u], SG_FMA(-fq, a[u], a[u]));   // a (1 - f) + b f, as in the fast path
              const int col = c0 + u * T;
              // bytes: the GL_LINEAR blend of two byte texels, rounded to the nearest byte
              if constexpr (OUT == OUT_DB_U8) {
                const unsigned char bq = (unsigned char)(val + 0.5f);
                if constexpr (ANALYSER) {
                  if (rg) rg[col] = bq;
                }
                if (col < o_len) o[col] = bq;
              } else {
                if (col < o_len) st_out(reinterpret_cast<float*>(o) + col, val);
              }
            }
        }
      }
      if constexpr (TWO_SYNC) __syncthreads();
      which ^= 1;
      if constexpr (!G::MULTIWAVE) frame_sync<false>();
    } else if constexpr (ANALYSER) {
      // getByteFrequencyData / getFloatFrequencyData (3D/visualizer.js:352,358): X^[k] = tau X^[k] + (1 - tau) |X[k]|
      // on the handle's state, 20 log10 X^, then the float dB or the byte; the byte also goes to the
      // history-ring row (the reference's texSubImage2D row, 3D/visualizer.js:399-409).  One frame per channel.
      if (live) {
        float* const st = p.an_state + ch * (long long)M;
        unsigned char* const ob = out_base + ch * p.out_channel_stride * (long long)ESZ;
        unsigned char* const rg = p.an_ring ? p.an_ring + ch * p.an_ring_stride : nullptr;
        auto emit = [&](int k, float pw) {
          float sm = SG_FMA(p.an_tau, st[k], (1.0f - p.an_tau) * __builtin_amdgcn_sqrtf(pw));
          if (!(__builtin_fabsf(sm) < __builtin_inff())) sm = 0.f;  // NaN / inf resets the smoother
          st[k] = sm;
          const float y = 6.02059991327962f * __builtin_amdgcn_logf(sm);  // 20 log10 = 20 log10(2) * log2
          const unsigned char bq = (unsigned char)__builtin_amdgcn_fmed3f((y - p.an_min_db) * p.an_byte_mul, 0.f, 255.f);
          if (rg) rg[k] = bq;
          if (k < p.an_len) {
            if constexpr (OUT == OUT_DB_U8) ob[k] = bq;
            else reinterpret_cast<float*>(ob)[k] = y;
          }
        };
#pragma unroll
        for (int i = 0; i < VW; ++i)
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            const int k = jv[i] + TV * a;
            emit(k, pk[i][a]);
            if (k != 0) emit(M - k, pm[i][a]);
          }
        if (tp == 0) emit(M / 2, pmid);
      }
    } else if constexpr (OUT != OUT_DB_U8) {
      if constexpr (SG_ABLATE & 1) {
#pragma unroll
        for (int i = 0; i < VW; ++i)
#pragma unroll
          for (int a = 0; a < 4; ++a) { keep(finish(pk[i][a])); keep(finish(pm[i][a])); }
        keep(finish(pmid));
        if (p.nyquist == 12345) *(gptr<float>)(op + lo_off) = pk[0][0];
      } else if (T == 64 || live) {  // T == 64: one frame per wave, dead waves already left the loop
        const gptr<unsigned char> orow_s = op;
        const unsigned lo_o = UNI ? opaque_off(lo_off) : lo_off, hi_o = UNI ? opaque_off(hi_off) : hi_off;
        const unsigned hi0_o = UNI ? opaque_off(hi0_off) : hi0_off;
#pragma unroll
        for (int i = 0; i < VW; ++i)
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            st_out((gptr<float>)(orow_s + lo_o) + (T * i + TV * a), finish(pk[i][a]));
            if (i == 0 && a == 0) {
              const float vm = (tp == 0 && !p.nyquist) ? pmid : pm[0][0];
              st_out((gptr<float>)(orow_s + hi0_o), finish(vm));
            } else {
              st_out((gptr<float>)(orow_s + hi_o) - (T * i + TV * a), finish(pm[i][a]));
            }
          }
        if (__builtin_expect(p.nyquist != 0, 0)) {  // a kernel argument: a scalar branch, not an exec mask set up and restored every frame
          asm volatile("" ::: "memory");
          if (tp == 0) *((gptr<float>)(orow_s + lo_o) + M / 2) = finish(pmid);
        }
      }
      op += row_bytes;
    } else {
      // bytes straight from registers: 8 wave-stores of 64 contiguous bytes per virtual thread
      // (measured 6 % faster than staging the row in LDS for 8-byte stores)
      if (T == 64 || live) {
        const gptr<unsigned char> olo = op + (UNI ? opaque_off(lo_off) : lo_off);  // bin tp
        const gptr<unsigned char> ohi = op + (UNI ? opaque_off(hi_off) : hi_off);  // bin M - tp
        const gptr<unsigned char> ohi0 = op + (UNI ? opaque_off(hi0_off) : hi0_off);
#pragma unroll
        for (int i = 0; i < VW; ++i)
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            olo[T * i + TV * a] = (unsigned char)finish(pk[i][a]);
            if (i == 0 && a == 0) {
              const float vm = (tp == 0 && !p.nyquist) ? pmid : pm[0][0];
              ohi0[0] = (unsigned char)finish(vm);
            } else {
              ohi[-(T * i + TV * a)] = (unsigned char)finish(pm[i][a]);
            }
          }
        if (__builtin_expect(p.nyquist != 0, 0)) {
          asm volatile("" ::: "memory");
          if (tp == 0) olo[M / 2] = (unsigned char)finish(pmid);
        }
      }
      op += row_bytes;
    }

    if constexpr (LR_LATE) {
      lr_advance();
      lr_window();
    }
    stall_b(IC<9>{});

    // Ping-pong (NBUF == 2): `which` restarts at 0 each frame.  That is safe when the number of
    // exchanges per frame is even (the buffer about to be rewritten was last read two barriers
    // ago); with an odd count one extra barrier separates the frames.
    if constexpr (G::MULTIWAVE && G::NBUF == 2) {
      constexpr int n_ex = (XLM ? 1 : (G::REM == 0 ? G::P8 - 1 : G::P8)) + 1 + (LOGF ? 1 : 0);  // LDS exchanges per frame
      if constexpr ((n_ex & 1) != 0 || n_ex < 2) __syncthreads();
    }
    return true;
  };

  auto guarded = [&](auto uc, const int fi) -> bool {  // false: stop
    constexpr int U = decltype(uc)::value;
    if constexpr (U >= PERIOD) return true;
    else return fi + U < p.run_len && step(uc, fi + U);
  };
  static_assert(PERIOD <= 6, "extend the unrolled driver");
  // Whole-wave slots (UNI): nf is wave-uniform and every frame of the unrolled period sits behind a forward scalar branch;
  // the loop's only exit is its latch.  (With exits in the middle of the period -- `break` or `return` alike -- the
  // compiler funnels them through one latch / exit block whose counters are undefined on the early paths and
  // materialises them: two v_readfirstlane of an undefined register and an s_mov_b64 per frame.)
  if constexpr (UNI) {
    static_assert(!G::MULTIWAVE || SLOTS == 1, "a multi-wave frame is a workgroup of its own: every wave sees the same nf");
    const int lim = nf;  // frames this slot walks (<= p.run_len)
    auto fwd = [&](auto uc, const int fi) {
      constexpr int U = decltype(uc)::value;
      if constexpr (U < PERIOD) {
        if (fi + U < lim) (void)step(uc, fi + U);
      }
    };
    for (int fi = 0; fi < lim; fi += PERIOD) {
      fwd(IC<0>{}, fi);
      fwd(IC<1>{}, fi);
      fwd(IC<2>{}, fi);
      fwd(IC<3>{}, fi);
      fwd(IC<4>{}, fi);
      fwd(IC<5>{}, fi);
    }
  } else {
    for (int fi = 0; fi < p.run_len; fi += PERIOD) {
      if (!guarded(IC<0>{}, fi) || !guarded(IC<1>{}, fi) || !guarded(IC<2>{}, fi) || !guarded(IC<3>{}, fi) ||
          !guarded(IC<4>{}, fi) || !guarded(IC<5>{}, fi))
        break;
    }
  }
#if SG_STALL
  if constexpr (STALL_K) {
    if (run_ok && (tp & 63) == 0 && p.lf_tab) {  // per wave: cycles of the run, cycles between the chosen stamps
      __builtin_amdgcn_s_waitcnt(0);
      unsigned long long* st = reinterpret_cast<unsigned long long*>(const_cast<uint2*>(p.lf_tab)) + 4 * run + 2 * (tp >> 6);
      st[0] = __builtin_amdgcn_s_memtime() - stall_run0;
      st[1] = stall_acc;
    }
  }
#endif
#if SG_STAMP
  if constexpr (!LOGF) {
    if (run_ok && tp == 0 && p.lf_tab) {
      __builtin_amdgcn_s_waitcnt(0);  // the run's stores have left the wave
      unsigned long long* st = reinterpret_cast<unsigned long long*>(const_cast<uint2*>(p.lf_tab)) + 4 * run;
      const unsigned long long c1 = __builtin_amdgcn_s_memtime();
      st[0] = stamp_t0;
      st[1] = __builtin_amdgcn_s_memrealtime();
      st[2] = stamp_c0;
      st[3] = c1;
    }
  }
#endif
}

}  // namespace sg
