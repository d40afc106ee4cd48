// stft_kernel.hpp -- the STFT hot path as one HIP kernel template for gfx950.
//
// What it replaces: the platform code behind analyser.getByteFrequencyData()
// (reference call site 3D/visualizer.js:358; configuration UI/player.js:10-11),
// in the form BASELINE.json's north_star prescribes: window -> hop-overlapped
// framing -> FFT -> |X|^2 / log-magnitude per hop.
//
// Shape of the computation (DESIGN.md section 3 has the derivation):
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

#include "pk_math.hpp"

namespace sg {

enum : int { OUT_POWER = 0, OUT_MAG = 1, OUT_DB = 2, OUT_DB_U8 = 3 };

struct StftParams {
  const float* pcm;
  void* out;
  const f2* tw;   // tw[k] = (cos, -sin)(2 pi k / N), k in [0, N)
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
  float min_db;
  float byte_mul;  // 255 / (max_db - min_db)
};

#define SG_FMA(a, b, c) __builtin_fmaf((a), (b), (c))

// ---- 8-point DFT, forward sign, natural order in and out, split in two so that the
// first butterfly stage can absorb either the window or the inter-pass twiddles.  All
// complex arithmetic is packed (pk_math.hpp): 12 + 18 instructions for the first pass,
// 18 + 18 for a twiddled pass.

// stage 1 with the (real) window fused: a_r = x_r w_r + x_{r+4} w_{r+4}, a_{r+4} = x_r w_r - x_{r+4} w_{r+4}.
// `rot` is the physical rotation of the sliding register window (compile-time after inlining).
template <int RING>
__device__ __forceinline__ void dft8_stage1_win(f2 (&v)[8], const f2 (&x)[RING], const f2 (&w)[8], int rot) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const f2 xa = x[(r + rot) % RING], xb = x[(r + 4 + rot) % RING];
    const f2 t = xa * w[r];
    const f2 pb = xb * w[r + 4];  // contracted into the two FMAs below
    v[r] = t + pb;
    v[r + 4] = t - pb;
  }
}

// stage 1 with the inter-pass twiddles fused (w[r-1] = W^{k r}, r = 1..7; W^0 = 1):
//   a_r = w_r v_r + w_{r+4} v_{r+4},  a_{r+4} = 2 w_r v_r - a_r.
__device__ __forceinline__ void dft8_stage1_tw(f2 (&v)[8], const f2* w, const f2 two) {
  {
    const f2 a = pk_cfma(v[4], w[3], v[0]);
    v[4] = pk_fms(v[0], two, a);
    v[0] = a;
  }
#pragma unroll
  for (int r = 1; r < 4; ++r) {
    const f2 tt = pk_cmul(v[r], w[r - 1]);
    const f2 a = pk_cfma(v[r + 4], w[r + 3], tt);
    v[r + 4] = pk_fms(tt, two, a);
    v[r] = a;
  }
}

// stages 2 and 3 on the stage-1 output.  Rotations by -i/+i ride on op_sel/neg modifiers,
// and the two 1/sqrt(2) rotations are folded into the last stage's FMAs (cc.x = 1/sqrt 2).
__device__ __forceinline__ void dft8_stage23(f2 (&v)[8], const f2 cc) {
  const f2 a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3], a4 = v[4], a5 = v[5], a6 = v[6], a7 = v[7];
  const f2 b0 = a0 + a2, b2 = a0 - a2;
  const f2 b1 = a1 + a3, d13 = a1 - a3;
  const f2 b4 = pk_add_mi(a4, a6), b6 = pk_add_pi(a4, a6);
  const f2 s5 = pk_add_mi(a5, a5);  // a5 W8^1 sqrt2    = (x + y, y - x)
  const f2 q7 = pk_add_pi(a7, a7);  // -(a7 W8^3 sqrt2) = (x - y, x + y)
  const f2 u = s5 - q7;             // sqrt2 (a5' + a7')
  const f2 g = s5 + q7;             // sqrt2 (a5' - a7')
  v[0] = b0 + b1;
  v[4] = b0 - b1;
  v[2] = pk_add_mi(b2, d13);
  v[6] = pk_add_pi(b2, d13);
  v[1] = pk_fma_x(u, cc, b4);
  v[5] = pk_fnma_x(u, cc, b4);
  v[3] = pk_fma_mi_x(g, cc, b6);
  v[7] = pk_fma_pi_x(g, cc, b6);
}

__device__ __forceinline__ void dft4(f2& x0, f2& x1, f2& x2, f2& x3) {
  const f2 a0 = x0 + x2, a2 = x0 - x2;
  const f2 a1 = x1 + x3, d = x1 - x3;
  x0 = a0 + a1;
  x2 = a0 - a1;
  x1 = pk_add_mi(a2, d);
  x3 = pk_add_pi(a2, d);
}

// LDS index padding: one spare element per 8 keeps the radix-8 exchange writes
// (stride-8 element bursts per lane) off each other's banks.
__device__ __forceinline__ int pad(int i) { return i + (i >> 3); }

template <int LOG2N>
struct Geo {
  static constexpr int N = 1 << LOG2N;
  static constexpr int M = N / 2;
  static constexpr int LOG2M = LOG2N - 1;
  static constexpr int T = M / 8;                  // threads per frame
  static constexpr int WG = T > 256 ? T : 256;     // workgroup size
  static constexpr int SLOTS = WG / T;             // frames in flight per workgroup
  static constexpr bool MULTIWAVE = T > 64;        // a frame spans several waves
  static constexpr int P8 = LOG2M / 3;             // radix-8 passes
  static constexpr int REM = LOG2M % 3;            // final radix 2^REM pass if != 0
  static constexpr int BUF = M + M / 8;            // padded elements per exchange buffer
  static constexpr int NBUF = MULTIWAVE ? 2 : 1;   // ping-pong only when barriers are needed
  static constexpr int LDS_BYTES = SLOTS * NBUF * BUF * 8;
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

// Diagnostic ablations (tools/tune.py only; never defined in the product build):
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
#define SG_PREFETCH_DEPTH 2  // frames of cover for the PCM loads (HBM latency under load > 1 frame)
#endif

// SHIFT: 0 = generic loads (any hop, any stride); S in {1,2,4} = sliding window,
// hop == S * 2T, unit stride, 8-byte aligned rows.
template <int LOG2N, int SHIFT, int OUT>
__global__ __launch_bounds__(Geo<LOG2N>::WG, SG_MIN_WAVES) void stft_kernel(const StftParams p) {
  using G = Geo<LOG2N>;
  constexpr int N = G::N, M = G::M, T = G::T, SLOTS = G::SLOTS;
  // The sliding window lives in a ring of RING = 8 + S (DEPTH - 1) register pairs: the S new
  // points of frame f + DEPTH are requested while frame f is computed, into the slots frame f
  // has just finished with.  The frame loop is unrolled over the rotation period RING / S.
  constexpr int DEPTH = (SHIFT == 2 || SHIFT == 4) ? SG_PREFETCH_DEPTH : 1;
  constexpr int RING = SHIFT > 0 ? 8 + SHIFT * (DEPTH - 1) : 8;
  constexpr int PERIOD = (SHIFT == 2 || SHIFT == 4) ? RING / SHIFT : 1;
  static_assert(SHIFT == 0 || RING % (SHIFT ? SHIFT : 1) == 0, "ring must be a whole number of hops");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

  const int tid = threadIdx.x;
  const int slot = tid / T;
  const int t = tid % T;
  f2* buf0 = reinterpret_cast<f2*>(smem_raw) + slot * (G::NBUF * G::BUF);

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
  const long long ch = run_ok ? run / p.runs_per_channel : 0;
  const long long f0 = run_ok ? (run % p.runs_per_channel) * (long long)p.run_len : 0;
  long long nf_ll = p.frames - f0;
  if (nf_ll > p.run_len) nf_ll = p.run_len;
  const int nf = run_ok ? (int)nf_ll : 0;

#ifdef SG_STAGGER  // experiment: de-phase the waves of a CU (diagnostic)
  {
    const int wid = (int)(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int d = __builtin_amdgcn_readfirstlane((wid * 7) % SG_STAGGER);
    for (int i = 0; i < d; ++i) __builtin_amdgcn_s_sleep(4);
  }
#endif
  // ---- per-thread constants: window pairs and twiddles, held in registers for the run
  f2 win[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) win[j] = p.win[t + T * j];

  f2 tw8[G::P8 > 1 ? G::P8 - 1 : 1][7];
#pragma unroll
  for (int ps = 1; ps < G::P8; ++ps) {
    const int Ns = 1 << (3 * ps);
    const int k = t & (Ns - 1);
    const int step = N / (8 * Ns);  // W_{8Ns}^{k r} = W_N^{k r step}
#pragma unroll
    for (int r = 1; r < 8; ++r) tw8[ps - 1][r - 1] = p.tw[(k * r * step) & (N - 1)];
  }
  f2 twf[G::REM == 2 ? 6 : 4];  // final radix-4 / radix-2 pass twiddles
  if constexpr (G::REM == 2) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 1; r < 4; ++r) twf[i * 3 + r - 1] = p.tw[(2 * (t + T * i) * r) & (N - 1)];
  } else if constexpr (G::REM == 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) twf[i] = p.tw[2 * (t + T * i)];
  }
  const f2 two = f2{2.f, 2.f};
  const f2 cc = f2{0.70710678118654752440f, 0.70710678118654752440f};
  f2 twp[4];  // post-process W_N^k, k = t + T i
#pragma unroll
  for (int i = 0; i < 4; ++i) twp[i] = p.tw[t + T * i];

  // ---- input addressing
  const float* src = p.pcm + ch * p.channel_stride + f0 * (long long)p.hop * p.sample_stride;
  const long long frame_step = (long long)p.hop * p.sample_stride;
  f2 raw[RING];  // logical point j of the frame at unrolled step u sits in slot (j + u*SHIFT) % RING
#pragma unroll
  for (int j = 0; j < RING; ++j) raw[j] = f2{0.f, 0.f};
  if constexpr (SHIFT > 0) {
    if (nf > 0) {
      const f2* s2 = reinterpret_cast<const f2*>(src);
#pragma unroll
      for (int j = 0; j < 8; ++j) raw[j] = s2[t + T * j];
      // new points of frames 1 .. DEPTH-1 (slot 8 + S (d-1) + j = point 8 - S + j of frame d)
#pragma unroll
      for (int d = 1; d < DEPTH; ++d) {
        if (d < nf) {
          const f2* sd = reinterpret_cast<const f2*>(src + (long long)d * frame_step);
#pragma unroll
          for (int j = 0; j < SHIFT; ++j) raw[8 + SHIFT * (d - 1) + j] = sd[t + T * (8 - SHIFT + j)];
        }
      }
    }
  }

  const int bins = p.bins;
  unsigned char* out_base = reinterpret_cast<unsigned char*>(p.out);
  const long long out_row0 = ch * p.out_channel_stride + f0 * (long long)bins;

  auto finish = [&](float pw) -> float {
    if constexpr (OUT == OUT_POWER) return pw;
    else if constexpr (OUT == OUT_MAG) return __builtin_amdgcn_sqrtf(pw);
    else {
      float y = __builtin_amdgcn_logf(pw) * p.db_mul;  // v_log_f32 = log2
      y = fmaxf(y, p.floor_db);
      if constexpr (OUT == OUT_DB) return y;
      else {
        const float bv = __builtin_floorf((y - p.min_db) * p.byte_mul);
        return fminf(fmaxf(bv, 0.f), 255.f);
      }
    }
  };

  // One frame.  U = position inside the unrolled rotation period (compile time).
  // Returns false when this wave has nothing left to do.
  auto frame = [&](auto uc, const int fi) -> bool {
    constexpr int U = decltype(uc)::value;
    constexpr int ROT = (U * SHIFT) % RING;
    const bool live = fi < nf;
    if constexpr (!G::MULTIWAVE) {
      // frames never span waves here, so a wave whose slots are all done can leave
      if (!__any(live)) return false;
    }

    f2 v[8];
    if constexpr (SHIFT > 0) {
      // window multiply fused into the first butterfly stage
      dft8_stage1_win<RING>(v, raw, win, ROT);
      if constexpr (PERIOD == 1) {  // SHIFT == 1: no unrolled rotation, slide by moving
#pragma unroll
        for (int j = 0; j < 8 - SHIFT; ++j) raw[j] = raw[j + SHIFT];
      }
      // Request the S new points of frame fi + DEPTH straight into the slots this frame has
      // just finished with (its logical 0..S-1 = logical 8-S..7 of frame fi + DEPTH).
      if (!(SG_ABLATE & 8) && fi + DEPTH < nf) {
        const f2* s2 = reinterpret_cast<const f2*>(src + (long long)(fi + DEPTH) * frame_step);
#pragma unroll
        for (int j = 0; j < SHIFT; ++j) {
          const int slot_j = PERIOD == 1 ? 8 - SHIFT + j : (j + ROT) % RING;
          raw[slot_j] = s2[t + T * (8 - SHIFT + j)];
        }
      }
    } else {
      f2 x[8];
      if (live) {
        const float* s = src + (long long)fi * frame_step;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const long long n2 = 2LL * (t + T * j);
          x[j] = f2{s[n2 * p.sample_stride], s[(n2 + 1) * p.sample_stride]};
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = f2{0.f, 0.f};
      }
      dft8_stage1_win<8>(v, x, win, 0);
    }

    // ---- Stockham passes
    int which = 0;  // ping-pong buffer index (compile-time after unrolling)
#pragma unroll
    for (int ps = 0; ps < G::P8; ++ps) {
      const int Ns = 1 << (3 * ps);
      const int k = t & (Ns - 1);
      if constexpr (!(SG_ABLATE & 4)) {
        if (ps > 0) dft8_stage1_tw(v, tw8[ps > 0 ? ps - 1 : 0], two);
        dft8_stage23(v, cc);
      }
      const bool last = (ps == G::P8 - 1) && (G::REM == 0);
      if (!last && !(SG_ABLATE & 2)) {
        f2* b = buf0 + (G::NBUF == 2 ? which * G::BUF : 0);
        const int o = ((t - k) << 3) + k;
#pragma unroll
        for (int q = 0; q < 8; ++q) b[pad(o + q * Ns)] = v[q];
        frame_sync<G::MULTIWAVE>();
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = b[pad(t + T * r)];
        which ^= 1;
      }
    }
    if constexpr (G::REM == 2) {  // two radix-4 butterflies: (v[i], v[i+2], v[i+4], v[i+6])
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        v[i + 2] = pk_cmul(v[i + 2], twf[i * 3 + 0]);
        v[i + 4] = pk_cmul(v[i + 4], twf[i * 3 + 1]);
        v[i + 6] = pk_cmul(v[i + 6], twf[i * 3 + 2]);
        dft4(v[i], v[i + 2], v[i + 4], v[i + 6]);
      }
    } else if constexpr (G::REM == 1) {  // four radix-2 butterflies: (v[i], v[i+4])
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f2 a = pk_cfma(v[i + 4], twf[i], v[i]);
        v[i + 4] = pk_fms(v[i], two, a);
        v[i] = a;
      }
    }
    // now v[r] = Z[t + T r] (half-scaled), natural order.

    // ---- partner exchange: publish the upper half, fetch Z[M - k] for k = t + T i
    f2 zp[4];
    if constexpr (SG_ABLATE & 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) zp[i] = v[7 - i];
    } else {
      f2* b = buf0 + (G::NBUF == 2 ? which * G::BUF : 0);
#pragma unroll
      for (int r = 4; r < 8; ++r) b[pad(t + T * (r - 4))] = v[r];
      frame_sync<G::MULTIWAVE>();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = 4 * T - t - T * i;  // (M - k) - 4T; == 4T only for k == 0
        zp[i] = b[pad(idx)];
      }
      if (t == 0) zp[0] = v[0];  // k = 0 pairs with itself: yields DC and Nyquist
      which ^= 1;
    }
    if constexpr (!G::MULTIWAVE && OUT == OUT_DB_U8) frame_sync<false>();

    // ---- real-input post-process: with E = Z_k + conj Z_{M-k}, D = Z_k - conj Z_{M-k},
    // X_k = E - i W D and conj X_{M-k} = E + i W D = 2E - X_k.  14 scalar ops per bin pair.
    float pk[4], pm[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f2 e = pk_add_cj(v[i], zp[i]);
      const f2 d = pk_sub_cj(v[i], zp[i]);
      const f2 a = pk_fma_y(d, twp[i], pk_fma_mi_x(d, twp[i], e));  // X_k = E - i W D
      const f2 bq = pk_fms(e, two, a);                              // conj X_{M-k} = 2E - X_k
      pk[i] = SG_FMA(a.x, a.x, a.y * a.y);
      pm[i] = SG_FMA(bq.x, bq.x, bq.y * bq.y);
    }
    const float pmid = 4.f * SG_FMA(v[4].x, v[4].x, v[4].y * v[4].y);  // bin M/2, thread 0 only

    const long long row = out_row0 + (long long)fi * bins;
    if constexpr (OUT != OUT_DB_U8) {
      float* o = reinterpret_cast<float*>(out_base) + row;
      if constexpr (SG_ABLATE & 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { keep(finish(pk[i])); keep(finish(pm[i])); }
        keep(finish(pmid));
        if (p.nyquist == 12345) o[t] = pk[0];
      } else if (T == 64 || live) {  // T == 64: one slot per wave, dead waves already left the loop
        // Branch-free stores: the (t,i) = (0,0) lane has no mirror bin, so its second store
        // carries bin M/2 instead (or the Nyquist bin when that is requested).
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int k = t + T * i;
          o[k] = finish(pk[i]);
          if (i == 0) {
            const bool self = (t == 0);
            const int km = self ? (p.nyquist ? M : M / 2) : M - k;
            const float vm = (self && !p.nyquist) ? pmid : pm[0];
            o[km] = finish(vm);
          } else {
            o[M - k] = finish(pm[i]);
          }
        }
        if (p.nyquist && t == 0) o[M / 2] = finish(pmid);
      }
    } else {
      // bytes go through LDS so that each thread stores 8 consecutive bins
      unsigned char* lb = reinterpret_cast<unsigned char*>(buf0 + (G::NBUF == 2 ? which * G::BUF : 0));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = t + T * i;
        lb[k] = (unsigned char)finish(pk[i]);
        if (k != 0) lb[M - k] = (unsigned char)finish(pm[i]);
        else lb[M] = (unsigned char)finish(pm[i]);
      }
      if (t == 0) lb[M / 2] = (unsigned char)finish(pmid);
      frame_sync<G::MULTIWAVE>();
      unsigned char* o = out_base + row;
      if (live) {
        if ((bins & 7) == 0 && ((reinterpret_cast<uintptr_t>(o) & 7) == 0)) {
          *reinterpret_cast<uint2*>(o + 8 * t) = *reinterpret_cast<const uint2*>(lb + 8 * t);
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) o[t + T * i] = lb[t + T * i];
          if (p.nyquist && t == 0) o[M] = lb[M];
        }
      }
      which ^= 1;
      if constexpr (!G::MULTIWAVE) frame_sync<false>();
    }

    // MULTIWAVE: `which` restarts at 0 each frame.  That is safe when the number of
    // exchanges per frame is even (the buffer about to be rewritten was last read two
    // barriers ago); with an odd count one extra barrier separates the frames.
    if constexpr (G::MULTIWAVE) {
      constexpr int n_ex = (G::REM == 0 ? G::P8 - 1 : G::P8) + 1 + (OUT == OUT_DB_U8 ? 1 : 0);
      if constexpr ((n_ex & 1) != 0 || n_ex < 2) __syncthreads();
    }
    return true;
  };

  auto step = [&](auto uc, const int fi) -> bool {  // false: stop
    constexpr int U = decltype(uc)::value;
    if constexpr (U >= PERIOD) return true;
    else return fi + U < p.run_len && frame(uc, fi + U);
  };
  static_assert(PERIOD <= 6, "extend the unrolled driver");
  for (int fi = 0; fi < p.run_len; fi += PERIOD) {
    if (!step(IC<0>{}, fi) || !step(IC<1>{}, fi) || !step(IC<2>{}, fi) || !step(IC<3>{}, fi) ||
        !step(IC<4>{}, fi) || !step(IC<5>{}, fi))
      break;
  }
}

}  // namespace sg
