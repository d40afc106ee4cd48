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
//     (register-resident sliding window: no overlap re-reads at all).
//   * The M-point FFT is a Stockham autosort: radix-8 passes done in registers,
//     each followed by an exchange through a padded LDS buffer; every pass reads
//     in[t + T*r], so reads are always lane-contiguous.  After the last pass
//     thread t holds Z[t + T*r] in natural order.
//   * Real-input post-process pairs bin k = t + T*i with M-k: one fetch of the
//     partner through LDS gives both X[k] and X[M-k].
//   * Epilogue fused: |X|^2, |X|, dB or the AnalyserNode byte, stored coalesced.
// Not a dense contraction: no MFMA.  The kernel is HBM-bound by design; LDS is
// used only for the transposes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sg {

typedef float f2 __attribute__((ext_vector_type(2)));

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

__device__ __forceinline__ f2 cmul(f2 a, f2 b) {
  return f2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
__device__ __forceinline__ f2 mul_mi(f2 a) { return f2{a.y, -a.x}; }  // a * (-i)

// 8-point DFT, forward sign, natural order in and out.
__device__ __forceinline__ void dft8(f2 (&v)[8]) {
  const float c = 0.70710678118654752440f;
  f2 a0 = v[0] + v[4], a4 = v[0] - v[4];
  f2 a1 = v[1] + v[5], a5 = v[1] - v[5];
  f2 a2 = v[2] + v[6], a6 = v[2] - v[6];
  f2 a3 = v[3] + v[7], a7 = v[3] - v[7];
  a5 = f2{(a5.x + a5.y) * c, (a5.y - a5.x) * c};   // * W8^1
  a6 = mul_mi(a6);                                 // * W8^2
  a7 = f2{(a7.y - a7.x) * c, -(a7.x + a7.y) * c};  // * W8^3
  f2 b0 = a0 + a2, b2 = a0 - a2;
  f2 b1 = a1 + a3, b3 = mul_mi(a1 - a3);
  f2 b4 = a4 + a6, b6 = a4 - a6;
  f2 b5 = a5 + a7, b7 = mul_mi(a5 - a7);
  v[0] = b0 + b1; v[4] = b0 - b1;
  v[2] = b2 + b3; v[6] = b2 - b3;
  v[1] = b4 + b5; v[5] = b4 - b5;
  v[3] = b6 + b7; v[7] = b6 - b7;
}

__device__ __forceinline__ void dft4(f2& x0, f2& x1, f2& x2, f2& x3) {
  f2 a0 = x0 + x2, a2 = x0 - x2;
  f2 a1 = x1 + x3, a3 = mul_mi(x1 - x3);
  x0 = a0 + a1; x2 = a0 - a1;
  x1 = a2 + a3; x3 = a2 - a3;
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

// SHIFT: 0 = generic loads (any hop, any stride); S in {1,2,4} = sliding window,
// hop == S * 2T, unit stride, 8-byte aligned rows.
template <int LOG2N, int SHIFT, int OUT>
__global__ __launch_bounds__(Geo<LOG2N>::WG) void stft_kernel(const StftParams p) {
  using G = Geo<LOG2N>;
  constexpr int N = G::N, M = G::M, T = G::T, SLOTS = G::SLOTS;
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
  f2 twp[4];  // post-process W_N^k, k = t + T i
#pragma unroll
  for (int i = 0; i < 4; ++i) twp[i] = p.tw[t + T * i];

  // ---- input addressing
  const float* src = p.pcm + ch * p.channel_stride + f0 * (long long)p.hop * p.sample_stride;
  const long long frame_step = (long long)p.hop * p.sample_stride;
  f2 raw[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) raw[j] = f2{0.f, 0.f};
  if constexpr (SHIFT > 0) {
    if (nf > 0) {
      const f2* s2 = reinterpret_cast<const f2*>(src);
#pragma unroll
      for (int j = 0; j < 8; ++j) raw[j] = s2[t + T * j];
    }
  }

  const int bins = p.bins;
  unsigned char* out_base = reinterpret_cast<unsigned char*>(p.out);
  const long long out_row0 = ch * p.out_channel_stride + f0 * (long long)bins;

  for (int fi = 0; fi < p.run_len; ++fi) {
    const bool live = fi < nf;
    if constexpr (!G::MULTIWAVE) {
      // frames never span waves here, so a wave whose slots are all done can leave
      if (!__any(live)) break;
    }

    f2 v[8];
    f2 nxt[SHIFT > 0 ? SHIFT : 1];
    if constexpr (SHIFT > 0) {
      // prefetch the next frame's S new points; consumed at the end of this iteration
      if (fi + 1 < nf) {
        const f2* s2 = reinterpret_cast<const f2*>(src + (long long)(fi + 1) * frame_step);
#pragma unroll
        for (int j = 0; j < SHIFT; ++j) nxt[j] = s2[t + T * (8 - SHIFT + j)];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = raw[j] * win[j];
    } else {
      if (live) {
        const float* s = src + (long long)fi * frame_step;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const long long n2 = 2LL * (t + T * j);
          v[j] = f2{s[n2 * p.sample_stride], s[(n2 + 1) * p.sample_stride]} * win[j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = f2{0.f, 0.f};
      }
    }

    // ---- Stockham passes
    int which = 0;  // ping-pong buffer index (compile-time after unrolling)
#pragma unroll
    for (int ps = 0; ps < G::P8; ++ps) {
      const int Ns = 1 << (3 * ps);
      const int k = t & (Ns - 1);
      if (ps > 0) {
#pragma unroll
        for (int r = 1; r < 8; ++r) v[r] = cmul(v[r], tw8[ps > 0 ? ps - 1 : 0][r - 1]);
      }
      dft8(v);
      const bool last = (ps == G::P8 - 1) && (G::REM == 0);
      if (!last) {
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
        v[i + 2] = cmul(v[i + 2], twf[i * 3 + 0]);
        v[i + 4] = cmul(v[i + 4], twf[i * 3 + 1]);
        v[i + 6] = cmul(v[i + 6], twf[i * 3 + 2]);
        dft4(v[i], v[i + 2], v[i + 4], v[i + 6]);
      }
    } else if constexpr (G::REM == 1) {  // four radix-2 butterflies: (v[i], v[i+4])
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f2 b = cmul(v[i + 4], twf[i]);
        v[i + 4] = v[i] - b;
        v[i] = v[i] + b;
      }
    }
    // now v[r] = Z[t + T r] (half-scaled), natural order.

    // ---- partner exchange: publish the upper half, fetch Z[M - k] for k = t + T i
    f2 zp[4];
    {
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

    // ---- real-input post-process + epilogue
    float pk[4], pm[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f2 zk = v[i];
      const f2 e = f2{zk.x + zp[i].x, zk.y - zp[i].y};
      const f2 d = f2{zk.x - zp[i].x, zk.y + zp[i].y};
      const f2 tt = cmul(twp[i], d);
      const f2 a = f2{e.x + tt.y, e.y - tt.x};  // X[k]
      const f2 bb = f2{e.x - tt.y, e.y + tt.x}; // conj X[M-k]
      pk[i] = a.x * a.x + a.y * a.y;
      pm[i] = bb.x * bb.x + bb.y * bb.y;
    }
    float pmid = 4.f * (v[4].x * v[4].x + v[4].y * v[4].y);  // bin M/2, thread 0 only

    auto finish = [&](float pw) -> float {
      if constexpr (OUT == OUT_POWER) return pw;
      else if constexpr (OUT == OUT_MAG) return __builtin_amdgcn_sqrtf(pw);
      else {
        float y = __builtin_amdgcn_logf(pw) * p.db_mul;  // v_log_f32 = log2
        y = fmaxf(y, p.floor_db);
        if constexpr (OUT == OUT_DB) return y;
        else {
          float bv = __builtin_floorf((y - p.min_db) * p.byte_mul);
          return fminf(fmaxf(bv, 0.f), 255.f);
        }
      }
    };

    const long long row = out_row0 + (long long)fi * bins;
    if constexpr (OUT != OUT_DB_U8) {
      float* o = reinterpret_cast<float*>(out_base) + row;
      if (live) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int k = t + T * i;
          o[k] = finish(pk[i]);
          if (k != 0) o[M - k] = finish(pm[i]);
          else if (p.nyquist) o[M] = finish(pm[i]);
        }
        if (t == 0) o[M / 2] = finish(pmid);
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

    if constexpr (SHIFT > 0) {
#pragma unroll
      for (int j = 0; j < 8 - SHIFT; ++j) raw[j] = raw[j + SHIFT];
      if (fi + 1 < nf) {
#pragma unroll
        for (int j = 0; j < SHIFT; ++j) raw[8 - SHIFT + j] = nxt[j];
      }
    }
    // MULTIWAVE with an odd number of exchanges per frame would desynchronise the
    // ping-pong parity between iterations; `which` restarts at 0 each frame, which is
    // safe because every exchange is separated from the previous use of the same
    // buffer by at least one barrier in between.
    if constexpr (G::MULTIWAVE) {
      constexpr int n_ex = (G::REM == 0 ? G::P8 - 1 : G::P8) + 1 + (OUT == OUT_DB_U8 ? 1 : 0);
      if constexpr ((n_ex & 1) != 0 || n_ex < 2) __syncthreads();
    }
  }
}

}  // namespace sg
