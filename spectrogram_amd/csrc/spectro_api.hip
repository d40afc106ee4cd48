// spectro_api.hip -- implementation of the C ABI declared in include/spectro.h.
//
// Host side only: argument checking, table upload, launch geometry, staging of
// host buffers, and the poll-mode state (sample history + smoothing).  All the
// arithmetic of the hot path is in stft_kernel.hpp; there is no CPU fallback --
// if HIP is unusable every call fails with SG_ERR_HIP.
#include "../../include/spectro.h"

#include <hip/hip_runtime.h>
#include <hip/hip_version.h>
#include <dlfcn.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "launch.hpp"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace

// shared with wav.cpp: records the thread-local message and returns `code`
int sg_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

namespace {

#define SG_HIP(expr)                                                                   \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess)                                                              \
      return fail(SG_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                  __FILE__, __LINE__);                                                 \
  } while (0)

int ilog2(int n) {
  int l = 0;
  while ((1 << l) < n) ++l;
  return l;
}

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return SG_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return fail(SG_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    cap = bytes;
    return SG_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

}  // namespace

struct sg_handle {
  sg_config cfg;
  int log2n;
  int device;
  sg::f2* d_tw = nullptr;   // N entries
  sg::f2* d_win = nullptr;  // N/2 entries, pre-scaled
  uint2* d_lf = nullptr;    // logfreq_cols entries {i0 | i1 << 16, frac bits}
  float* d_mixw = nullptr;  // cfg.downmix weights where SG_DOWNMIX_SPEAKERS is not the mean (K floats), else null
  // Tuning / test switches, set through sg_debug_set (include/spectro.h); the library never reads the environment.
  // None of them changes a bit of the output.
  int run_len_override = 0;         // "run_len": frames per run
  bool no_lring = false;            // "no_lring": generic loads where the LDS sample ring would run
  bool lring_all = false;           // "lring_all": the 4-row ring variant at every ring size
  bool no_single_round = false;     // "no_single_round": 16-64-frame runs for launches one dispatch round could hold
  bool no_fused_smoothing = false;  // "no_fused_smoothing": two-pass batch smoothing
  bool no_pinned_direct = false;    // "no_pinned_direct": page-locked buffers through the copy engine, not in place
  int extra_lds = 0;                // "extra_lds": bytes of dynamic LDS added to every transform launch (occupancy throttle)
  bool no_gen_unit = false;         // "no_gen_unit": plain rows through the general generic-load instantiation (A/B, tests)
  bool no_lffast = false;           // "no_lffast": 256-column rows through the general log-frequency instantiation (A/B, tests)
  bool pinned_direct_generic = false;  // "pinned_direct_generic": page-locked input read in place even by the generic loads (A/B only)
  int n_cu = 256;                   // compute units of the handle's device (run-length geometry only)
  // staging for host-pointer calls
  DevBuf stage_in, stage_out, stage_mag;
  // poll mode
  int n_channels = 0;
  long long hist_keep = 0;  // H: samples of history retained per channel (cfg.history_samples, or N)
  float* d_hist = nullptr;  // [n_channels][2H]: double-written ring, sample L at L mod H and at L mod H + H (aux_kernels.hip)
  float* d_state = nullptr; // [n_channels][N/2]
  DevBuf poll_out;          // device staging for host-side poll results
  unsigned char* d_ring = nullptr;  // [n_channels][ring_h][N/2] byte history (the reference's texture)
  int ring_h = 0, ring_off = 0;
  int ring_pinned = 0;      // 1: every poll writes row 0 (the reference outside its sonogram modes, 3D/visualizer.js:385-390)
  long long hist_pos = 0;   // in (H, 2H]: the newest sample's upper copy ends here; the newest n <= H samples are [hist_pos - n, hist_pos)
  // the staging buffer of host-pointer calls may still be read by an asynchronous call (page-locked input staged by
  // sg_stft returns without waiting): the next writer on ANOTHER stream waits for this event first
  // page-locked staging of sg_push_poll_* calls that bring small pageable buffers (the N-API / Python analyser objects):
  // the block is copied here, the ONE fused launch reads it in place and stores the rows here, one synchronise, copy out
  void* pin_in = nullptr;
  void* pin_out = nullptr;
  size_t pin_in_cap = 0, pin_out_cap = 0;
  hipEvent_t stage_in_ev = nullptr;
  hipStream_t stage_in_stream = nullptr;
  bool stage_in_busy = false;
  // ... and so may the |X| scratch of two-pass batch smoothing, when the rows go to device / page-locked memory and the
  // call only enqueues (two such calls on two streams raced on it)
  hipEvent_t stage_mag_ev = nullptr;
  hipStream_t stage_mag_stream = nullptr;
  bool stage_mag_busy = false;
  bool dirty = true;        // samples arrived since the last frequency poll: the next poll runs the transform
  sg_launch_info last = {0, 0, 0, 0, 0};  // how the last transform was launched (sg_last_launch)
};

namespace {
struct AnalyserArgs {  // poll-mode epilogue parameters (stft_kernel.hpp, OUT_ANALYSER)
  float* state;
  unsigned char* ring_row;
  long long ring_stride;
  float tau, min_db, byte_mul;
  int len;
  // fused push + poll: the newest n_new samples of every channel still lie in the caller's block (sg_push_poll_*)
  const void* pend_src = nullptr;
  long long pend_stride = 0;
  int pend_new = 0, pend_s16 = 0;
  int pend_wstart = 0, pend_h = 0;  // where the window starts in the 2H-float history row, and H (the ring's other copy)
};
}  // namespace

namespace {

struct DeviceGuard {
  int prev = -1;
  bool changed = false;
  int enter(int dev) {
    SG_HIP(hipGetDevice(&prev));
    if (prev != dev) {
      SG_HIP(hipSetDevice(dev));
      changed = true;
    }
    return SG_OK;
  }
  ~DeviceGuard() {
    if (changed) (void)hipSetDevice(prev);
  }
};

// A handle-owned buffer (stage_in, stage_mag) is about to be rewritten on `stream`: if an earlier asynchronous call on another
// stream may still be using it, that stream's work is waited for first (device-side, no host stall)
int stage_acquire(bool busy, hipStream_t owner, hipEvent_t ev, hipStream_t stream) {
  if (busy && owner != stream) SG_HIP(hipStreamWaitEvent(stream, ev, 0));
  return SG_OK;
}
// the call that used the buffer returns without synchronising: remember where its users run
int stage_release_async(hipEvent_t* ev, hipStream_t* owner, bool* busy, hipStream_t stream) {
  if (!*ev) SG_HIP(hipEventCreateWithFlags(ev, hipEventDisableTiming));
  SG_HIP(hipEventRecord(*ev, stream));
  *owner = stream;
  *busy = true;
  return SG_OK;
}
int stage_in_acquire(sg_handle* h, hipStream_t stream) { return stage_acquire(h->stage_in_busy, h->stage_in_stream, h->stage_in_ev, stream); }
int stage_in_release_async(sg_handle* h, hipStream_t stream) {
  return stage_release_async(&h->stage_in_ev, &h->stage_in_stream, &h->stage_in_busy, stream);
}

int check_config(const sg_config& c) {
  if (c.fft_size < (1 << sg::kMinLog2N) || c.fft_size > 32768 || (c.fft_size & (c.fft_size - 1)))
    return fail(SG_ERR_INVALID, "fft_size must be a power of two in [32, 32768], got %d", c.fft_size);
  if (c.fft_size > (1 << sg::kMaxLog2N))
    return fail(SG_ERR_UNSUPPORTED, "fft_size %d: this build has kernels up to %d", c.fft_size, 1 << sg::kMaxLog2N);
  if (c.hop < 1) return fail(SG_ERR_INVALID, "hop must be >= 1, got %d", c.hop);
  if (c.window < SG_WINDOW_HANN || c.window > SG_WINDOW_RECT) return fail(SG_ERR_INVALID, "bad window %d", c.window);
  if (c.scale < SG_SCALE_NONE || c.scale > SG_SCALE_INV_N) return fail(SG_ERR_INVALID, "bad scale %d", c.scale);
  if (c.out_kind < SG_OUT_POWER || c.out_kind > SG_OUT_DB_U8) return fail(SG_ERR_INVALID, "bad out_kind %d", c.out_kind);
  if (c.nyquist != 0 && c.nyquist != 1) return fail(SG_ERR_INVALID, "nyquist must be 0 or 1");
  if (!(c.max_db > c.min_db)) return fail(SG_ERR_INVALID, "max_db must exceed min_db");
  if (!(c.smoothing >= 0.f && c.smoothing <= 1.f)) return fail(SG_ERR_INVALID, "smoothing must be in [0, 1]");
  if (c.pcm_format < SG_PCM_F32 || c.pcm_format > SG_PCM_S24) return fail(SG_ERR_INVALID, "bad pcm_format %d", c.pcm_format);
  if (c.downmix < 0 || c.downmix > 64) return fail(SG_ERR_INVALID, "downmix must be in [0, 64]");
  if (c.downmix_mode != SG_DOWNMIX_MEAN && c.downmix_mode != SG_DOWNMIX_SPEAKERS) return fail(SG_ERR_INVALID, "bad downmix_mode %d", c.downmix_mode);
  if (c.logfreq_cols < 0 || c.logfreq_cols > 8192) return fail(SG_ERR_INVALID, "logfreq_cols must be in [0, 8192]");
  // (the column table keeps two 16-bit byte offsets into the float row: 4 * (N/2 - 1) = 65532 at N = 32768 still fits)
  if (c.logfreq_cols > 0 && c.nyquist) return fail(SG_ERR_INVALID, "logfreq_cols resamples the N/2-bin row: nyquist must be 0");
  if (c.logfreq_align != SG_LOGFREQ_PIXEL_CENTRE && c.logfreq_align != SG_LOGFREQ_VERTEX)
    return fail(SG_ERR_INVALID, "bad logfreq_align %d", c.logfreq_align);
  if (c.history_samples != 0 && (c.history_samples < c.fft_size || c.history_samples > (1 << 22)))
    return fail(SG_ERR_INVALID, "history_samples must be 0 or in [fft_size, 4194304], got %d", c.history_samples);
  if (c.logfreq_cols > 0) {
    // the column table (8 bytes per column) sits in LDS behind the exchange buffers of one workgroup
    const int need = sg::lds_bytes_per_block(ilog2(c.fft_size)) + 8 * c.logfreq_cols;
    if (need > sg::kLdsPerCu)
      return fail(SG_ERR_UNSUPPORTED, "logfreq_cols %d with fft_size %d needs %d bytes of LDS per workgroup (limit %d): use <= %d columns",
                  c.logfreq_cols, c.fft_size, need, sg::kLdsPerCu,
                  (sg::kLdsPerCu - sg::lds_bytes_per_block(ilog2(c.fft_size))) / 8);
  }
  return SG_OK;
}

// where column c of `cols` samples the row, as the texture coordinate u (sg_logfreq_align; the vertex form is the
// reference's mesh, 3D/visualizer.js:213-214: u = x / (sonogram3DWidth - 1))
double logfreq_u(int cols, double c, int align) {
  if (align == SG_LOGFREQ_VERTEX) return cols > 1 ? c / (double)(cols - 1) : 0.0;
  return (c + 0.5) / (double)cols;
}

// ---- Stored twiddle triplets of the sizes whose kernels derive W^3, W^5, W^6, W^7 of a radix-8 pass from W^1, W^2, W^4
// (N >= 4096 in batch mode, N >= 8192 in poll mode: stft_kernel.hpp, Consts::DERIVE).  A derived twiddle carries the
// table roundings of its two factors plus the rounding of the product -- about three times the error variance of a table
// entry -- and that showed: rms error 1.04-1.07 of the float32 radix-2 twin's at N = 4096 / 8192 against 0.94 with all
// seven twiddles held.  The products are deterministic, so the host picks the triplet for them:
// every component of (W^1, W^2, W^4) is tried at its nearest float and the two floats next to it (3^6 triplets), the
// four products are formed exactly as the kernel forms them (pk_cmul: one rounded product, one fused multiply-add per
// component), and the triplet with the least squared error over all seven twiddles against the float64 values is
// stored.  Squared error summed over the seven twiddles of one k, in units of (2^-24)^2 (tests/test_twiddles.py): table
// entries 1.0, derived from nearest floats 3.2, derived from the chosen triplets 1.5.  No instruction changes in the kernel; it reads these triplets instead of
// three entries of the W_N^k table.  Layout behind the N table entries: pass ps = 1, 2, ... (Ns = 8^ps twiddle indices
// k each) at offset (8^ps - 8) / 7, three float2 per k.
struct Cf {
  float x, y;
};
inline Cf kernel_cmul(Cf a, Cf w) {  // pk_math.hpp pk_cmul: d = (a.x w.x, a.y w.x); d = (fma(-a.y, w.y, d.x), fma(a.x, w.y, d.y))
  const float px = a.x * w.x, py = a.y * w.x;
  return Cf{std::fmaf(-a.y, w.y, px), std::fmaf(a.x, w.y, py)};
}
void derived_triplets(int n, std::vector<float>& out) {
  const int m = n / 2;
  int log2m = 0;
  while ((1 << log2m) < m) ++log2m;
  const int p8 = log2m / 3;
  const double two_pi = 6.283185307179586476925286766559;
  for (int ps = 1; ps < p8; ++ps) {
    const int ns = 1 << (3 * ps);
    for (int k = 0; k < ns; ++k) {
      double ex[7][2];
      for (int r = 1; r <= 7; ++r) {
        const double a = two_pi * (double)k * (double)r / (8.0 * (double)ns);
        ex[r - 1][0] = std::cos(a);
        ex[r - 1][1] = -std::sin(a);
      }
      const int base[3] = {0, 1, 3};  // W^1, W^2, W^4
      float cand[6][3];
      for (int c = 0; c < 6; ++c) {
        const float v = (float)ex[base[c / 2]][c & 1];
        cand[c][0] = v;
        cand[c][1] = std::nextafterf(v, -2.f);
        cand[c][2] = std::nextafterf(v, 2.f);
      }
      double best = 1e300;
      float keep[6] = {cand[0][0], cand[1][0], cand[2][0], cand[3][0], cand[4][0], cand[5][0]};
      for (int code = 0; code < 729; ++code) {
        float t[6];
        for (int c = 0, q = code; c < 6; ++c, q /= 3) t[c] = cand[c][q % 3];
        const Cf b0{t[0], t[1]}, b1{t[2], t[3]}, b2{t[4], t[5]};
        const Cf w3 = kernel_cmul(b0, b1), w5 = kernel_cmul(b0, b2), w6 = kernel_cmul(b1, b2), w7 = kernel_cmul(w3, b2);
        const Cf all[7] = {b0, b1, w3, b2, w5, w6, w7};
        double err = 0.0;
        for (int r = 0; r < 7; ++r) {
          const double dx = (double)all[r].x - ex[r][0], dy = (double)all[r].y - ex[r][1];
          err += dx * dx + dy * dy;
        }
        if (err < best) {  // code 0 (the nearest floats) comes first and wins ties
          best = err;
          for (int c = 0; c < 6; ++c) keep[c] = t[c];
        }
      }
      out.insert(out.end(), keep, keep + 6);
    }
  }
}

// device tables of one configuration: built aside, then swapped into the handle (sg_create; sg_set_fft_size on a live handle)
struct Tables {
  sg::f2* d_tw = nullptr;
  sg::f2* d_win = nullptr;
  uint2* d_lf = nullptr;
  float* d_mixw = nullptr;
  void release() {
    if (d_tw) (void)hipFree(d_tw);
    if (d_win) (void)hipFree(d_win);
    if (d_lf) (void)hipFree(d_lf);
    if (d_mixw) (void)hipFree(d_mixw);
    d_tw = nullptr;
    d_win = nullptr;
    d_lf = nullptr;
    d_mixw = nullptr;
  }
};

// The W3C "speakers" down-mix of K channels to mono (AnalyserNode input mixing; channel order L R C LFE SL SR).  Returns
// false where it IS the mean (K = 1, 2, 4: those keep the mean's code path, bit for bit); otherwise K weights: 5.1, or
// "discrete" (channel 0 alone) for channel counts that have no speaker layout.
bool speakers_weights(int k, float* w) {
  if (k <= 2 || k == 4) return false;
  for (int c = 0; c < k; ++c) w[c] = 0.f;
  if (k == 6) {
    w[0] = w[1] = 0.70710678118654752440f;  // sqrt(1/2) (L + R)
    w[2] = 1.f;                             // + C
    w[4] = w[5] = 0.5f;                     // + 0.5 (SL + SR); LFE (3) dropped
  } else {
    w[0] = 1.f;
  }
  return true;
}

int make_tables_(const sg_config& cfg, Tables* t) {
  const int n = cfg.fft_size, m = n / 2;
  std::vector<float> tw(2 * (size_t)n), win(2 * (size_t)m);
  const double two_pi = 6.283185307179586476925286766559;
  for (int k = 0; k < n; ++k) {
    const double a = two_pi * (double)k / (double)n;
    tw[2 * k] = (float)std::cos(a);
    tw[2 * k + 1] = (float)(-std::sin(a));
  }
  // 1/2 of the real-input split and the optional 1/N are folded into the window
  // ... and so is the 1/32768 of int16 (1/8388608 of int24) input
  const double c = 0.5 * (cfg.scale == SG_SCALE_INV_N ? 1.0 / (double)n : 1.0) *
                   (cfg.pcm_format == SG_PCM_S16 ? 1.0 / 32768.0 : cfg.pcm_format == SG_PCM_S24 ? 1.0 / 8388608.0 : 1.0);
  for (int i = 0; i < n; ++i) {
    const double a = two_pi * (double)i / (double)n;
    double w = 1.0;
    if (cfg.window == SG_WINDOW_HANN) w = 0.5 - 0.5 * std::cos(a);
    else if (cfg.window == SG_WINDOW_BLACKMAN) w = 0.42 - 0.5 * std::cos(a) + 0.08 * std::cos(2.0 * a);
    win[i] = (float)(c * w);
  }
  if (n >= 4096) derived_triplets(n, tw);  // behind the N entries of W_N^k
  SG_HIP(hipMalloc((void**)&t->d_tw, sizeof(float) * tw.size()));
  SG_HIP(hipMalloc((void**)&t->d_win, sizeof(float) * 2 * m));
  SG_HIP(hipMemcpy(t->d_tw, tw.data(), sizeof(float) * tw.size(), hipMemcpyHostToDevice));
  SG_HIP(hipMemcpy(t->d_win, win.data(), sizeof(float) * 2 * m, hipMemcpyHostToDevice));
  if (cfg.logfreq_cols > 0) {
    // column c samples the row of m bins at x = 256^(u-1), u by cfg.logfreq_align, GL_LINEAR with
    // CLAMP_TO_EDGE: texel centres at (i + 0.5)/m (sonogram-vertex.shader:51-52, visualizer.js:313-316)
    const int cols = cfg.logfreq_cols;
    std::vector<uint32_t> tab(2 * (size_t)cols);
    for (int c = 0; c < cols; ++c) {
      const double u = logfreq_u(cols, (double)c, cfg.logfreq_align);
      const double pp = std::pow(256.0, u - 1.0) * (double)m - 0.5;
      const double fl = std::floor(pp);
      int i0 = (int)fl, i1;
      float fr = (float)(pp - fl);
      // CLAMP_TO_EDGE, kept in the form "two NEIGHBOURING texels + a fraction" (the kernels read them as one pair and
      // blend a (1 - f) + b f, exact at f = 0 and 1): below the first texel centre -> texel 0 with f = 0, at or past the
      // last one -> texels m - 2, m - 1 with f = 1
      if (i0 < 0) {
        i0 = 0;
        fr = 0.f;
      } else if (i0 >= m - 1) {
        i0 = m - 2;
        fr = 1.f;
      }
      i1 = i0 + 1;
      tab[2 * c] = (uint32_t)(4 * i0) | ((uint32_t)(4 * i1) << 16);  // byte offsets into the float row
      std::memcpy(&tab[2 * c + 1], &fr, 4);
    }
    SG_HIP(hipMalloc((void**)&t->d_lf, sizeof(uint32_t) * 2 * cols));
    SG_HIP(hipMemcpy(t->d_lf, tab.data(), sizeof(uint32_t) * 2 * cols, hipMemcpyHostToDevice));
  }
  float mixw[64];
  if (cfg.downmix > 1 && cfg.downmix_mode == SG_DOWNMIX_SPEAKERS && speakers_weights(cfg.downmix, mixw)) {
    SG_HIP(hipMalloc((void**)&t->d_mixw, sizeof(float) * cfg.downmix));
    SG_HIP(hipMemcpy(t->d_mixw, mixw, sizeof(float) * cfg.downmix, hipMemcpyHostToDevice));
  }
  return SG_OK;
}

int make_tables(const sg_config& cfg, Tables* t) {
  const int rc = make_tables_(cfg, t);
  if (rc) t->release();  // nothing half-built survives a failure
  return rc;
}

int upload_tables(sg_handle* h) {
  Tables t;
  const int rc = make_tables(h->cfg, &t);
  if (rc) return rc;
  Tables old{h->d_tw, h->d_win, h->d_lf, h->d_mixw};
  old.release();
  h->d_tw = t.d_tw;
  h->d_win = t.d_win;
  h->d_lf = t.d_lf;
  h->d_mixw = t.d_mixw;
  return SG_OK;
}

hipError_t dispatch(int log2n, int shift, int out, const sg::StftParams& p, long long nb, hipStream_t s) {
  switch (log2n) {
#define SG_CASE(n) case n: return sg::launch_stft_##n(shift, out, p, nb, s);
    SG_CASE(5) SG_CASE(6) SG_CASE(7) SG_CASE(8) SG_CASE(9) SG_CASE(10) SG_CASE(11) SG_CASE(12) SG_CASE(13) SG_CASE(14) SG_CASE(15)
#undef SG_CASE
  }
  return hipErrorInvalidValue;
}

// Which load variant a transform of this shape takes (sg_launch_info.load_variant without the format bits): 2 / 4 the
// register sliding window, 1 / 3 the LDS sample ring -- both fetch every sample from memory ONCE per run -- or 0, the
// generic loads, which re-read the overlap every frame (served by L2 for device memory).
int pick_load_variant(const sg_handle* h, const void* d_pcm, int fmt, long long channel_stride, long long sample_stride,
                      int hop, bool an_poll, int out_kind, long long frames) {
  const int n = h->cfg.fft_size;
  const int tpf = sg::virtual_threads_per_frame(h->log2n);
  // sliding-window variant: hop == S * 2T, unit stride, rows aligned to one complex point
  // (poll mode reads the mono history: sg_push has mixed K-channel input down already)
  const int dmx = (!an_poll && h->cfg.downmix > 1) ? h->cfg.downmix : 1;
  int shift = 0;
  if (fmt != SG_PCM_S24 && dmx == 1 && sample_stride == 1 && hop % (2 * tpf) == 0 && (channel_stride % 2) == 0 &&
      (reinterpret_cast<uintptr_t>(d_pcm) % (fmt == SG_PCM_S16 ? 4 : 8)) == 0) {
    const int s = hop / (2 * tpf);
    if (s == 2 || s == 4) shift = s;
  }
  // LDS sample ring: any other hop up to N - 4 at unit stride (N = 1024 ... 8192; float32, int16, and packed int24 --
  // for which it is also the hop = N/4, N/2 path: twelve-byte granules have no register-window form): the frame slot
  // keeps its window in LDS and fetches every sample once per run.  Load variant 1 brings in up to 2 T granules of
  // four samples per frame (hop <= N/2 - 4), variant 3 up to 4 T (hop <= N - 4).
  if (shift == 0 && !an_poll && dmx == 1 && sample_stride == 1 && h->log2n >= sg::kLringMinLog2N &&
      h->log2n <= sg::kLringMaxLog2N && hop <= n - 4 && !h->no_lring) {
    shift = (hop + 3) / 4 + 1 <= 2 * tpf ? 1 : 3;
    // hops above N/2 overlap by less than half: the generic loads re-read under 2x and L2 absorbs it.  Measured
    // (tools/ring_vs_generic.py, profiles/r03_ring_vs_generic.txt): the 4-row ring wins 8-32 % at N = 2048 and 8192 and loses
    // 0-10 % at N = 1024 and 4096 (16 staging registers on top of a tighter register budget) -- it is used where it wins.
    // (packed int24 keeps the ring everywhere: its generic path is three byte loads per sample)
    if (shift == 3 && (h->log2n == 10 || h->log2n == 12) && fmt != SG_PCM_S24 && !h->lring_all) shift = 0;
  }
  // the ring addresses a run's granules with 32-bit byte offsets from the run's first sample: whole-channel runs (fused
  // batch smoothing) of more than 4 GiB of samples stay on the generic loads (64-bit frame addressing)
  if ((shift == 1 || shift == 3) && (out_kind & sg::OUT_SMOOTH) &&
      (frames * (long long)hop + n) * (long long)(fmt == SG_PCM_S16 ? 2 : fmt == SG_PCM_S24 ? 3 : 4) >= (1LL << 32))
    shift = 0;
  // generic loads of plain rows (unit stride, float32, no down-mix): the instantiation that carries those loads alone (100-112
  // VGPRs against 148-154, and no scratch at N >= 16384) -- where it wins (tools/generic_ab.py, profiles/r05_generic_ab.txt: N = 16384 /
  // 32768 +50 / +68 %, N = 512 +3 ... +13 %; N = 256 -13 %, N = 1024 / 2048 at hop >= N -1 ... -3 %: those keep variant 0)
  if (shift == 0 && !an_poll && fmt == SG_PCM_F32 && dmx == 1 && sample_stride == 1 && !h->no_gen_unit &&
      (h->log2n == 9 || h->log2n >= 14))
    shift = 5;
  return shift;
}

// true for the load variants that re-read the overlap between frames from memory (the generic loads, 0 and 5)
inline bool generic_loads(int variant) { return variant == 0 || variant == 5; }

// Enqueue the STFT kernel over device-resident data.
int enqueue_stft(sg_handle* h, const void* d_pcm, int fmt, long long n_channels, long long n_samples,
                 long long channel_stride, long long sample_stride, void* d_out,
                 long long out_channel_stride, int hop, int out_kind, int nyquist, hipStream_t stream,
                 const AnalyserArgs* an = nullptr, long long in_extent = 0, float smooth_tau = 0.f) {
  const int n = h->cfg.fft_size, m = n / 2;
  const long long frames = n_samples < n ? 0 : (n_samples - n) / hop + 1;
  if (frames == 0 || n_channels == 0) return SG_OK;
  const bool logf = (out_kind & sg::OUT_LOGFREQ) != 0;
  const int bins = logf ? h->cfg.logfreq_cols : m + (nyquist ? 1 : 0);
  const int tpf = sg::virtual_threads_per_frame(h->log2n);
  const int slots = sg::slots_per_block(h->log2n);

  const int dmx = (!an && h->cfg.downmix > 1) ? h->cfg.downmix : 1;
  const int shift = pick_load_variant(h, d_pcm, fmt, channel_stride, sample_stride, hop, an != nullptr, out_kind, frames);
  const bool lring = shift == 1 || shift == 3;
  const int shift_code = shift | (fmt << 3);  // SG_PCM_S16 = 1 -> 8, SG_PCM_S24 = 2 -> 16 (generic loads only)

  // frames per run: long enough to amortise the (N - hop) halo and the per-run
  // table loads, short enough that the grid still covers the chip several times
  long long total_frames = frames * n_channels;
  long long want_runs = (long long)(4 * h->n_cu) * 2048 / tpf / 4;  // ~4 resident rounds of the device's CUs (256 on MI355X)
  if (want_runs < 1) want_runs = 1;
  long long run_len = (total_frames + want_runs - 1) / want_runs;
  if (run_len < 16) run_len = 16;
  if (run_len > 64) run_len = 64;
  // A launch that cannot fill the chip's frame slots with 16-frame runs lasts as long as one run: cut it finer, down
  // to 4 frames a run (tools/short_runs.py, profiles/r03_short_runs.txt: N=512 hop 128, 7k frames 6.4 -> 3.1 us; the
  // LDS-ring variants below and the register-window variants of N >= 1024 have floors of their own)
  const long long all_slots = (long long)h->n_cu * 16 * 64 / tpf;  // at 4 waves per SIMD
  if (run_len * all_slots > total_frames) {
    long long rl = (total_frames + all_slots - 1) / all_slots;
    if (rl < 4) rl = 4;
    if (rl < run_len) run_len = rl;
  }
  if (h->run_len_override > 0) run_len = h->run_len_override;
  if (run_len > frames) run_len = frames;
  // generic loads re-read the overlap every frame, so a run shares nothing but its prologue (constants, address
  // setup: about one frame's worth of issue): 16-32 frames measured best (8: -7 %, 64: -2 %)
  if (generic_loads(shift) && h->run_len_override <= 0 && run_len > 32) run_len = 32;
  // A launch the chip can hold at once (<= 40 frames per resident frame slot): one exact dispatch round of equal runs,
  // the waves ranking themselves by remaining work so that they finish together (a third less
  // halo than 16-frame runs, no half-empty second round; measured 1-2 % on the 112k-frame workload).
  int prio_rem = 0;
  if (lring && h->run_len_override <= 0) {
    // A run's prologue fetches a whole window (N samples) for hop new samples per frame after it, so runs want to be
    // long (the (N - hop)-sample halo is re-read once per run); and the launch wants to be a whole number of dispatch
    // rounds of equal runs -- with 3.5 rounds the last one runs half empty (measured: 13 % on a 341k-frame batch).
    // k rounds of `resident` frame slots, k the smallest that keeps runs at <= 64 frames.
    const long long occ = h->log2n == 11 ? 3 : 4;                  // waves per SIMD these variants reach
    const long long resident = (long long)h->n_cu * 4 * occ * 64 / tpf;  // frame slots the chip holds at once
    long long k = (total_frames + resident * 64 - 1) / (resident * 64);
    if (k < 1) k = 1;
    long long rl = (total_frames + k * resident - 1) / (k * resident);
    if (rl < 4) rl = 4;  // (16 until late in round 3: a 3,500-frame call at N=2048 hop 735 took 8.5 us, 5.2 us with 4-frame runs)
    // runs are cut per channel: lengthen until the runs of all channels fit the k rounds
    while (rl < 96 && ((frames + rl - 1) / rl) * n_channels > k * resident) ++rl;
    run_len = rl > frames ? frames : rl;
  }
  if ((shift == 2 || shift == 4) && tpf >= 64 && h->run_len_override <= 0 && !h->no_single_round) {
    const long long resident = (long long)h->n_cu * 16 * 64 / tpf;  // frame slots at 4 waves/SIMD
    if (total_frames <= resident * 40) {
      const long long per_ch_slots = resident / n_channels > 0 ? resident / n_channels : 1;
      long long rl = (frames + per_ch_slots - 1) / per_ch_slots;
      // a launch that cannot fill the chip with 16-frame runs is cut finer: its duration is one run's, and a run of 7
      // frames still amortises its prologue and halo (tools/short_runs.py: 7,028 frames 16.6 -> 9.2 us, 14,059 frames
      // 16.7 -> 13.6 us, 28,122 and more unchanged)
      if (rl < 7) rl = 7;
      if (rl <= 64) {
        run_len = rl > frames ? frames : rl;
        prio_rem = 1;
      }
    }
  }

  if (out_kind & sg::OUT_SMOOTH) {
    // fused batch smoothing: the recurrence rides in registers along the run, so a run is the whole channel
    run_len = frames;
    prio_rem = 0;
  }

  sg::StftParams p;
  p.pcm = d_pcm;
  p.out = d_out;
  p.tw = h->d_tw;
  p.win = h->d_win;
  p.n_channels = n_channels;
  p.frames = frames;
  p.channel_stride = channel_stride;
  p.sample_stride = sample_stride;
  p.out_channel_stride = out_channel_stride ? out_channel_stride : frames * bins;
  p.runs_per_channel = (frames + run_len - 1) / run_len;
  p.total_runs = p.runs_per_channel * n_channels;
  p.hop = hop;
  p.run_len = (int)run_len;
  p.bins = bins;
  p.nyquist = nyquist;
  p.db_mul = 3.01029995663981195f;  // 10 / log2(10)
  p.floor_db = h->cfg.floor_db;
  {
    const double bm = 255.0 / ((double)h->cfg.max_db - (double)h->cfg.min_db);
    p.byte_a = (float)(3.01029995663981195 * bm);
    p.byte_b = (float)(-(double)h->cfg.min_db * bm);
    const double bf = ((double)h->cfg.floor_db - (double)h->cfg.min_db) * bm;
    p.byte_floor = (float)(bf > 0.0 ? bf : 0.0);
  }
  p.downmix = dmx;
  p.downmix_scale = 1.0f / (float)dmx;
  p.downmix_w = dmx > 1 ? h->d_mixw : nullptr;
  p.lf_tab = h->d_lf;
  p.lf_cols = logf ? h->cfg.logfreq_cols : 0;
  p.an_state = an ? an->state : nullptr;
  p.an_ring = an ? an->ring_row : nullptr;
  p.an_ring_stride = an ? an->ring_stride : 0;
  p.an_tau = an ? an->tau : smooth_tau;
  p.an_min_db = an ? an->min_db : 0.f;
  p.an_byte_mul = an ? an->byte_mul : 0.f;
  p.an_len = an ? an->len : 0;
  p.prio_rem = prio_rem;
  p.pend_src = an ? an->pend_src : nullptr;
  p.pend_stride = an ? an->pend_stride : 0;
  p.pend_split = an ? n - an->pend_new : 0;
  p.pend_s16 = an ? an->pend_s16 : 0;
  p.pend_wstart = an ? an->pend_wstart : 0;
  p.pend_h = an ? an->pend_h : 0;
  p.in_extent = in_extent;
  p.host_extra_lds = h->extra_lds;
  h->last.load_variant = shift_code;
  h->last.run_len = (int32_t)run_len;
  h->last.single_round = prio_rem;
  h->last.total_runs = p.total_runs;
  const long long nblocks = (p.total_runs + slots - 1) / slots;
  if (nblocks > 0x7fffffffLL) return fail(SG_ERR_INVALID, "grid too large (%lld blocks)", nblocks);
#if SG_STAMP || SG_STALL  // diagnostic builds only: per-run start / end stamps (or per-wave stall sums) of the last launch -> $SG_STAMP_FILE
  static unsigned long long* d_stamps = nullptr;
  static long long stamp_cap = 0;
  if (!logf && std::getenv("SG_STAMP_FILE")) {
    if (stamp_cap < p.total_runs) {
      if (d_stamps) (void)hipFree(d_stamps);
      SG_HIP(hipMalloc((void**)&d_stamps, (size_t)p.total_runs * 32));
      stamp_cap = p.total_runs;
    }
    p.lf_tab = reinterpret_cast<const uint2*>(d_stamps);
    SG_HIP(dispatch(h->log2n, shift_code, out_kind, p, nblocks, stream));
    SG_HIP(hipStreamSynchronize(stream));
    std::vector<unsigned long long> hs((size_t)p.total_runs * 4);  // per run: realtime start / end, shader-clock start / end
    SG_HIP(hipMemcpy(hs.data(), d_stamps, hs.size() * 8, hipMemcpyDeviceToHost));
    if (FILE* f = std::fopen(std::getenv("SG_STAMP_FILE"), "wb")) {
      std::fwrite(hs.data(), 8, hs.size(), f);
      std::fclose(f);
    }
    return SG_OK;
  }
#endif
  // the reference's row -- 256 columns (sonogram3DWidth, 3D/visualizer.js:54) at N = 1024 / 2048, all of them taken -- has an
  // instantiation that carries the consecutive-column path alone (stft_kernel.hpp, OUT_LFFAST)
  if (logf && h->cfg.logfreq_cols == 256 && (h->log2n == 10 || h->log2n == 11) && (!an || an->len == 256) && !h->no_lffast &&
      !(out_kind & sg::OUT_SMOOTH))
    out_kind |= sg::OUT_LFFAST;
  SG_HIP(dispatch(h->log2n, shift_code, out_kind, p, nblocks, stream));
  return SG_OK;
}

}  // namespace

extern "C" {

int sg_config_default(sg_config* cfg) {
  if (!cfg) return fail(SG_ERR_INVALID, "cfg is NULL");
  std::memset(cfg, 0, sizeof *cfg);
  cfg->struct_size = sizeof(sg_config);
  cfg->fft_size = 1024;
  cfg->hop = 256;
  cfg->window = SG_WINDOW_HANN;
  cfg->scale = SG_SCALE_NONE;
  cfg->out_kind = SG_OUT_POWER;
  cfg->nyquist = 0;
  cfg->min_db = -100.f;
  cfg->max_db = -30.f;
  cfg->floor_db = -200.f;
  cfg->smoothing = 0.f;
  cfg->device_id = -1;
  cfg->logfreq_cols = 0;
  cfg->pcm_format = SG_PCM_F32;
  cfg->downmix = 0;
  cfg->logfreq_align = SG_LOGFREQ_PIXEL_CENTRE;
  cfg->history_samples = 0;
  cfg->downmix_mode = SG_DOWNMIX_MEAN;
  return SG_OK;
}

int sg_version(void) { return SG_VERSION; }

int sg_runtime_info(sg_runtime* out) {
  if (!out) return fail(SG_ERR_INVALID, "NULL argument");
  std::memset(out, 0, sizeof *out);
  out->built_hip_version = HIP_VERSION;
  int v = 0;
  if (hipRuntimeGetVersion(&v) == hipSuccess) out->runtime_version = v;
  else (void)hipGetLastError();
  v = 0;
  if (hipDriverGetVersion(&v) == hipSuccess) out->driver_version = v;
  else (void)hipGetLastError();
  // the shared object that holds the runtime's entry point as THIS library resolved it
  Dl_info di;
  if (dladdr(reinterpret_cast<const void*>(&hipRuntimeGetVersion), &di) && di.dli_fname)
    std::snprintf(out->library_path, sizeof out->library_path, "%s", di.dli_fname);
  return SG_OK;
}

const char* sg_last_error(void) { return g_err; }

int sg_create(const sg_config* cfg, sg_handle** out) {
  if (!cfg || !out) return fail(SG_ERR_INVALID, "sg_create: NULL argument");
  *out = nullptr;
  if (cfg->struct_size != sizeof(sg_config))
    return fail(SG_ERR_INVALID, "sg_config.struct_size is %u, expected %zu (call sg_config_default first)",
                cfg->struct_size, sizeof(sg_config));
  int rc = check_config(*cfg);
  if (rc) return rc;
  int ndev = 0;
  SG_HIP(hipGetDeviceCount(&ndev));
  if (ndev < 1) return fail(SG_ERR_HIP, "no HIP device visible; libspectro has no CPU path");
  int dev = cfg->device_id;
  if (dev < 0) SG_HIP(hipGetDevice(&dev));
  if (dev >= ndev) return fail(SG_ERR_INVALID, "device_id %d out of range (%d devices)", dev, ndev);
  sg_handle* h = new (std::nothrow) sg_handle();
  if (!h) return fail(SG_ERR_NOMEM, "out of host memory");
  h->cfg = *cfg;
  h->cfg.device_id = dev;
  h->device = dev;
  h->log2n = ilog2(cfg->fft_size);
  int n_cu = 0;
  if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n_cu > 0) h->n_cu = n_cu;
  DeviceGuard g;
  rc = g.enter(dev);
  if (!rc) rc = upload_tables(h);
  if (rc) {
    sg_destroy(h);
    return rc;
  }
  *out = h;
  g_err[0] = 0;
  return SG_OK;
}

int sg_destroy(sg_handle* h) {
  if (!h) return SG_OK;
  DeviceGuard g;
  (void)g.enter(h->device);
  if (h->d_tw) (void)hipFree(h->d_tw);
  if (h->d_win) (void)hipFree(h->d_win);
  if (h->d_lf) (void)hipFree(h->d_lf);
  if (h->d_mixw) (void)hipFree(h->d_mixw);
  if (h->d_hist) (void)hipFree(h->d_hist);
  if (h->d_state) (void)hipFree(h->d_state);
  if (h->d_ring) (void)hipFree(h->d_ring);
  h->stage_in.release();
  h->stage_out.release();
  h->stage_mag.release();
  h->poll_out.release();
  if (h->stage_in_ev) (void)hipEventDestroy(h->stage_in_ev);
  if (h->stage_mag_ev) (void)hipEventDestroy(h->stage_mag_ev);
  if (h->pin_in) (void)hipHostFree(h->pin_in);
  if (h->pin_out) (void)hipHostFree(h->pin_out);
  delete h;
  return SG_OK;
}

int64_t sg_num_frames(const sg_handle* h, int64_t n_samples) {
  if (!h) return fail(SG_ERR_INVALID, "NULL handle");
  return n_samples < h->cfg.fft_size ? 0 : (n_samples - h->cfg.fft_size) / h->cfg.hop + 1;
}

int32_t sg_num_bins(const sg_handle* h) {
  if (!h) return fail(SG_ERR_INVALID, "NULL handle");
  if (h->cfg.logfreq_cols > 0) return h->cfg.logfreq_cols;
  return h->cfg.fft_size / 2 + (h->cfg.nyquist ? 1 : 0);
}

int32_t sg_out_elem_size(const sg_handle* h) {
  if (!h) return fail(SG_ERR_INVALID, "NULL handle");
  return h->cfg.out_kind == SG_OUT_DB_U8 ? 1 : 4;
}

int sg_get_config(const sg_handle* h, sg_config* out) {
  if (!h || !out) return fail(SG_ERR_INVALID, "NULL argument");
  *out = h->cfg;
  return SG_OK;
}

int sg_stft(sg_handle* h, const float* pcm, int64_t n_channels, int64_t n_samples,
            int64_t channel_stride, int64_t sample_stride, void* out, int64_t out_channel_stride,
            uint32_t flags, void* hip_stream) {
  if (!h) return fail(SG_ERR_INVALID, "NULL handle");
  if (n_channels < 0 || n_samples < 0) return fail(SG_ERR_INVALID, "negative size");
  if (sample_stride < 1) return fail(SG_ERR_INVALID, "sample_stride must be >= 1");
  if (channel_stride < 0 || out_channel_stride < 0) return fail(SG_ERR_INVALID, "negative stride");
  const int64_t frames = sg_num_frames(h, n_samples);
  if (frames == 0 || n_channels == 0) return SG_OK;  // ragged / empty input: nothing to write
  if (!pcm || !out) return fail(SG_ERR_INVALID, "NULL buffer");
  const int bins = sg_num_bins(h);
  const int esz = sg_out_elem_size(h);
  if (out_channel_stride == 0) out_channel_stride = frames * bins;
  if (out_channel_stride < frames * bins && n_channels > 1)
    return fail(SG_ERR_INVALID, "out_channel_stride %lld < frames*bins %lld", (long long)out_channel_stride,
                (long long)(frames * bins));
  hipStream_t stream = (hipStream_t)hip_stream;
  DeviceGuard g;
  int rc = g.enter(h->device);
  if (rc) return rc;

  const int fmt = h->cfg.pcm_format;
  const size_t ssz = fmt == SG_PCM_S16 ? 2 : fmt == SG_PCM_S24 ? 3 : 4;
  const void* d_in = pcm;
  void* d_out = out;
  long long in_cs = channel_stride, in_ss = sample_stride;
  // extent of the input actually touched, in floats
  const long long kmix = h->cfg.downmix > 1 ? h->cfg.downmix : 1;  // n_samples counts analysed (down-mixed) samples
  const long long in_extent = (n_channels - 1) * channel_stride + (n_samples * kmix - 1) * sample_stride + 1;
  const size_t row_bytes = (size_t)(frames * bins) * esz;  // one channel's rows, dense
  // page-locked host buffers (SG_IN_PINNED / SG_OUT_PINNED): mapped into the device's address space, the kernel reads the
  // samples and stores the rows in place -- input and output cross PCIe at the same time instead of one copy after the
  // other; the call still synchronises the stream before it returns unless both sides are device / pinned
  bool in_dev = (flags & SG_IN_DEVICE) != 0, out_dev = (flags & SG_OUT_DEVICE) != 0;
  // frame slots the chip holds at once (4 waves per SIMD): with at least half that many channels a run can be a whole
  // channel without starving the chip, and batch smoothing is fused into the transform
  const long long resident_slots = (long long)h->n_cu * 16 * 64 / sg::virtual_threads_per_frame(h->log2n);
  const bool fuse_smoothing = h->cfg.smoothing > 0.f && frames < 0x7fffffffLL &&
                              2 * n_channels >= resident_slots && !h->no_fused_smoothing;
  bool in_async = false;  // page-locked input that goes through the staging copy: a true DMA, the call need not wait for it
  bool used_stage_mag = false;  // two-pass smoothing ran through the |X| scratch
  if (!in_dev && (flags & SG_IN_PINNED)) {
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, const_cast<float*>(pcm), 0) != hipSuccess || !dp) {
      (void)hipGetLastError();
      return fail(SG_ERR_INVALID, "SG_IN_PINNED: the input buffer is not page-locked, device-mapped host memory");
    }
    // In place when the overlap between frames is not re-fetched over PCIe: always for the load variants that fetch every
    // sample ONCE (register window, LDS ring); for the generic loads when frames do not overlap (hop >= N) or the window is
    // small (N <= 512, unit stride: the re-reads of a run hit in L1 / L2 -- measured 5x ... 32x overlap at N = 256 / 512: in
    // place 1.48 / 2.73 / 1.91 ms against 1.54 / 2.85 / 2.34 ms staged).  Large windows do re-read over the link (N = 16384,
    // hop 1000, device output: 7.0 ms in place against 1.7 ms staged), and strided / down-mixed rows fetch lines they use a
    // part of: those shapes -- and "no_pinned_direct" -- take the staged copy below, which from page-locked memory is a plain
    // async DMA (tools/bench_host.py).
    const int variant = pick_load_variant(h, dp, fmt, channel_stride, sample_stride, h->cfg.hop, false,
                                          fuse_smoothing ? sg::OUT_SMOOTH : 0, frames);
    const bool generic_ok = h->cfg.hop >= h->cfg.fft_size ||
                            (h->cfg.fft_size <= 512 && sample_stride == 1 && h->cfg.downmix <= 1);
    if ((!generic_loads(variant) || generic_ok || h->pinned_direct_generic) && !h->no_pinned_direct) {
      d_in = dp;
      in_dev = true;
    } else {
      in_async = true;
    }
  }
  if (!out_dev && (flags & SG_OUT_PINNED)) {
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, out, 0) != hipSuccess || !dp) {
      (void)hipGetLastError();
      return fail(SG_ERR_INVALID, "SG_OUT_PINNED: the output buffer is not page-locked, device-mapped host memory");
    }
    d_out = dp;
    out_dev = true;
  }
  if (!in_dev) {
    rc = stage_in_acquire(h, stream);
    if (rc) return rc;
    if ((size_t)in_extent * ssz > h->stage_in.cap && h->stage_in_busy) {  // growing frees the buffer: nothing may still read it
      SG_HIP(hipStreamSynchronize(h->stage_in_stream));
      h->stage_in_busy = false;
    }
    rc = h->stage_in.reserve((size_t)in_extent * ssz);
    if (rc) return rc;
    SG_HIP(hipMemcpyAsync(h->stage_in.p, pcm, (size_t)in_extent * ssz, hipMemcpyHostToDevice, stream));
    d_in = h->stage_in.p;
  }
  long long d_out_cs = out_channel_stride;
  if (!out_dev) {
    // staged densely: the gaps of a padded caller layout are never written
    rc = h->stage_out.reserve(row_bytes * (size_t)n_channels);
    if (rc) return rc;
    d_out = h->stage_out.p;
    d_out_cs = frames * bins;
  }
  if (fuse_smoothing) {
    // batch-mode time smoothing (3D/visualizer.js:351), many channels: ONE launch, no |X| scratch -- the smoothed
    // magnitudes of a lane's bins stay in its registers from frame to frame of the channel
    // (with cfg.logfreq_cols the row that leaves is the log-frequency columns of the smoothed bins)
    rc = enqueue_stft(h, d_in, fmt, n_channels, n_samples, in_cs, in_ss, d_out, d_out_cs, h->cfg.hop,
                      h->cfg.out_kind | sg::OUT_SMOOTH | (h->cfg.logfreq_cols > 0 ? sg::OUT_LOGFREQ : 0), h->cfg.nyquist, stream,
                      nullptr, in_extent, h->cfg.smoothing);
    if (rc) return rc;
  } else if (h->cfg.smoothing > 0.f) {
    // few channels: |X| of every bin into a scratch buffer, then the recurrence along frames in chunks (aux_kernels.hip),
    // which also samples the log-frequency columns out of the smoothed row when they are configured
    const int mbins = h->cfg.fft_size / 2 + (h->cfg.nyquist ? 1 : 0);   // the scratch holds bins, whatever the row is
    const size_t need = (size_t)n_channels * frames * mbins * sizeof(float);
    // the scratch may still be read by an earlier call that only enqueued, on another stream (device-side wait); growing
    // frees it, so then nothing may be in flight at all
    rc = stage_acquire(h->stage_mag_busy, h->stage_mag_stream, h->stage_mag_ev, stream);
    if (rc) return rc;
    if (need > h->stage_mag.cap && h->stage_mag_busy) {
      SG_HIP(hipStreamSynchronize(h->stage_mag_stream));
      h->stage_mag_busy = false;
    }
    rc = h->stage_mag.reserve(need);
    if (rc) return rc;
    rc = enqueue_stft(h, d_in, fmt, n_channels, n_samples, in_cs, in_ss, h->stage_mag.p, frames * mbins, h->cfg.hop,
                      SG_OUT_MAG, h->cfg.nyquist, stream, nullptr, in_extent);
    if (rc) return rc;
    SG_HIP(sg::launch_smooth_scan((const float*)h->stage_mag.p, d_out, h->cfg.out_kind, n_channels, frames, mbins,
                                  d_out_cs, h->cfg.smoothing, h->cfg.floor_db, h->cfg.min_db,
                                  255.0f / (h->cfg.max_db - h->cfg.min_db), h->cfg.logfreq_cols > 0 ? h->d_lf : nullptr,
                                  h->cfg.logfreq_cols, stream));
    used_stage_mag = true;
  } else {
    rc = enqueue_stft(h, d_in, fmt, n_channels, n_samples, in_cs, in_ss, d_out, d_out_cs, h->cfg.hop,
                      h->cfg.out_kind | (h->cfg.logfreq_cols > 0 ? sg::OUT_LOGFREQ : 0), h->cfg.nyquist, stream, nullptr,
                      in_extent);
    if (rc) return rc;
  }
  if (!out_dev) {
    if (out_channel_stride == frames * bins || n_channels == 1)
      SG_HIP(hipMemcpyAsync(out, d_out, row_bytes * (size_t)n_channels, hipMemcpyDeviceToHost, stream));
    else
      SG_HIP(hipMemcpy2DAsync(out, (size_t)out_channel_stride * esz, d_out, row_bytes, row_bytes, (size_t)n_channels,
                              hipMemcpyDeviceToHost, stream));
  }
  // host pointers: results (and the right to reuse the input) on return.  Device pointers: only enqueued.  Page-locked
  // host buffers on both sides (or one side device, the other pinned): enqueued too -- synchronise the stream before
  // reading the rows or rewriting the samples.
  if (!((in_dev || in_async) && out_dev)) {
    SG_HIP(hipStreamSynchronize(stream));
    if (!in_dev) h->stage_in_busy = false;  // this stream waited for the buffer's earlier readers, and has drained
    if (used_stage_mag) h->stage_mag_busy = false;
  } else {
    if (in_async) {
      // the staging copy and the kernel that reads it are still in flight when this call returns
      rc = stage_in_release_async(h, stream);
      if (rc) return rc;
    }
    if (used_stage_mag) {  // ... and so are the transform that fills the |X| scratch and the scan that reads it
      rc = stage_release_async(&h->stage_mag_ev, &h->stage_mag_stream, &h->stage_mag_busy, stream);
      if (rc) return rc;
    }
  }
  return SG_OK;
}

/* ---- poll mode --------------------------------------------------------- */

int sg_stream_begin(sg_handle* h, int32_t n_channels) {
  if (!h) return fail(SG_ERR_INVALID, "NULL handle");
  if (n_channels < 1) return fail(SG_ERR_INVALID, "n_channels must be >= 1");
  DeviceGuard g;
  int rc = g.enter(h->device);
  if (rc) return rc;
  if (h->d_hist) (void)hipFree(h->d_hist);
  if (h->d_state) (void)hipFree(h->d_state);
  h->d_hist = h->d_state = nullptr;
  if (h->d_ring) (void)hipFree(h->d_ring);
  h->d_ring = nullptr;
  h->ring_h = h->ring_off = 0;
  h->dirty = true;  // the first poll transforms the N zeros of history
  const size_t n = h->cfg.fft_size, m = n / 2;
  const size_t keep = h->cfg.history_samples > 0 ? (size_t)h->cfg.history_samples : n;  // H >= N samples retained per channel
  SG_HIP(hipMalloc((void**)&h->d_hist, sizeof(float) * 2 * keep * n_channels));
  SG_HIP(hipMalloc((void**)&h->d_state, sizeof(float) * m * n_channels));
  SG_HIP(hipMemset(h->d_hist, 0, sizeof(float) * 2 * keep * n_channels));
  SG_HIP(hipMemset(h->d_state, 0, sizeof(float) * m * n_channels));
  h->n_channels = n_channels;
  h->hist_keep = (long long)keep;
  h->hist_pos = 2 * (long long)keep;  // H zeros of history: polls before N samples arrive see silence
  return SG_OK;
}

// The history rows are double-written rings (aux_kernels.hip): `take` new samples go to ring positions a0, a0 + 1, ...
// (mod H) and to the same positions + H; afterwards the newest n <= H samples are [hist_pos - n, hist_pos) again.
static long long hist_ring_pos(const sg_handle* h) { return h->hist_pos - h->hist_keep == h->hist_keep ? 0 : h->hist_pos - h->hist_keep; }
static void hist_advance(sg_handle* h, long long take) {
  long long e = hist_ring_pos(h) + take;  // a0 < H, take <= H
  if (e <= h->hist_keep) e += h->hist_keep;
  h->hist_pos = e;
}

int sg_push(sg_handle* h, const float* pcm, int64_t n_new, int64_t channel_stride, uint32_t flags,
            void* hip_stream) {
  if (!h) return fail(SG_ERR_INVALID, "NULL handle");
  if (!h->d_hist) return fail(SG_ERR_STATE, "sg_push before sg_stream_begin");
  if (n_new < 0) return fail(SG_ERR_INVALID, "negative n_new");
  if (n_new == 0) return SG_OK;
  if (!pcm) return fail(SG_ERR_INVALID, "NULL pcm");
  // cfg.downmix = K > 1: every analysed sample is the mean of K interleaved input samples (the reference's sources reach
  // the analyser through the summing `mix` node, UI/player.js:25,110-115); n_new counts analysed samples, a channel's
  // row holds n_new * K floats
  const long long kmix = h->cfg.downmix > 1 ? h->cfg.downmix : 1;
  if (channel_stride == 0) channel_stride = n_new * kmix;
  if (channel_stride < n_new * kmix) return fail(SG_ERR_INVALID, "channel_stride < n_new%s", kmix > 1 ? " * downmix" : "");
  hipStream_t stream = (hipStream_t)hip_stream;
  DeviceGuard g;
  int rc = g.enter(h->device);
  if (rc) return rc;
  const long long n = h->hist_keep, row = 2 * n;   // n = H: the samples of history a channel retains (>= fft_size)
  const int fmt = h->cfg.pcm_format;                   // float32, or int16 / packed int24 (converted on the way into the float history)
  const size_t ssz = fmt == SG_PCM_S16 ? 2 : fmt == SG_PCM_S24 ? 3 : 4;
  const float* mixw = kmix > 1 ? h->d_mixw : nullptr;  // SG_DOWNMIX_SPEAKERS where it is not the mean
  const long long take = n_new >= n ? n : n_new;       // only the latest H samples can matter
  const unsigned char* src = reinterpret_cast<const unsigned char*>(pcm) + (size_t)((n_new - take) * kmix) * ssz;
  const int a0 = (int)hist_ring_pos(h);
  // page-locked host input: a kernel reads it in place over PCIe (mapped into the device's address space) -- no DMA
  // descriptor set-up, which costs ~25 us per hop whatever the size; "no_pinned_direct" keeps the copy engine
  const void* mapped = nullptr;
  if ((flags & SG_IN_PINNED) && !(flags & SG_IN_DEVICE) && !h->no_pinned_direct) {
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, const_cast<unsigned char*>(src), 0) == hipSuccess && dp) mapped = dp;
    else (void)hipGetLastError();  // not mapped: fall back to the copy
  }
  bool staged_async = false;
  if (mapped) {
    SG_HIP(sg::launch_push_mix(mapped, fmt, channel_stride, h->d_hist, row, h->n_channels, (int)take, (int)kmix, a0, (int)n, mixw, stream));
  } else {
    const void* d_src = src;
    long long d_stride = channel_stride;
    if (!(flags & SG_IN_DEVICE)) {  // rows staged densely as they are, then converted / mixed into the ring on the device
      rc = stage_in_acquire(h, stream);
      if (rc) return rc;
      const size_t need = (size_t)h->n_channels * take * kmix * ssz;
      if (need > h->stage_in.cap && h->stage_in_busy) {
        SG_HIP(hipStreamSynchronize(h->stage_in_stream));
        h->stage_in_busy = false;
      }
      rc = h->stage_in.reserve(need);
      if (rc) return rc;
      SG_HIP(hipMemcpy2DAsync(h->stage_in.p, take * kmix * ssz, src, channel_stride * ssz,
                              take * kmix * ssz, h->n_channels, hipMemcpyHostToDevice, stream));
      d_src = h->stage_in.p;
      d_stride = take * kmix;
      staged_async = (flags & SG_IN_PINNED) != 0;
    }
    SG_HIP(sg::launch_push_mix(d_src, fmt, d_stride, h->d_hist, row, h->n_channels, (int)take, (int)kmix, a0, (int)n, mixw, stream));
  }
  if (staged_async) {
    rc = stage_in_release_async(h, stream);
    if (rc) return rc;
  }
  hist_advance(h, take);
  h->dirty = true;
  // pageable host memory: the caller may reuse pcm on return.  SG_IN_PINNED (memory from sg_alloc_pinned or otherwise
  // page-locked): a kernel ordered on the stream reads pcm in place (or the copy engine does, "no_pinned_direct") and
  // this call does not wait: pcm may be rewritten only after a stream synchronise, an event recorded behind this call,
  // or a poll on the same stream that waits (include/spectro.h, SG_IN_PINNED).
  if (!(flags & (SG_IN_DEVICE | SG_IN_PINNED))) {
    SG_HIP(hipStreamSynchronize(stream));
    h->stage_in_busy = false;  // (pageable rows are always staged) this stream waited for the buffer's earlier readers, and has drained
  }
  return SG_OK;
}

// elements of one poll-mode row: the N/2 bins (the Nyquist bin is never part of it, 3D/visualizer.js:301), or their
// log-frequency columns
static int poll_row_len(const sg_handle* h) { return h->cfg.logfreq_cols > 0 ? h->cfg.logfreq_cols : h->cfg.fft_size / 2; }

struct PendingBlock {  // sg_push_poll_*: samples the fused launch reads straight from the caller's block
  const void* src = nullptr;
  long long stride = 0;
  int n_new = 0, s16 = 0;
};

static int poll_freq(sg_handle* h, float* f32_out, uint8_t* u8_out, int64_t len, int64_t stride,
                     uint32_t flags, hipStream_t stream, const PendingBlock* pend = nullptr) {
  if (!h) return fail(SG_ERR_INVALID, "NULL handle");
  if (!h->d_hist) return fail(SG_ERR_STATE, "poll before sg_stream_begin");
  if (len < 0) return fail(SG_ERR_INVALID, "negative len");
  if (!f32_out && !u8_out) return fail(SG_ERR_INVALID, "NULL output");
  const int n = h->cfg.fft_size, m = n / 2;
  // a row is the N/2 bins, or -- with cfg.logfreq_cols -- their log-frequency columns (what the reference's shader
  // samples out of the row, sonogram-vertex.shader:51-52): the poll hands out, and the ring keeps, that row
  const bool logf = h->cfg.logfreq_cols > 0;
  const int row = poll_row_len(h);
  if (len > row) len = row;  // short or long caller buffers: write min(len, row), like the platform
  if (stride == 0) stride = len;
  if (stride < len) return fail(SG_ERR_INVALID, "bins_stride < len");
  // the history is float32 mono: sg_push mixes K-channel input down and converts int16 / packed int24 input on the way in
  // (as the float of the integer; the window carries the 1/32768 or 1/8388608, make_tables_)
  DeviceGuard g;
  int rc = g.enter(h->device);
  if (rc) return rc;
  bool dev_out = (flags & SG_OUT_DEVICE) != 0;
  const size_t esz = u8_out ? 1 : 4;
  void* d_dst = u8_out ? (void*)u8_out : (void*)f32_out;
  long long d_stride = stride;
  if (!dev_out && (flags & SG_OUT_PINNED)) {
    // page-locked host output: the kernel stores the rows there itself (no staging buffer, no D2H copy, no wait)
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, d_dst, 0) != hipSuccess || !dp) {
      (void)hipGetLastError();
      return fail(SG_ERR_INVALID, "SG_OUT_PINNED: the output buffer is not page-locked, device-mapped host memory");
    }
    d_dst = dp;
    dev_out = true;
  }
  if (!dev_out) {
    rc = h->poll_out.reserve((size_t)h->n_channels * len * esz + 16);
    if (rc) return rc;
    d_dst = h->poll_out.p;
    d_stride = len;
  }
  unsigned char* ring_row = nullptr;
  if (u8_out && h->d_ring) ring_row = h->d_ring + (size_t)h->ring_off * row;
  const float byte_mul = 255.0f / (h->cfg.max_db - h->cfg.min_db);
  if (h->dirty) {
    // new samples since the last poll: ONE launch -- transform of the latest N samples of every channel with
    // the analyser epilogue fused (smoothing state, 20 log10, dB or byte, log-frequency columns, ring row);
    // nothing round-trips HBM
    AnalyserArgs an{h->d_state, ring_row, (long long)h->ring_h * row, h->cfg.smoothing, h->cfg.min_db, byte_mul, (int)len};
    if (pend) {
      an.pend_src = pend->src;
      an.pend_stride = pend->stride;
      an.pend_new = pend->n_new;
      an.pend_s16 = pend->s16;
      an.pend_wstart = (int)(h->hist_pos - n);
      an.pend_h = (int)h->hist_keep;
    }
    rc = enqueue_stft(h, h->d_hist + (h->hist_pos - n), SG_PCM_F32, h->n_channels, n, 2LL * h->hist_keep, 1, d_dst, d_stride, n,
                      sg::OUT_ANALYSER | (pend ? sg::OUT_PEND : 0) | (logf ? sg::OUT_LOGFREQ : 0) | (u8_out ? sg::OUT_DB_U8 : sg::OUT_DB), 0, stream, &an);
    if (rc) return rc;
    h->dirty = false;
  } else {
    // nothing was pushed: an AnalyserNode hands out the same block's data again (it does not smooth twice);
    // re-emit from the state, which is only read
    SG_HIP(sg::launch_analyser_emit(h->d_state, u8_out ? nullptr : (float*)d_dst,
                                    u8_out ? (unsigned char*)d_dst : nullptr, h->n_channels, m, (int)len, d_stride,
                                    h->cfg.min_db, byte_mul, h->cfg.floor_db, ring_row, (long long)h->ring_h * row,
                                    logf ? h->d_lf : nullptr, row, stream));
  }
  // yoffset = (yoffset + 1) % TEXTURE_HEIGHT per draw (3D/visualizer.js:415); pinned to 0 outside the sonogram modes (:385-390)
  if (ring_row && !h->ring_pinned) h->ring_off = (h->ring_off + 1) % h->ring_h;
  if (!dev_out) {
    if (len > 0)
      SG_HIP(hipMemcpy2DAsync(u8_out ? (void*)u8_out : (void*)f32_out, stride * esz, d_dst, len * esz,
                              len * esz, h->n_channels, hipMemcpyDeviceToHost, stream));
    SG_HIP(hipStreamSynchronize(stream));
  }
  return SG_OK;
}

int sg_synchronize(sg_handle* h, void* hip_stream) {
  if (!h) return fail(SG_ERR_INVALID, "NULL handle");
  DeviceGuard g;
  int rc = g.enter(h->device);
  if (rc) return rc;
  SG_HIP(hipStreamSynchronize((hipStream_t)hip_stream));
  return SG_OK;
}

int sg_poll_u8(sg_handle* h, uint8_t* out, int64_t len, int64_t bins_stride, uint32_t flags, void* hip_stream) {
  return poll_freq(h, nullptr, out, len, bins_stride, flags, (hipStream_t)hip_stream);
}

// sg_push + one frequency poll.  When the block can be read where it lies (page-locked or device memory, no down-mix, at
// most one window of an even number of samples) the two are ONE launch: the transform takes the newest samples straight
// from the caller's block and files them in the history on its way.  Anything else is the two calls in sequence.
static int push_poll(sg_handle* h, const float* pcm, int64_t n_new, int64_t channel_stride, float* f32_out, uint8_t* u8_out,
                     int64_t len, int64_t bins_stride, uint32_t flags, void* hip_stream) {
  if (!h) return fail(SG_ERR_INVALID, "NULL handle");
  if (!h->d_hist) return fail(SG_ERR_STATE, "sg_push_poll before sg_stream_begin");
  const int fmt = h->cfg.pcm_format;
  const long long n = h->cfg.fft_size;
  const bool shape_ok = pcm && n_new > 0 && n_new <= n && (n_new % 2) == 0 && h->cfg.downmix <= 1 &&
                        (fmt == SG_PCM_F32 || fmt == SG_PCM_S16) && !h->no_pinned_direct &&
                        (channel_stride == 0 || channel_stride >= n_new);
  DeviceGuard g;  // page-locked buffers are looked up, and everything is enqueued, on the handle's device
  int rc = g.enter(h->device);
  if (rc) return rc;
  // Small pageable blocks (an analyser object polled from JS or Python: one channel, a few KB per animation frame) take
  // the one-launch path too, through page-locked staging the handle owns: copy in, ONE fused launch that reads the block
  // in place and stores the rows into page-locked memory, one synchronise, copy out -- instead of a staged push (copy +
  // kernel + synchronise) followed by a poll (kernel + copy + synchronise).
  constexpr size_t kSmall = 256 * 1024;
  const size_t ssz = fmt == SG_PCM_S16 ? 2 : 4;
  bool staged_in = false, staged_out = false;
  uint8_t* user_u8 = u8_out;
  float* user_f32 = f32_out;
  int64_t user_stride = bins_stride;
  long long out_l = 0;
  if (shape_ok && !(flags & (SG_IN_DEVICE | SG_IN_PINNED)) && (size_t)h->n_channels * (size_t)n_new * ssz <= kSmall) {
    const size_t need = (size_t)h->n_channels * (size_t)n_new * ssz;
    if (need > h->pin_in_cap) {
      if (h->pin_in) (void)hipHostFree(h->pin_in);
      h->pin_in = nullptr;
      h->pin_in_cap = 0;
      if (hipHostMalloc(&h->pin_in, need < 4096 ? 4096 : need, hipHostMallocDefault) == hipSuccess) h->pin_in_cap = need < 4096 ? 4096 : need;
      else (void)hipGetLastError();
    }
    if (h->pin_in_cap >= need) {
      const int64_t cs = channel_stride ? channel_stride : n_new;
      for (int c = 0; c < h->n_channels; ++c)
        std::memcpy((unsigned char*)h->pin_in + (size_t)c * n_new * ssz, (const unsigned char*)pcm + (size_t)c * cs * ssz, (size_t)n_new * ssz);
      pcm = (const float*)h->pin_in;
      channel_stride = n_new;
      flags |= SG_IN_PINNED;
      staged_in = true;
      // ... and the rows come back the same way when the caller's array is pageable too
      const long long row = poll_row_len(h);
      out_l = len > row ? row : len;
      const size_t esz = u8_out ? 1 : 4;
      const size_t oneed = (size_t)h->n_channels * (size_t)(out_l > 0 ? out_l : 0) * esz;
      if (!(flags & (SG_OUT_DEVICE | SG_OUT_PINNED)) && (u8_out || f32_out) && out_l > 0 && oneed <= kSmall &&
          (bins_stride == 0 || bins_stride >= out_l)) {
        if (oneed > h->pin_out_cap) {
          if (h->pin_out) (void)hipHostFree(h->pin_out);
          h->pin_out = nullptr;
          h->pin_out_cap = 0;
          if (hipHostMalloc(&h->pin_out, oneed < 4096 ? 4096 : oneed, hipHostMallocDefault) == hipSuccess) h->pin_out_cap = oneed < 4096 ? 4096 : oneed;
          else (void)hipGetLastError();
        }
        if (h->pin_out_cap >= oneed) {
          if (u8_out) u8_out = (uint8_t*)h->pin_out;
          else f32_out = (float*)h->pin_out;
          bins_stride = out_l;
          len = out_l;
          flags |= SG_OUT_PINNED;
          staged_out = true;
        }
      }
    }
  }
  auto finish_staged = [&](int code) -> int {
    if (!staged_in) return code;
    // the block was the handle's own staging: nothing may still read it when the next call refills it
    const hipError_t e = hipStreamSynchronize((hipStream_t)hip_stream);
    if (code) return code;
    if (e != hipSuccess) return fail(SG_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(e));
    if (staged_out) {
      const size_t esz = user_u8 ? 1 : 4;
      const int64_t us = user_stride ? user_stride : out_l;
      for (int c = 0; c < h->n_channels; ++c)
        std::memcpy((user_u8 ? (unsigned char*)user_u8 : (unsigned char*)user_f32) + (size_t)c * us * esz,
                    (const unsigned char*)h->pin_out + (size_t)c * out_l * esz, (size_t)out_l * esz);
    }
    return SG_OK;
  };
  const bool fusable = shape_ok && (flags & (SG_IN_DEVICE | SG_IN_PINNED)) != 0;
  const void* src = pcm;
  if (fusable && !(flags & SG_IN_DEVICE)) {
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, const_cast<float*>(pcm), 0) == hipSuccess && dp) src = dp;
    else {
      (void)hipGetLastError();
      src = nullptr;  // not mapped: the two-call path stages it
    }
  }
  if (!fusable || !src) {
    rc = sg_push(h, pcm, n_new, channel_stride, flags, hip_stream);
    if (rc) return finish_staged(rc);
    return finish_staged(poll_freq(h, f32_out, u8_out, len, bins_stride, flags, (hipStream_t)hip_stream));
  }
  // the poll's own argument checks, BEFORE the history pointer moves: a refused poll must not leave a block-sized hole
  if (len < 0) return finish_staged(fail(SG_ERR_INVALID, "negative len"));
  if (!f32_out && !u8_out) return finish_staged(fail(SG_ERR_INVALID, "NULL output"));
  {
    const long long row = poll_row_len(h), l = len > row ? row : len;
    if (bins_stride != 0 && bins_stride < l) return finish_staged(fail(SG_ERR_INVALID, "bins_stride < len"));
  }
  if (!(flags & SG_OUT_DEVICE) && (flags & SG_OUT_PINNED)) {
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, u8_out ? (void*)u8_out : (void*)f32_out, 0) != hipSuccess || !dp) {
      (void)hipGetLastError();
      return finish_staged(fail(SG_ERR_INVALID, "SG_OUT_PINNED: the output buffer is not page-locked, device-mapped host memory"));
    }
  }
  hipStream_t stream = (hipStream_t)hip_stream;
  hist_advance(h, n_new);  // the block's place in the history ring; the fused launch writes both copies there
  h->dirty = true;
  PendingBlock pb;
  pb.src = src;
  pb.stride = channel_stride ? channel_stride : n_new;
  pb.n_new = (int)n_new;
  pb.s16 = fmt == SG_PCM_S16;
  return finish_staged(poll_freq(h, f32_out, u8_out, len, bins_stride, flags, stream, &pb));
}

int sg_push_poll_u8(sg_handle* h, const float* pcm, int64_t n_new, int64_t channel_stride, uint8_t* out, int64_t len,
                    int64_t bins_stride, uint32_t flags, void* hip_stream) {
  return push_poll(h, pcm, n_new, channel_stride, nullptr, out, len, bins_stride, flags, hip_stream);
}

int sg_push_poll_f32(sg_handle* h, const float* pcm, int64_t n_new, int64_t channel_stride, float* out, int64_t len,
                     int64_t bins_stride, uint32_t flags, void* hip_stream) {
  return push_poll(h, pcm, n_new, channel_stride, out, nullptr, len, bins_stride, flags, hip_stream);
}

int sg_poll_f32(sg_handle* h, float* out, int64_t len, int64_t bins_stride, uint32_t flags, void* hip_stream) {
  return poll_freq(h, out, nullptr, len, bins_stride, flags, (hipStream_t)hip_stream);
}

}  // extern "C"

// the first `len` (<= N) of the latest N samples of every channel, as bytes or as floats
template <typename Tout>
static int poll_time(sg_handle* h, Tout* out, int64_t len, int64_t stride, uint32_t flags, hipStream_t stream) {
  if (!h) return fail(SG_ERR_INVALID, "NULL handle");
  if (!h->d_hist) return fail(SG_ERR_STATE, "poll before sg_stream_begin");
  if (!out) return fail(SG_ERR_INVALID, "NULL output");
  if (len < 0) return fail(SG_ERR_INVALID, "negative len");
  const long long n = h->cfg.fft_size;
  if (len > n) len = n;
  if (len == 0) return SG_OK;
  if (stride == 0) stride = len;
  if (stride < len) return fail(SG_ERR_INVALID, "stride < len");
  DeviceGuard g;
  int rc = g.enter(h->device);
  if (rc) return rc;
  const bool dev_out = (flags & SG_OUT_DEVICE) != 0;
  Tout* d_dst = out;
  long long d_stride = stride;
  if (!dev_out) {
    rc = h->poll_out.reserve((size_t)h->n_channels * len * sizeof(Tout) + 16);
    if (rc) return rc;
    d_dst = (Tout*)h->poll_out.p;
    d_stride = len;
  }
  const float in_scale = h->cfg.pcm_format == SG_PCM_S16 ? 1.0f / 32768.0f : h->cfg.pcm_format == SG_PCM_S24 ? 1.0f / 8388608.0f : 1.0f;
  if constexpr (sizeof(Tout) == 1) {
    SG_HIP(sg::launch_time_domain_u8(h->d_hist + (h->hist_pos - n), 2 * h->hist_keep, d_dst, h->n_channels, (int)len, d_stride, in_scale, stream));
  } else {
    SG_HIP(sg::launch_time_domain_f32(h->d_hist + (h->hist_pos - n), 2 * h->hist_keep, d_dst, h->n_channels, (int)len, d_stride, in_scale, stream));
  }
  if (!dev_out) {
    SG_HIP(hipMemcpy2DAsync(out, stride * sizeof(Tout), d_dst, len * sizeof(Tout), len * sizeof(Tout), h->n_channels,
                            hipMemcpyDeviceToHost, stream));
    SG_HIP(hipStreamSynchronize(stream));
  }
  return SG_OK;
}

extern "C" {

int sg_poll_time_u8(sg_handle* h, uint8_t* out, int64_t len, int64_t stride, uint32_t flags, void* hip_stream) {
  return poll_time(h, out, len, stride, flags, (hipStream_t)hip_stream);
}

int sg_poll_time_f32(sg_handle* h, float* out, int64_t len, int64_t stride, uint32_t flags, void* hip_stream) {
  return poll_time(h, out, len, stride, flags, (hipStream_t)hip_stream);
}

int sg_ring_begin(sg_handle* h, int32_t height) {
  if (!h) return fail(SG_ERR_INVALID, "NULL handle");
  if (!h->d_hist) return fail(SG_ERR_STATE, "sg_ring_begin before sg_stream_begin");
  if (height < 1 || height > 65536) return fail(SG_ERR_INVALID, "ring height must be in [1, 65536]");
  DeviceGuard g;
  int rc = g.enter(h->device);
  if (rc) return rc;
  if (h->d_ring) (void)hipFree(h->d_ring);
  h->d_ring = nullptr;
  const size_t bytes = (size_t)h->n_channels * height * (size_t)poll_row_len(h);  // rows of N/2 bins, or of logfreq_cols columns
  SG_HIP(hipMalloc((void**)&h->d_ring, bytes));
  SG_HIP(hipMemset(h->d_ring, 0, bytes));
  h->ring_h = height;
  h->ring_off = 0;
  h->ring_pinned = 0;
  return SG_OK;
}

int sg_ring_set_pinned(sg_handle* h, int32_t pinned) {
  if (!h) return fail(SG_ERR_INVALID, "NULL handle");
  if (!h->d_ring) return fail(SG_ERR_STATE, "sg_ring_set_pinned before sg_ring_begin");
  h->ring_pinned = pinned ? 1 : 0;
  if (pinned) h->ring_off = 0;
  return SG_OK;
}

int sg_alloc_pinned(size_t bytes, void** out) {
  if (!out) return fail(SG_ERR_INVALID, "NULL out");
  *out = nullptr;
  hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
  if (e != hipSuccess) return fail(SG_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  return SG_OK;
}

int sg_free_pinned(void* ptr) {
  if (!ptr) return SG_OK;
  SG_HIP(hipHostFree(ptr));
  return SG_OK;
}

int64_t sg_ring_bytes(const sg_handle* h) {
  if (!h || !h->d_ring) return fail(SG_ERR_STATE, "no ring");
  return (int64_t)h->n_channels * h->ring_h * poll_row_len(h);
}

int32_t sg_ring_offset(const sg_handle* h) {
  if (!h || !h->d_ring) return fail(SG_ERR_STATE, "no ring");
  return h->ring_off;
}

int sg_ring_read(sg_handle* h, uint8_t* out, int64_t channel_stride, uint32_t flags, void* hip_stream) {
  if (!h) return fail(SG_ERR_INVALID, "NULL handle");
  if (!h->d_ring) return fail(SG_ERR_STATE, "sg_ring_read before sg_ring_begin");
  if (!out) return fail(SG_ERR_INVALID, "NULL output");
  const size_t m = (size_t)poll_row_len(h), img = (size_t)h->ring_h * m;
  if (channel_stride == 0) channel_stride = (int64_t)img;
  if ((size_t)channel_stride < img && h->n_channels > 1) return fail(SG_ERR_INVALID, "channel_stride too small");
  hipStream_t stream = (hipStream_t)hip_stream;
  DeviceGuard g;
  int rc = g.enter(h->device);
  if (rc) return rc;
  const hipMemcpyKind kind = (flags & SG_OUT_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  // oldest row first: rows [off, H) then [0, off)
  const size_t tail = (size_t)(h->ring_h - h->ring_off) * m, head = (size_t)h->ring_off * m;
  SG_HIP(hipMemcpy2DAsync(out, channel_stride, h->d_ring + head, img, tail, h->n_channels, kind, stream));
  if (head)
    SG_HIP(hipMemcpy2DAsync(out + tail, channel_stride, h->d_ring, img, head, h->n_channels, kind, stream));
  if (!(flags & SG_OUT_DEVICE)) SG_HIP(hipStreamSynchronize(stream));
  return SG_OK;
}

int sg_last_launch(const sg_handle* h, sg_launch_info* out) {
  if (!h || !out) return fail(SG_ERR_INVALID, "NULL argument");
  *out = h->last;
  return SG_OK;
}

int sg_set_fft_size(sg_handle* h, int32_t fft_size) {
  if (!h) return fail(SG_ERR_INVALID, "NULL handle");
  sg_config c = h->cfg;
  c.fft_size = fft_size;
  if (h->d_hist) {
    if (c.history_samples == 0) c.history_samples = h->cfg.fft_size;  // what the live history retains
    if (fft_size > h->hist_keep)
      return fail(SG_ERR_INVALID, "fft_size %d exceeds the %lld samples of history this handle keeps (cfg.history_samples)",
                  fft_size, h->hist_keep);
  }
  int rc = check_config(c);
  if (rc) return rc;
  DeviceGuard g;
  rc = g.enter(h->device);
  if (rc) return rc;
  // Everything the new size needs is built ASIDE first -- tables, smoothing state, byte ring -- and swapped in only when
  // all of it exists: a failed allocation leaves the handle exactly as it was (a half-committed handle polled
  // with a null state pointer).
  Tables t;
  rc = make_tables(c, &t);
  if (rc) return rc;
  float* new_state = nullptr;
  unsigned char* new_ring = nullptr;
  const size_t m = (size_t)fft_size / 2;
  const size_t new_row = c.logfreq_cols > 0 ? (size_t)c.logfreq_cols : m;
  auto undo = [&](int code) {
    t.release();
    if (new_state) (void)hipFree(new_state);
    if (new_ring) (void)hipFree(new_ring);
    return code;
  };
  if (h->d_hist) {
    // the smoothing state and the byte ring's rows have fftSize / 2 entries: both start from zero; the samples stay
    hipError_t e = hipMalloc((void**)&new_state, sizeof(float) * m * h->n_channels);
    if (e == hipSuccess) e = hipMemset(new_state, 0, sizeof(float) * m * h->n_channels);
    if (e == hipSuccess && h->d_ring) {
      const size_t bytes = (size_t)h->n_channels * h->ring_h * new_row;
      e = hipMalloc((void**)&new_ring, bytes);
      if (e == hipSuccess) e = hipMemset(new_ring, 0, bytes);
    }
    if (e != hipSuccess)
      return undo(fail(SG_ERR_NOMEM, "sg_set_fft_size(%d): %s; the handle keeps fft_size %d", fft_size, hipGetErrorString(e),
                       h->cfg.fft_size));
  }
  {
    const hipError_t e = hipDeviceSynchronize();  // nothing in flight reads the tables or the state that go away
    if (e != hipSuccess) return undo(fail(SG_ERR_HIP, "hipDeviceSynchronize failed: %s", hipGetErrorString(e)));
  }
  // commit: nothing below can fail
  if (!h->d_hist) c.history_samples = h->cfg.history_samples;
  Tables old{h->d_tw, h->d_win, h->d_lf, h->d_mixw};
  old.release();
  h->d_tw = t.d_tw;
  h->d_win = t.d_win;
  h->d_lf = t.d_lf;
  h->d_mixw = t.d_mixw;
  h->cfg = c;
  h->log2n = ilog2(fft_size);
  if (h->d_hist) {
    if (h->d_state) (void)hipFree(h->d_state);
    h->d_state = new_state;
    if (h->d_ring) {
      (void)hipFree(h->d_ring);
      h->d_ring = new_ring;
      h->ring_off = 0;  // (height and the pinned flag stay)
    }
    h->dirty = true;
  }
  return SG_OK;
}

int64_t sg_debug_twiddle_triplets(int32_t fft_size, float* out, int64_t cap) {
  if (fft_size < 32 || fft_size > 32768 || (fft_size & (fft_size - 1))) return fail(SG_ERR_INVALID, "fft_size must be a power of two in [32, 32768]");
  if (cap < 0 || (cap > 0 && !out)) return fail(SG_ERR_INVALID, "bad output buffer");
  std::vector<float> t;
  if (fft_size >= 4096) derived_triplets(fft_size, t);
  const int64_t n = (int64_t)t.size() < cap ? (int64_t)t.size() : cap;
  if (n > 0) std::memcpy(out, t.data(), sizeof(float) * (size_t)n);
  return (int64_t)t.size();
}

int sg_debug_set(sg_handle* h, const char* key, int64_t value) {
  if (!h || !key) return fail(SG_ERR_INVALID, "NULL argument");
  if (!std::strcmp(key, "run_len")) {
    if (value < 0 || value > 0x7fffffff) return fail(SG_ERR_INVALID, "run_len must be >= 0");
    h->run_len_override = (int)value;
  } else if (!std::strcmp(key, "no_lring")) h->no_lring = value != 0;
  else if (!std::strcmp(key, "lring_all")) h->lring_all = value != 0;
  else if (!std::strcmp(key, "no_single_round")) h->no_single_round = value != 0;
  else if (!std::strcmp(key, "no_fused_smoothing")) h->no_fused_smoothing = value != 0;
  else if (!std::strcmp(key, "no_pinned_direct")) h->no_pinned_direct = value != 0;
  else if (!std::strcmp(key, "pinned_direct_generic")) h->pinned_direct_generic = value != 0;
  else if (!std::strcmp(key, "no_lffast")) h->no_lffast = value != 0;
  else if (!std::strcmp(key, "no_gen_unit")) h->no_gen_unit = value != 0;
  else if (!std::strcmp(key, "extra_lds")) {
    if (value < 0 || value > 160 * 1024) return fail(SG_ERR_INVALID, "extra_lds must be in [0, 163840]");
    h->extra_lds = (int)value;
  }
  else return fail(SG_ERR_INVALID, "sg_debug_set: unknown key \"%s\"", key);
  return SG_OK;
}

int sg_set_smoothing(sg_handle* h, float tau) {
  if (!h) return fail(SG_ERR_INVALID, "NULL handle");
  if (!(tau >= 0.f && tau <= 1.f)) return fail(SG_ERR_INVALID, "smoothing must be in [0, 1]");
  h->cfg.smoothing = tau;
  return SG_OK;
}

int sg_set_db_range(sg_handle* h, float min_db, float max_db) {
  if (!h) return fail(SG_ERR_INVALID, "NULL handle");
  if (!(max_db > min_db)) return fail(SG_ERR_INVALID, "max_db must exceed min_db");
  h->cfg.min_db = min_db;
  h->cfg.max_db = max_db;
  return SG_OK;
}

}  // extern "C"
