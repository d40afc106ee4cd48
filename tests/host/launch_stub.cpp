// launch_stub.cpp -- TEST INFRASTRUCTURE ONLY: host stand-ins for the kernel launchers the host side of libspectro.so calls
// (csrc/launch.hpp: sg::launch_stft_<log2n>, launch_push_mix, launch_analyser_emit, launch_smooth_scan,
// launch_time_domain_*), used with tests/host/hip_stub.cpp by `make -C oracle check-shim-asan`.
//
// No transform here.  Each launcher does what the real kernel does to MEMORY, on the host, so that AddressSanitizer sees
// every address the host side hands to a kernel:
//   * the transform launchers check the launch geometry spectro_api.hip computed (grid covers the runs, runs cover the
//     frames, the LDS-ring extent covers the last window), read the first and last sample of every frame's window in the
//     caller's format, and fill every output row / smoothing-state row / history-ring row they would write with the
//     launch's sequence number -- so that the drivers can tell WHICH poll wrote a row;
//   * the push, time-domain and fused push + poll paths are emulated exactly (the double-written history ring: sample L at
//     L mod H and L mod H + H), so that a sequence of ragged pushes can be read back sample for sample.
// Compiled by hipcc --cuda-host-only (it needs csrc/launch.hpp's types), never linked into the product.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../spectrogram_amd/csrc/launch.hpp"

extern "C" long sg_stub_bytes_from(const void* p);   // hip_stub.cpp: bytes to the end of the block p lies in, -1 if none
extern "C" int sg_stub_owner_of(const void* p);      // ... the device that block belongs to (-1: page-locked / not a device block)
extern "C" int sg_stub_current_device(void);

namespace {
long g_seq = 0;        // transform launches so far
long g_violations = 0;

void violation(const char* what) {
  std::fprintf(stderr, "launch_stub: %s\n", what);
  ++g_violations;
}
#define REQUIRE(cond) do { if (!(cond)) violation(#cond); } while (0)
// a kernel runs on the device that is current at its launch and no peer access is enabled: every device pointer it is handed
// must belong to that device (page-locked host memory, owner -1, is mapped everywhere)
#define ON_THIS_DEVICE(ptr) do { if (ptr) { const int o_ = sg_stub_owner_of(ptr); if (o_ >= 0 && o_ != sg_stub_current_device()) violation(#ptr " belongs to another device than the launch"); } } while (0)

inline void touch_read(const void* p, size_t bytes) {   // ASan flags it if [p, p + bytes) is not inside a live block
  if (!bytes) return;
  volatile unsigned char a = ((const volatile unsigned char*)p)[0];
  volatile unsigned char b = ((const volatile unsigned char*)p)[bytes - 1];
  (void)a; (void)b;
}

float sample_at(const void* base, int fmt, long long idx) {
  if (fmt == 1) return (float)((const short*)base)[idx];
  if (fmt == 2) {
    const unsigned char* b = (const unsigned char*)base + 3 * idx;
    return (float)((int)(signed char)b[2] * 65536 + ((int)b[1] << 8) + (int)b[0]);
  }
  return ((const float*)base)[idx];
}

hipError_t stft_effects(int log2n, int shift_code, int out, const sg::StftParams& p, long long nblocks) {
  const int n = 1 << log2n, m = n / 2;
  const int fmt = (shift_code >> 3) & 3, shift = shift_code & 7;
  const size_t ssz = fmt == 1 ? 2 : fmt == 2 ? 3 : 4;
  const bool analyser = (out & sg::OUT_ANALYSER) != 0, logf = (out & sg::OUT_LOGFREQ) != 0, u8 = (out & 3) == sg::OUT_DB_U8;
  const size_t esz = u8 ? 1 : 4;
  ++g_seq;
  ON_THIS_DEVICE(p.pcm); ON_THIS_DEVICE(p.out); ON_THIS_DEVICE(p.tw); ON_THIS_DEVICE(p.win); ON_THIS_DEVICE(p.lf_tab);
  ON_THIS_DEVICE(p.an_state); ON_THIS_DEVICE(p.an_ring); ON_THIS_DEVICE(p.pend_src); ON_THIS_DEVICE(p.downmix_w);
  // launch geometry (spectro_api.hip, enqueue_stft)
  REQUIRE(p.run_len >= 1 && p.runs_per_channel * (long long)p.run_len >= p.frames);
  REQUIRE((p.runs_per_channel - 1) * (long long)p.run_len < p.frames);
  REQUIRE(p.total_runs == p.runs_per_channel * p.n_channels);
  REQUIRE(nblocks * sg::slots_per_block(log2n) >= p.total_runs && (nblocks - 1) * sg::slots_per_block(log2n) < p.total_runs);
  REQUIRE(p.tw && p.win && sg_stub_bytes_from(p.tw) >= (long)(sizeof(float) * 2 * n) && sg_stub_bytes_from(p.win) >= (long)(sizeof(float) * 2 * m));
  REQUIRE(!logf || (p.lf_tab && p.lf_cols > 0 && sg_stub_bytes_from(p.lf_tab) >= 8L * p.lf_cols));
  REQUIRE((out & sg::OUT_SMOOTH) == 0 || p.run_len == p.frames);
  if (shift == 2 || shift == 4) REQUIRE(p.hop == shift * (n / 8) && p.sample_stride == 1 && p.downmix <= 1);
  if (shift == 1 || shift == 3) {
    REQUIRE(p.sample_stride == 1 && p.downmix <= 1 && p.hop <= n - 4);
    REQUIRE(p.in_extent >= (p.n_channels - 1) * p.channel_stride + (p.frames - 1) * (long long)p.hop + n);
  }
  if (p.downmix > 1 && p.downmix_w) REQUIRE(sg_stub_bytes_from(p.downmix_w) >= 4L * p.downmix);
  const int dmx = p.downmix > 1 ? p.downmix : 1;
  const int rowlen = logf ? p.lf_cols : p.bins;
  for (long long c = 0; c < p.n_channels; ++c) {
    const long long ch0 = c * p.channel_stride;
    for (long long f = 0; f < p.frames; ++f) {
      // the window: samples (f * hop + i) * dmx + k of the channel, i < N, k < dmx, sample_stride apart
      const long long first = ch0 + f * (long long)p.hop * dmx * p.sample_stride;
      const long long last = first + ((long long)n * dmx - 1) * p.sample_stride;
      if (p.pend_src && analyser) {
        // fused push + poll: samples [pend_split, N) come from the caller's block and are filed in the history ring
        REQUIRE(p.pend_split >= 0 && p.pend_split < n && (p.pend_split % 2) == 0);
        float* row = const_cast<float*>((const float*)p.pcm) + ch0 - p.pend_wstart;
        const int n_new = n - p.pend_split;
        const int h = p.pend_h;
        if (p.pend_split > 0) touch_read((const float*)p.pcm + ch0, 4 * (size_t)p.pend_split);
        for (int i = 0; i < n_new; ++i) {
          const float v = p.pend_s16 ? (float)((const short*)p.pend_src)[c * p.pend_stride + i] : ((const float*)p.pend_src)[c * p.pend_stride + i];
          long long q = (long long)p.pend_wstart + p.pend_split + i, t = q + h;
          if (t >= 2LL * h) t -= 2LL * h;
          row[q] = v;
          row[t] = v;
        }
      } else {
        touch_read((const unsigned char*)p.pcm + (size_t)first * ssz, ssz);
        touch_read((const unsigned char*)p.pcm + (size_t)last * ssz, ssz);
      }
      if (!analyser) {
        unsigned char* o = (unsigned char*)p.out + (size_t)(c * p.out_channel_stride + f * rowlen) * esz;
        std::memset(o, (int)(g_seq & 0xff), (size_t)rowlen * esz);
      }
    }
    if (analyser) {
      REQUIRE(p.frames == 1 && p.an_state);
      float* st = p.an_state + c * (long long)m;
      touch_read(st, 4 * (size_t)m);
      for (int k = 0; k < m; ++k) st[k] = (float)g_seq;
      if (p.an_ring) std::memset(p.an_ring + c * p.an_ring_stride, (int)(g_seq & 0xff), (size_t)(logf ? p.lf_cols : m));
      REQUIRE(p.an_len >= 0 && p.an_len <= (logf ? p.lf_cols : m));
      unsigned char* o = (unsigned char*)p.out + (size_t)(c * p.out_channel_stride) * esz;
      if (u8) std::memset(o, (int)(g_seq & 0xff), (size_t)p.an_len);
      else for (int k = 0; k < p.an_len; ++k) ((float*)o)[k] = (float)g_seq;
    }
  }
  return hipSuccess;
}
}  // namespace

extern "C" long sg_stub_launch_seq(void) { return g_seq; }
extern "C" long sg_stub_violations(void) { return g_violations; }

namespace sg {

#define SG_STUB(n) \
  hipError_t launch_stft_##n(int shift, int out, const StftParams& p, long long nblocks, hipStream_t) { return stft_effects(n, shift, out, p, nblocks); }
SG_STUB(5) SG_STUB(6) SG_STUB(7) SG_STUB(8) SG_STUB(9) SG_STUB(10) SG_STUB(11) SG_STUB(12) SG_STUB(13) SG_STUB(14) SG_STUB(15)
#undef SG_STUB

// exact: the double-written history ring (csrc/aux_kernels.hip push_mix_kernel / push_copy4_kernel)
hipError_t launch_push_mix(const void* src, int fmt, long long src_stride, float* dst, long long dst_stride, int n_channels,
                           int n_new, int k, int a0, int hist_h, const float* w, hipStream_t) {
  REQUIRE(a0 >= 0 && a0 < hist_h && n_new >= 1 && n_new <= hist_h && k >= 1);
  ON_THIS_DEVICE(src); ON_THIS_DEVICE(dst); ON_THIS_DEVICE(w);
  REQUIRE(sg_stub_bytes_from(dst) >= (long)(4LL * ((n_channels - 1) * dst_stride + 2LL * hist_h)));
  for (int c = 0; c < n_channels; ++c)
    for (int t = 0; t < n_new; ++t) {
      float a = 0.f;
      for (int j = 0; j < k; ++j) {
        const float s = sample_at(src, fmt, c * src_stride + (long long)t * k + j);
        a = (w && k > 1) ? __builtin_fmaf(w[j], s, a) : a + s;
      }
      const float v = (k > 1 && !w) ? a * (1.0f / (float)k) : a;
      int q = a0 + t;
      if (q >= hist_h) q -= hist_h;
      dst[c * dst_stride + q] = v;
      dst[c * dst_stride + q + hist_h] = v;
    }
  return hipSuccess;
}

hipError_t launch_analyser_emit(const float* state, float* db_out, unsigned char* u8_out, int n_channels, int bins, int len,
                                long long out_stride, float, float, float, unsigned char* ring_row, long long ring_stride,
                                const uint2* lf_tab, int row, hipStream_t) {
  REQUIRE(len >= 0 && len <= row && (lf_tab != nullptr) == (row != bins));
  ON_THIS_DEVICE(state); ON_THIS_DEVICE(db_out); ON_THIS_DEVICE(u8_out); ON_THIS_DEVICE(ring_row); ON_THIS_DEVICE(lf_tab);
  for (int c = 0; c < n_channels; ++c) {
    touch_read(state + (long long)c * bins, 4 * (size_t)bins);
    const float seq = state[(long long)c * bins];   // the transform stub left its sequence number there
    if (ring_row) std::memset(ring_row + c * ring_stride, (int)((long)seq & 0xff), (size_t)row);
    if (u8_out) std::memset(u8_out + c * out_stride, (int)((long)seq & 0xff), (size_t)len);
    if (db_out) for (int k = 0; k < len; ++k) db_out[c * out_stride + k] = seq;
  }
  return hipSuccess;
}

hipError_t launch_smooth_scan(const float* mag, void* out, int out_kind, long long n_channels, long long frames, int bins,
                              long long out_channel_stride, float, float, float, float, const uint2* lf_tab, int cols, hipStream_t) {
  const int row = lf_tab ? cols : bins;
  const size_t esz = out_kind == OUT_DB_U8 ? 1 : 4;
  ON_THIS_DEVICE(mag); ON_THIS_DEVICE(out); ON_THIS_DEVICE(lf_tab);
  touch_read(mag, 4 * (size_t)(n_channels * frames * bins));
  for (long long c = 0; c < n_channels; ++c)
    std::memset((unsigned char*)out + (size_t)(c * out_channel_stride) * esz, 0x5A, (size_t)(frames * row) * esz);
  return hipSuccess;
}

// exact: clip(floor(128 (1 + x)), 0, 255) / x of the first `len` window samples (csrc/aux_kernels.hip time_domain_kernel)
hipError_t launch_time_domain_u8(const float* hist, long long hist_stride, unsigned char* out, int n_channels, int len,
                                 long long out_stride, float in_scale, hipStream_t) {
  ON_THIS_DEVICE(hist); ON_THIS_DEVICE(out);
  for (int c = 0; c < n_channels; ++c)
    for (int k = 0; k < len; ++k) {
      const float v = 128.f + __builtin_floorf(128.f * in_scale * hist[c * hist_stride + k]);
      out[c * out_stride + k] = (unsigned char)(v < 0.f ? 0.f : (v > 255.f ? 255.f : v));
    }
  return hipSuccess;
}
hipError_t launch_time_domain_f32(const float* hist, long long hist_stride, float* out, int n_channels, int len,
                                  long long out_stride, float in_scale, hipStream_t) {
  ON_THIS_DEVICE(hist); ON_THIS_DEVICE(out);
  for (int c = 0; c < n_channels; ++c)
    for (int k = 0; k < len; ++k) out[c * out_stride + k] = in_scale * hist[c * hist_stride + k];
  return hipSuccess;
}

}  // namespace sg
