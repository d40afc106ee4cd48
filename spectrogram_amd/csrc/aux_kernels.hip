// aux_kernels.hip -- the small kernels around the STFT hot path (poll mode).
#include "launch.hpp"
#include <cmath>

namespace sg {

// A frequency poll with no sg_push since the last one: an AnalyserNode hands out the same block's data again (it does
// not smooth twice), so the row is re-emitted from the smoothing state as it stands -- dB + byte map, one thread per
// (channel, row element), the state only read (3D/visualizer.js:352,358 may poll float and byte data in one draw).
// With a column table the row is the log-frequency columns of the bins (the same blend of floored byte / clamped dB
// texels as the fused epilogue in stft_kernel.hpp).
__global__ void analyser_emit_kernel(const float* __restrict__ state, float* __restrict__ db_out,
                                     unsigned char* __restrict__ u8_out, int n_channels, int bins, int len,
                                     long long out_stride, float min_db, float byte_mul, float floor_db,
                                     unsigned char* __restrict__ ring_row, long long ring_stride,
                                     const uint2* __restrict__ lf_tab, int row) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)n_channels * row) return;
  const int c = (int)(i / row), k = (int)(i % row);
  const float* st = state + (long long)c * bins;
  auto db = [&](int b) { return 6.02059991327962f * __builtin_amdgcn_logf(st[b]); };  // 20 log10 = 20 log10(2) * log2
  auto byte = [&](float y) { return fminf(fmaxf(__builtin_floorf((y - min_db) * byte_mul), 0.f), 255.f); };
  float y;
  unsigned char bq;
  if (lf_tab) {
    const uint2 e = lf_tab[k];  // {byte offsets of the two bins in a float row, blend fraction}
    const int i0 = (int)(e.x & 0xffffu) >> 2, i1 = (int)(e.x >> 16) >> 2;
    const float f = __uint_as_float(e.y);
    const float ya = fmaxf(db(i0), floor_db), yb = fmaxf(db(i1), floor_db);
    y = __builtin_fmaf(f, yb, __builtin_fmaf(-f, ya, ya));   // a (1 - f) + b f: the fused epilogue's formula, bit for bit
    const float ba = byte(db(i0)), bb = byte(db(i1));
    bq = (unsigned char)(__builtin_fmaf(f, bb, __builtin_fmaf(-f, ba, ba)) + 0.5f);
  } else {
    y = db(k);
    bq = (unsigned char)byte(y);
  }
  // history ring row (the reference's texture row at yoffset): always the full row
  if (ring_row) ring_row[c * ring_stride + k] = bq;
  if (k >= len) return;
  if (db_out) db_out[c * out_stride + k] = y;
  if (u8_out) u8_out[c * out_stride + k] = bq;
}

hipError_t launch_analyser_emit(const float* state, float* db_out, unsigned char* u8_out, int n_channels, int bins,
                                int len, long long out_stride, float min_db, float byte_mul, float floor_db,
                                unsigned char* ring_row, long long ring_stride, const uint2* lf_tab, int row,
                                hipStream_t s) {
  const long long n = (long long)n_channels * row;
  const unsigned nb = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(analyser_emit_kernel, dim3(nb), dim3(256), 0, s, state, db_out, u8_out, n_channels, bins, len,
                     out_stride, min_db, byte_mul, floor_db, ring_row, ring_stride, lf_tab, row);
  return hipGetLastError();
}

// sg_push on its way into the float history: cfg.downmix = K interleaved input samples averaged into one (the reference's
// mic / file sources reach the analyser through `mix`, a GainNode that sums whatever channels the source has:
// UI/player.js:25,110-115,125-131) -- same order of additions and the same 1/K as the batch kernel's fused mix -- and
// int16 input (a 16-bit capture device: half the bytes over PCIe) stored as the float of its integer value, the 1/32768
// being folded into the window exactly as in the batch path.
//
// The history row of a channel is a DOUBLE-WRITTEN ring of H samples: 2 H floats, sample number L stored at L mod H and
// at L mod H + H.  Every run of up to H consecutive samples therefore lies contiguous somewhere in the row (the newest N
// end at ((L_end - 1) mod H) + 1 + H), the transform reads its window as a plain strided batch, and nothing is ever
// moved: rounds 2-4 kept rows of 3 H floats with an append pointer and copied the newest H - take samples to the front
// whenever the pointer reached the end (65 MB through a device-to-device copy every 64th hop with 512 channels of 32768
// samples: ADVICE r4).  `a0` = the ring position (< H) of the first new sample.
// Tin = s24: packed little-endian 24-bit samples (three bytes each, no alignment), stored as the float of the integer like
// int16 (the window carries the 1 / 8388608).  `w` = null: the mean; else K weights (the fused batch mix's order of FMAs).
template <typename Tin>
__global__ void push_mix_kernel(const Tin* __restrict__ src, long long src_stride, float* __restrict__ dst,
                                long long dst_stride, int n_channels, int n_new, int k, float scale, int a0, int hist_h,
                                const float* __restrict__ w) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)n_channels * n_new) return;
  const int c = (int)(i / n_new), t = (int)(i % n_new);
  const Tin* s = src + c * src_stride + (long long)t * k;
  float a = 0.f;
  if (w && k > 1) {
    for (int j = 0; j < k; ++j) a = __builtin_fmaf(w[j], (float)s[j], a);
  } else {
    for (int j = 0; j < k; ++j) a += (float)s[j];
  }
  const float v = (k > 1 && !w) ? a * scale : a;
  int q = a0 + t;               // a0 < H, t < n_new <= H
  if (q >= hist_h) q -= hist_h;
  float* const row = dst + c * dst_stride;
  row[q] = v;
  row[q + hist_h] = v;
}

// the common push (K = 1, a multiple of four samples): four samples per thread, 16-byte (float32) / 8-byte (int16)
// loads -- the shape PCIe reads of page-locked host memory want; both copies leave as one 16-byte store each unless the
// four samples straddle the ring's end
template <typename Tin>
__global__ void push_copy4_kernel(const Tin* __restrict__ src, long long src_stride, float* __restrict__ dst,
                                  long long dst_stride, int n_channels, int quads, int a0, int hist_h) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)n_channels * quads) return;
  const int c = (int)(i / quads), q = (int)(i % quads);
  const Tin* s = src + c * src_stride + 4LL * q;
  f4 v;
  if constexpr (sizeof(Tin) == 4) {
    typedef f4 f4_a4 __attribute__((aligned(4)));
    v = *reinterpret_cast<const f4_a4*>(s);
  } else {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    typedef u2 u2_a2 __attribute__((aligned(2)));
    const u2 r = *reinterpret_cast<const u2_a2*>(s);
    v = f4{(float)(short)(r.x & 0xffffu), (float)((int)r.x >> 16), (float)(short)(r.y & 0xffffu), (float)((int)r.y >> 16)};
  }
  typedef f4 f4_a4s __attribute__((aligned(4)));
  float* const row = dst + c * dst_stride;
  int pos = a0 + 4 * q;
  if (pos >= hist_h) pos -= hist_h;
  if (pos + 4 <= hist_h) {
    *reinterpret_cast<f4_a4s*>(row + pos) = v;
    *reinterpret_cast<f4_a4s*>(row + pos + hist_h) = v;
  } else {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int w = pos + u;
      if (w >= hist_h) w -= hist_h;
      row[w] = v[u];
      row[w + hist_h] = v[u];
    }
  }
}

hipError_t launch_push_mix(const void* src, int fmt, long long src_stride, float* dst, long long dst_stride, int n_channels,
                           int n_new, int k, int a0, int hist_h, const float* w, hipStream_t s) {
  if (fmt == 2) {  // packed int24: the general kernel (byte loads)
    const long long n24 = (long long)n_channels * n_new;
    hipLaunchKernelGGL(push_mix_kernel<s24>, dim3((unsigned)((n24 + 255) / 256)), dim3(256), 0, s, (const s24*)src, src_stride, dst,
                       dst_stride, n_channels, n_new, k, 1.0f / (float)k, a0, hist_h, w);
    return hipGetLastError();
  }
  if (k == 1 && n_new % 4 == 0 && n_new > 0) {
    const long long n4 = (long long)n_channels * (n_new / 4);
    const unsigned nb4 = (unsigned)((n4 + 255) / 256);
    if (fmt == 1)
      hipLaunchKernelGGL(push_copy4_kernel<short>, dim3(nb4), dim3(256), 0, s, (const short*)src, src_stride, dst, dst_stride,
                         n_channels, n_new / 4, a0, hist_h);
    else
      hipLaunchKernelGGL(push_copy4_kernel<float>, dim3(nb4), dim3(256), 0, s, (const float*)src, src_stride, dst, dst_stride,
                         n_channels, n_new / 4, a0, hist_h);
    return hipGetLastError();
  }
  const long long n = (long long)n_channels * n_new;
  const unsigned nb = (unsigned)((n + 255) / 256);
  if (fmt == 1)
    hipLaunchKernelGGL(push_mix_kernel<short>, dim3(nb), dim3(256), 0, s, (const short*)src, src_stride, dst, dst_stride,
                       n_channels, n_new, k, 1.0f / (float)k, a0, hist_h, w);
  else
    hipLaunchKernelGGL(push_mix_kernel<float>, dim3(nb), dim3(256), 0, s, (const float*)src, src_stride, dst, dst_stride,
                       n_channels, n_new, k, 1.0f / (float)k, a0, hist_h, w);
  return hipGetLastError();
}

// Batch-mode "smoothing over time" (AnalyserNode, tau != 0; the reference's bar-graph mode uses
// 0.75, 3D/visualizer.js:351): X^[f][k] = tau X^[f-1][k] + (1 - tau) |X[f][k]| along the frames of
// one channel, then the configured output kind.
//
// The recurrence forgets: after W steps the state it started from is scaled by tau^W.  So the frame
// axis is cut into chunks of `chunk` frames, and the thread of (channel, bin, chunk j) starts `warm`
// frames before its chunk from a zero state; with tau^warm < 2^-30 what it then writes equals the
// serial scan to float32 rounding (chunk 0 starts at frame 0 exactly as the serial scan does).
// 64 consecutive bins per wave: every step is one coalesced 256-byte row segment; reads are
// (chunk + warm) / chunk of the |X| buffer, writes each output once.
// With a column table (`lf_tab`, `cols` > 0) a thread owns one log-frequency COLUMN instead of one bin: it carries the
// recurrences of the column's two neighbouring bins and blends their texels -- the configured kind's value of the smoothed
// magnitude, floored bytes for SG_OUT_DB_U8 -- exactly as the fused epilogue does (stft_kernel.hpp, LOGF): the reference
// smooths in the analyser (3D/visualizer.js:351) and samples the row it hands out in the shader (sonogram-vertex.shader:51-52).
__global__ __launch_bounds__(64) void smooth_scan_kernel(const float* __restrict__ mag, void* __restrict__ out, int out_kind,
                                                         long long n_channels, long long frames, int bins,
                                                         long long out_channel_stride, int chunk, int warm, int bin_blocks, float tau,
                                                         float floor_db, float min_db, float byte_mul,
                                                         const uint2* __restrict__ lf_tab, int cols) {
  const int kb = (int)(blockIdx.x % (unsigned)bin_blocks);
  const long long j = blockIdx.x / (unsigned)bin_blocks;  // chunk index
  const int k = kb * 64 + (int)threadIdx.x;
  const int row = lf_tab ? cols : bins;                    // elements of an output row
  if (k >= row) return;
  int k0 = k, k1 = k;
  float fr = 0.f;
  if (lf_tab) {
    const uint2 e = lf_tab[k];  // {byte offsets of the two bins in a float row, blend fraction}
    k0 = (int)(e.x & 0xffffu) >> 2;
    k1 = (int)(e.x >> 16) >> 2;
    fr = __uint_as_float(e.y);
  }
  const float one_m_tau = 1.0f - tau;
  auto value = [&](float sm) -> float {   // the texel / the output of one smoothed magnitude (bytes: floored, not yet converted)
    if (out_kind == OUT_MAG) return sm;
    if (out_kind == OUT_POWER) return sm * sm;
    const float y = fmaxf(6.02059991327962f * __builtin_amdgcn_logf(sm), floor_db);
    if (out_kind == OUT_DB) return y;
    return fminf(fmaxf(__builtin_floorf((y - min_db) * byte_mul), 0.f), 255.f);
  };
  for (long long c = blockIdx.y; c < n_channels; c += gridDim.y) {
  const long long f_begin = j * (long long)chunk;
  long long f_end = f_begin + chunk;
  if (f_end > frames) f_end = frames;
  long long f = f_begin - warm;
  if (f < 0) f = 0;
  const float* m = mag + (c * frames + f) * bins;
  float* of = reinterpret_cast<float*>(out) + c * out_channel_stride + f_begin * row + k;
  unsigned char* ob = reinterpret_cast<unsigned char*>(out) + c * out_channel_stride + f_begin * row + k;
  float sm0 = 0.f, sm1 = 0.f;
  auto advance = [&]() {
    sm0 = tau * sm0 + one_m_tau * m[k0];
    if (!(fabsf(sm0) < __builtin_inff())) sm0 = 0.f;
    if (lf_tab) {
      sm1 = tau * sm1 + one_m_tau * m[k1];
      if (!(fabsf(sm1) < __builtin_inff())) sm1 = 0.f;
    }
  };
  for (; f < f_begin; ++f, m += bins) advance();  // warm-up: state only
  for (; f < f_end; ++f, m += bins, of += row, ob += row) {
    advance();
    float v = value(sm0);
    if (lf_tab) {
      const float b = value(sm1);
      v = __builtin_fmaf(fr, b, __builtin_fmaf(-fr, v, v));   // a (1 - f) + b f: the fused epilogue's formula
      if (out_kind == OUT_DB_U8) v = (float)(unsigned char)(v + 0.5f);
    }
    if (out_kind == OUT_DB_U8) *ob = (unsigned char)v;
    else *of = v;
  }
  }  // channels
}

hipError_t launch_smooth_scan(const float* mag, void* out, int out_kind, long long n_channels, long long frames,
                              int bins, long long out_channel_stride, float tau, float floor_db, float min_db,
                              float byte_mul, const uint2* lf_tab, int cols, hipStream_t s) {
  // warm-up length: tau^warm < 2^-30; tau close to 1 remembers for longer than the signal: one chunk (serial, exact)
  long long warm = frames, chunk = frames;
  if (tau < 0.999999f) {
    const double w = 30.0 * 0.6931471805599453 / -std::log((double)tau);
    if (w < (double)frames / 2) {
      warm = (long long)w + 1;
      chunk = 4 * warm;            // 1.25x read overhead
      if (chunk < 64) chunk = 64;
    }
  }
  const int bin_blocks = ((lf_tab ? cols : bins) + 63) / 64;
  const long long chunks = (frames + chunk - 1) / chunk;
  if (chunks * bin_blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(smooth_scan_kernel, dim3((unsigned)(chunks * bin_blocks), (unsigned)(n_channels < 65535 ? n_channels : 65535)),
                     dim3(64), 0, s, mag, out, out_kind, n_channels, frames, bins, out_channel_stride, (int)chunk, (int)warm, bin_blocks, tau, floor_db, min_db,
                     byte_mul, lf_tab, cols);
  return hipGetLastError();
}

// getByteTimeDomainData: b = clip(floor(128 (1 + x)), 0, 255) (3D/visualizer.js:363); getFloatTimeDomainData: x
template <typename Tout>
__global__ void time_domain_kernel(const float* __restrict__ hist, long long hist_stride,
                                   Tout* __restrict__ out, int n_channels, int len,
                                   long long out_stride, float in_scale) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)n_channels * len) return;
  const int c = (int)(i / len), k = (int)(i % len);
  const float x = in_scale * hist[c * hist_stride + k];  // in_scale: 1, or 1/32768 for int16 pushes
  if constexpr (sizeof(Tout) == 1) {
    // floor(128 (1 + x)) of the REAL number: 1 + x rounds in float32 (x = -1e-9 would give 128, not 127); 128 x is exact,
    // so is its floor, so is adding the integer 128
    const float v = 128.f + __builtin_floorf(128.f * x);
    out[c * out_stride + k] = (unsigned char)fminf(fmaxf(v, 0.f), 255.f);
  } else {
    out[c * out_stride + k] = x;
  }
}

template <typename Tout>
static hipError_t launch_time_domain(const float* hist, long long hist_stride, Tout* out,
                                     int n_channels, int len, long long out_stride, float in_scale, hipStream_t s) {
  const long long n = (long long)n_channels * len;
  const unsigned nb = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(time_domain_kernel<Tout>, dim3(nb), dim3(256), 0, s, hist, hist_stride, out,
                     n_channels, len, out_stride, in_scale);
  return hipGetLastError();
}

hipError_t launch_time_domain_u8(const float* hist, long long hist_stride, unsigned char* out,
                                 int n_channels, int len, long long out_stride, float in_scale, hipStream_t s) {
  return launch_time_domain(hist, hist_stride, out, n_channels, len, out_stride, in_scale, s);
}

hipError_t launch_time_domain_f32(const float* hist, long long hist_stride, float* out,
                                  int n_channels, int len, long long out_stride, float in_scale, hipStream_t s) {
  return launch_time_domain(hist, hist_stride, out, n_channels, len, out_stride, in_scale, s);
}

}  // namespace sg
