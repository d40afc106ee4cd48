// aux_kernels.hip -- the small kernels around the STFT hot path (poll mode).
#include "launch.hpp"

namespace sg {

// AnalyserNode "smoothing over time" + dB + byte map, one thread per (channel, bin).
// Serves analyser.getByteFrequencyData / getFloatFrequencyData with
// smoothingTimeConstant != 0 (reference: 3D/visualizer.js:351-352 uses 0.75).
__global__ void analyser_post_kernel(const float* __restrict__ mag, float* __restrict__ state,
                                     float* __restrict__ db_out, unsigned char* __restrict__ u8_out,
                                     int n_channels, int bins, int len, long long out_stride,
                                     float tau, float min_db, float byte_mul,
                                     unsigned char* __restrict__ ring_row, long long ring_stride) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)n_channels * bins) return;
  const int c = (int)(i / bins), k = (int)(i % bins);
  float sm = tau * state[i] + (1.0f - tau) * mag[i];
  if (!(fabsf(sm) < __builtin_inff())) sm = 0.f;  // NaN / inf resets the smoother
  state[i] = sm;
  const float y = 6.02059991327962f * __builtin_amdgcn_logf(sm);  // 20 log10 = 20 log10(2) * log2
  const unsigned char bq = (unsigned char)fminf(fmaxf(__builtin_floorf((y - min_db) * byte_mul), 0.f), 255.f);
  // history ring row (the reference's texture row at yoffset): always the full row
  if (ring_row) ring_row[c * ring_stride + k] = bq;
  if (k >= len) return;
  if (db_out) db_out[c * out_stride + k] = y;
  if (u8_out) u8_out[c * out_stride + k] = bq;
}

hipError_t launch_analyser_post(const float* mag, float* state, float* db_out, unsigned char* u8_out,
                                int n_channels, int bins, int len, long long out_stride, float tau,
                                float min_db, float byte_mul, unsigned char* ring_row, long long ring_stride,
                                hipStream_t s) {
  const long long n = (long long)n_channels * bins;
  const unsigned nb = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(analyser_post_kernel, dim3(nb), dim3(256), 0, s, mag, state, db_out, u8_out,
                     n_channels, bins, len, out_stride, tau, min_db, byte_mul, ring_row, ring_stride);
  return hipGetLastError();
}

// Batch-mode "smoothing over time" (AnalyserNode, tau != 0; the reference's bar-graph mode uses
// 0.75, 3D/visualizer.js:351): X^[f][k] = tau X^[f-1][k] + (1 - tau) |X[f][k]| along the frames of
// one channel, then the configured output kind.  One thread per (channel, bin) walks the frames;
// consecutive lanes read consecutive bins, so every step is one coalesced row segment.
__global__ void smooth_scan_kernel(const float* __restrict__ mag, void* __restrict__ out, int out_kind,
                                   long long n_channels, long long frames, int bins,
                                   long long out_channel_stride, float tau, float floor_db, float min_db,
                                   float byte_mul) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_channels * bins) return;
  const long long c = i / bins;
  const int k = (int)(i % bins);
  const float* m = mag + c * frames * bins + k;
  float* of = reinterpret_cast<float*>(out) + c * out_channel_stride + k;
  unsigned char* ob = reinterpret_cast<unsigned char*>(out) + c * out_channel_stride + k;
  float sm = 0.f;
  for (long long f = 0; f < frames; ++f) {
    sm = tau * sm + (1.0f - tau) * m[f * bins];
    if (!(fabsf(sm) < __builtin_inff())) sm = 0.f;
    if (out_kind == OUT_MAG) of[f * bins] = sm;
    else if (out_kind == OUT_POWER) of[f * bins] = sm * sm;
    else {
      const float y = fmaxf(6.02059991327962f * __builtin_amdgcn_logf(sm), floor_db);
      if (out_kind == OUT_DB) of[f * bins] = y;
      else ob[f * bins] = (unsigned char)fminf(fmaxf(__builtin_floorf((y - min_db) * byte_mul), 0.f), 255.f);
    }
  }
}

hipError_t launch_smooth_scan(const float* mag, void* out, int out_kind, long long n_channels, long long frames,
                              int bins, long long out_channel_stride, float tau, float floor_db, float min_db,
                              float byte_mul, hipStream_t s) {
  const long long n = n_channels * bins;
  hipLaunchKernelGGL(smooth_scan_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, mag, out, out_kind,
                     n_channels, frames, bins, out_channel_stride, tau, floor_db, min_db, byte_mul);
  return hipGetLastError();
}

// getByteTimeDomainData: b = clip(floor(128 (1 + x)), 0, 255) (3D/visualizer.js:363)
__global__ void time_domain_u8_kernel(const float* __restrict__ hist, long long hist_stride,
                                      unsigned char* __restrict__ out, int n_channels, int len,
                                      long long out_stride) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)n_channels * len) return;
  const int c = (int)(i / len), k = (int)(i % len);
  const float v = __builtin_floorf(128.f * (1.f + hist[c * hist_stride + k]));
  out[c * out_stride + k] = (unsigned char)fminf(fmaxf(v, 0.f), 255.f);
}

hipError_t launch_time_domain_u8(const float* hist, long long hist_stride, unsigned char* out,
                                 int n_channels, int len, long long out_stride, hipStream_t s) {
  const long long n = (long long)n_channels * len;
  const unsigned nb = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(time_domain_u8_kernel, dim3(nb), dim3(256), 0, s, hist, hist_stride, out,
                     n_channels, len, out_stride);
  return hipGetLastError();
}

}  // namespace sg
