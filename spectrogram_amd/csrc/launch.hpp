// launch.hpp -- host-side dispatch over the per-size kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include "stft_kernel.hpp"

namespace sg {

constexpr int kMinLog2N = 5;   // N = 32   (AnalyserNode minimum fftSize)
constexpr int kMaxLog2N = 15;  // N = 32768 (AnalyserNode maximum; 1024 lanes x 2 virtual threads per frame)

constexpr int kMaxDevices = 64;                           // per-device caches (dynamic-LDS cap) hold this many ordinals
constexpr int kLringMinLog2N = 10, kLringMaxLog2N = 13;  // sizes that have the LDS-sample-ring load variants

#define SG_DECL(n) \
  hipError_t launch_stft_##n(int shift, int out, const StftParams& p, long long nblocks, hipStream_t s);
SG_DECL(5) SG_DECL(6) SG_DECL(7) SG_DECL(8) SG_DECL(9) SG_DECL(10) SG_DECL(11) SG_DECL(12) SG_DECL(13) SG_DECL(14) SG_DECL(15)
#undef SG_DECL

// host mirror of Geo<LOG2N>
inline int virtual_threads_per_frame(int log2n) { return Geo<5>::TV * ((1 << log2n) / 32); }  // N / 16
inline int slots_per_block(int log2n) {
  switch (log2n) {
#define SG_SLOTS(n) case n: return Geo<n>::SLOTS;
    SG_SLOTS(5) SG_SLOTS(6) SG_SLOTS(7) SG_SLOTS(8) SG_SLOTS(9) SG_SLOTS(10) SG_SLOTS(11) SG_SLOTS(12) SG_SLOTS(13)
    SG_SLOTS(14) SG_SLOTS(15)
#undef SG_SLOTS
  }
  return 1;
}

// dynamic LDS a workgroup of that size asks for at most (exchange buffers + constant slab + the sample rings of the
// LDS-ring load variants), before the log-frequency table
inline int lds_bytes_per_block(int log2n) {
  switch (log2n) {
#define SG_LDS(n) case n: return (n >= kLringMinLog2N && n <= kLringMaxLog2N) ? Plan<n, 1>::LDS_BYTES : Plan<n, 0>::LDS_BYTES;
    SG_LDS(5) SG_LDS(6) SG_LDS(7) SG_LDS(8) SG_LDS(9) SG_LDS(10) SG_LDS(11) SG_LDS(12) SG_LDS(13)
    SG_LDS(14) SG_LDS(15)
#undef SG_LDS
  }
  return 0;
}
constexpr int kLdsPerCu = 160 * 1024;  // gfx950

// auxiliary kernels (aux_kernels.hip)
hipError_t launch_analyser_emit(const float* state, float* db_out, unsigned char* u8_out, int n_channels, int bins,
                                int len, long long out_stride, float min_db, float byte_mul, float floor_db,
                                unsigned char* ring_row, long long ring_stride, const uint2* lf_tab, int row,
                                hipStream_t s);
// dst = channel 0's history row (2 * hist_h floats, the double-written ring), a0 = ring position of the first new sample
// fmt: 0 float32, 1 int16, 2 packed int24; w: null = the mean of the K interleaved samples, else K weights
hipError_t launch_push_mix(const void* src, int fmt, long long src_stride, float* dst, long long dst_stride, int n_channels,
                           int n_new, int k, int a0, int hist_h, const float* w, hipStream_t s);
hipError_t launch_smooth_scan(const float* mag, void* out, int out_kind, long long n_channels, long long frames,
                              int bins, long long out_channel_stride, float tau, float floor_db, float min_db,
                              float byte_mul, const uint2* lf_tab, int cols, hipStream_t s);
hipError_t launch_time_domain_u8(const float* hist, long long hist_stride, unsigned char* out,
                                 int n_channels, int len, long long out_stride, float in_scale, hipStream_t s);
hipError_t launch_time_domain_f32(const float* hist, long long hist_stride, float* out,
                                 int n_channels, int len, long long out_stride, float in_scale, hipStream_t s);

}  // namespace sg
