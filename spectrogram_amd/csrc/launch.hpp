// launch.hpp -- host-side dispatch over the per-size kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include "stft_kernel.hpp"

namespace sg {

constexpr int kMinLog2N = 5;   // N = 32   (AnalyserNode minimum fftSize)
constexpr int kMaxLog2N = 14;  // N = 16384 (one workgroup of 1024 threads per frame)

#define SG_DECL(n) \
  hipError_t launch_stft_##n(int shift, int out, const StftParams& p, long long nblocks, hipStream_t s);
SG_DECL(5) SG_DECL(6) SG_DECL(7) SG_DECL(8) SG_DECL(9) SG_DECL(10) SG_DECL(11) SG_DECL(12) SG_DECL(13) SG_DECL(14)
#undef SG_DECL

// host mirror of Geo<LOG2N> (threads per frame, frames in flight per workgroup)
inline int threads_per_frame(int log2n) { return (1 << log2n) / 16; }
inline int slots_per_block(int log2n) {
  const int t = threads_per_frame(log2n);
  return t > 64 ? 1 : 256 / t;
}

// auxiliary kernels (aux_kernels.hip)
hipError_t launch_analyser_post(const float* mag, float* state, float* db_out, unsigned char* u8_out,
                                int n_channels, int bins, int len, long long out_stride, float tau,
                                float min_db, float byte_mul, unsigned char* ring_row, long long ring_stride,
                                hipStream_t s);
hipError_t launch_time_domain_u8(const float* hist, long long hist_stride, unsigned char* out,
                                 int n_channels, int len, long long out_stride, hipStream_t s);

}  // namespace sg
