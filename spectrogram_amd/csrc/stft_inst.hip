// stft_inst.hip -- one translation unit per FFT size: compiled once per
// -DSG_LOG2N=<5..14> so the sizes build in parallel (see spectrogram_amd/build.py).
#include "stft_kernel.hpp"
#include "launch.hpp"
#include <atomic>

#ifndef SG_LOG2N
#error "compile with -DSG_LOG2N=<log2 of the FFT size>"
#endif

namespace sg {

template <int SHIFT, int OUT>
static hipError_t launch_one(const StftParams& p, long long nblocks, hipStream_t stream) {
  using G = Geo<SG_LOG2N>;
  const int LDSB = Plan<SG_LOG2N, (SHIFT & 7)>::LDS_BYTES + ((OUT & OUT_LOGFREQ) ? p.lf_cols * 8 : 0);
  auto kern = stft_kernel<SG_LOG2N, SHIFT, OUT>;
  const int extra_lds = p.host_extra_lds;  // tuning only (sg_debug_set "extra_lds"): throttles occupancy without changing the code
  if (extra_lds > 0) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              LDSB + extra_lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(G::WG), LDSB + extra_lds, stream, p);
    return hipGetLastError();
  }
  if (LDSB > 64 * 1024) {
    // raise the kernel's dynamic-LDS cap once per instantiation (and again only if a larger request turns up),
    // not on every launch
    // The attribute is kept per device: the cache is per (device, instantiation).  Handles on different devices
    // (multidev.py, several N-API analysers) each raise their own device's cap; a race between threads only
    // repeats the call.
    static std::atomic<int> granted[kMaxDevices];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= kMaxDevices || LDSB > granted[dev].load(std::memory_order_relaxed)) {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < kMaxDevices) granted[dev].store(LDSB, std::memory_order_relaxed);
    }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(G::WG), LDSB, stream, p);
  return hipGetLastError();
}

template <int SHIFT>
static hipError_t launch_out(int out, const StftParams& p, long long nb, hipStream_t s) {
  switch (out) {
    case OUT_POWER: return launch_one<SHIFT, OUT_POWER>(p, nb, s);
    case OUT_MAG: return launch_one<SHIFT, OUT_MAG>(p, nb, s);
    case OUT_DB: return launch_one<SHIFT, OUT_DB>(p, nb, s);
    case OUT_DB_U8: return launch_one<SHIFT, OUT_DB_U8>(p, nb, s);
    case OUT_LOGFREQ | OUT_POWER: return launch_one<SHIFT, OUT_LOGFREQ | OUT_POWER>(p, nb, s);
    case OUT_LOGFREQ | OUT_MAG: return launch_one<SHIFT, OUT_LOGFREQ | OUT_MAG>(p, nb, s);
    case OUT_LOGFREQ | OUT_DB: return launch_one<SHIFT, OUT_LOGFREQ | OUT_DB>(p, nb, s);
    case OUT_LOGFREQ | OUT_DB_U8: return launch_one<SHIFT, OUT_LOGFREQ | OUT_DB_U8>(p, nb, s);
    case OUT_SMOOTH | OUT_POWER: return launch_one<SHIFT, OUT_SMOOTH | OUT_POWER>(p, nb, s);
    case OUT_SMOOTH | OUT_MAG: return launch_one<SHIFT, OUT_SMOOTH | OUT_MAG>(p, nb, s);
    case OUT_SMOOTH | OUT_DB: return launch_one<SHIFT, OUT_SMOOTH | OUT_DB>(p, nb, s);
    case OUT_SMOOTH | OUT_DB_U8: return launch_one<SHIFT, OUT_SMOOTH | OUT_DB_U8>(p, nb, s);
    // ... with the row leaving as log-frequency columns of the smoothed bins
    case OUT_SMOOTH | OUT_LOGFREQ | OUT_POWER: return launch_one<SHIFT, OUT_SMOOTH | OUT_LOGFREQ | OUT_POWER>(p, nb, s);
    case OUT_SMOOTH | OUT_LOGFREQ | OUT_MAG: return launch_one<SHIFT, OUT_SMOOTH | OUT_LOGFREQ | OUT_MAG>(p, nb, s);
    case OUT_SMOOTH | OUT_LOGFREQ | OUT_DB: return launch_one<SHIFT, OUT_SMOOTH | OUT_LOGFREQ | OUT_DB>(p, nb, s);
    case OUT_SMOOTH | OUT_LOGFREQ | OUT_DB_U8: return launch_one<SHIFT, OUT_SMOOTH | OUT_LOGFREQ | OUT_DB_U8>(p, nb, s);
    case OUT_ANALYSER | OUT_DB:  // poll mode runs on the generic float32 loads only
      if constexpr (SHIFT == 0) return launch_one<SHIFT, OUT_ANALYSER | OUT_DB>(p, nb, s);
      break;
    case OUT_ANALYSER | OUT_DB_U8:
      if constexpr (SHIFT == 0) return launch_one<SHIFT, OUT_ANALYSER | OUT_DB_U8>(p, nb, s);
      break;
    case OUT_ANALYSER | OUT_LOGFREQ | OUT_DB:  // ... and hands out log-frequency columns
      if constexpr (SHIFT == 0) return launch_one<SHIFT, OUT_ANALYSER | OUT_LOGFREQ | OUT_DB>(p, nb, s);
      break;
    case OUT_ANALYSER | OUT_LOGFREQ | OUT_DB_U8:
      if constexpr (SHIFT == 0) return launch_one<SHIFT, OUT_ANALYSER | OUT_LOGFREQ | OUT_DB_U8>(p, nb, s);
      break;
    // ... fused with the push (sg_push_poll_*): the newest samples are read from the caller's block
    case OUT_PEND | OUT_ANALYSER | OUT_DB:
      if constexpr (SHIFT == 0) return launch_one<SHIFT, OUT_PEND | OUT_ANALYSER | OUT_DB>(p, nb, s);
      break;
    case OUT_PEND | OUT_ANALYSER | OUT_DB_U8:
      if constexpr (SHIFT == 0) return launch_one<SHIFT, OUT_PEND | OUT_ANALYSER | OUT_DB_U8>(p, nb, s);
      break;
    case OUT_PEND | OUT_ANALYSER | OUT_LOGFREQ | OUT_DB:
      if constexpr (SHIFT == 0) return launch_one<SHIFT, OUT_PEND | OUT_ANALYSER | OUT_LOGFREQ | OUT_DB>(p, nb, s);
      break;
    case OUT_PEND | OUT_ANALYSER | OUT_LOGFREQ | OUT_DB_U8:
      if constexpr (SHIFT == 0) return launch_one<SHIFT, OUT_PEND | OUT_ANALYSER | OUT_LOGFREQ | OUT_DB_U8>(p, nb, s);
      break;
  }
  // the 256-column rows of N = 1024 / 2048 (OUT_LFFAST: the launcher has checked the shape)
  if constexpr (SG_LOG2N == 10 || SG_LOG2N == 11) {
    constexpr int F = OUT_LOGFREQ | OUT_LFFAST;
    switch (out) {
      case F | OUT_POWER: return launch_one<SHIFT, F | OUT_POWER>(p, nb, s);
      case F | OUT_MAG: return launch_one<SHIFT, F | OUT_MAG>(p, nb, s);
      case F | OUT_DB: return launch_one<SHIFT, F | OUT_DB>(p, nb, s);
      case F | OUT_DB_U8: return launch_one<SHIFT, F | OUT_DB_U8>(p, nb, s);
      case F | OUT_ANALYSER | OUT_DB:
        if constexpr (SHIFT == 0) return launch_one<SHIFT, F | OUT_ANALYSER | OUT_DB>(p, nb, s);
        break;
      case F | OUT_ANALYSER | OUT_DB_U8:
        if constexpr (SHIFT == 0) return launch_one<SHIFT, F | OUT_ANALYSER | OUT_DB_U8>(p, nb, s);
        break;
      case F | OUT_PEND | OUT_ANALYSER | OUT_DB:
        if constexpr (SHIFT == 0) return launch_one<SHIFT, F | OUT_PEND | OUT_ANALYSER | OUT_DB>(p, nb, s);
        break;
      case F | OUT_PEND | OUT_ANALYSER | OUT_DB_U8:
        if constexpr (SHIFT == 0) return launch_one<SHIFT, F | OUT_PEND | OUT_ANALYSER | OUT_DB_U8>(p, nb, s);
        break;
    }
  }
  return hipErrorInvalidValue;
}

#define SG_CAT2(a, b) a##b
#define SG_CAT(a, b) SG_CAT2(a, b)

hipError_t SG_CAT(launch_stft_, SG_LOG2N)(int shift, int out, const StftParams& p,
                                          long long nblocks, hipStream_t stream) {
  // LDS sample ring (load variants 1, 3): whole-wave frame slots with one point column per lane
  constexpr bool kLring = SG_LOG2N >= kLringMinLog2N && SG_LOG2N <= kLringMaxLog2N;
  switch (shift) {  // load variant | 8 for int16 input | 16 for packed int24 input
    case 0: return launch_out<0>(out, p, nblocks, stream);
    case 1: if constexpr (kLring) return launch_out<1>(out, p, nblocks, stream); break;
    case 3: if constexpr (kLring) return launch_out<3>(out, p, nblocks, stream); break;
    case 9: if constexpr (kLring) return launch_out<9>(out, p, nblocks, stream); break;
    case 11: if constexpr (kLring) return launch_out<11>(out, p, nblocks, stream); break;
    case 17: if constexpr (kLring) return launch_out<17>(out, p, nblocks, stream); break;  // packed int24 through the LDS ring
    case 19: if constexpr (kLring) return launch_out<19>(out, p, nblocks, stream); break;
    case 5:  // generic loads, unit stride / float32 / no down-mix known at launch: the sizes where it wins (spectro_api.hip)
      if constexpr (SG_LOG2N == 9 || SG_LOG2N >= 14) return launch_out<5>(out, p, nblocks, stream);
      break;
    case 2: return launch_out<2>(out, p, nblocks, stream);
    case 4: return launch_out<4>(out, p, nblocks, stream);
    case 8: return launch_out<8>(out, p, nblocks, stream);
    case 10: return launch_out<10>(out, p, nblocks, stream);
    case 12: return launch_out<12>(out, p, nblocks, stream);
    case 16: return launch_out<16>(out, p, nblocks, stream);  // packed int24, generic loads
  }
  return hipErrorInvalidValue;
}

}  // namespace sg
