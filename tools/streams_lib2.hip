// streams_lib2.hip -- round 6 diagnostic (not product): write-stream skeletons of the config-2 launch with the store shape as a
// parameter, to see which shapes the output allocation's physical placement does NOT slow down (tools/addr_synth2.py).
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o /tmp/libstreams2.so tools/streams_lib2.hip
#include <hip/hip_runtime.h>

typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));

// every wave owns `rows` consecutive 2 KiB rows.  W = bytes per lane and store: 4 (eight 256-byte wave-stores per row: the STFT's
// shape), 8 (four of 512 bytes), 16 (two of 1 KiB).  NT: non-temporal.  PAIR: two rows are written per visit (4 KiB bursts).
template <int W, bool NT, bool PAIR>
__global__ __launch_bounds__(64) void wr(float* __restrict__ out, int rows) {
  float* base = out + (long long)blockIdx.x * rows * 512;
  const int l = threadIdx.x;
  float v = (float)blockIdx.x;
  const int step = PAIR ? 2 : 1;
  for (int r = 0; r + step <= rows; r += step) {
#pragma unroll
    for (int rr = 0; rr < step; ++rr) {
      float* p = base + (long long)(r + rr) * 512;
      if constexpr (W == 4) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { if (NT) __builtin_nontemporal_store(v, p + i * 64 + l); else p[i * 64 + l] = v; }
      } else if constexpr (W == 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { f2v x = {v, v}; f2v* q = (f2v*)(p + i * 128) + l; if (NT) __builtin_nontemporal_store(x, q); else *q = x; }
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) { f4v x = {v, v, v, v}; f4v* q = (f4v*)(p + i * 256) + l; if (NT) __builtin_nontemporal_store(x, q); else *q = x; }
      }
    }
    v += 1.f;
  }
}

extern "C" int sb2_launch(int shape, float* out, int n_streams, int rows, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  dim3 g(n_streams), b(64);
  switch (shape) {
    case 0: hipLaunchKernelGGL((wr<4, true, false>), g, b, 0, s, out, rows); break;
    case 1: hipLaunchKernelGGL((wr<4, false, false>), g, b, 0, s, out, rows); break;
    case 2: hipLaunchKernelGGL((wr<8, true, false>), g, b, 0, s, out, rows); break;
    case 3: hipLaunchKernelGGL((wr<16, true, false>), g, b, 0, s, out, rows); break;
    case 4: hipLaunchKernelGGL((wr<16, false, false>), g, b, 0, s, out, rows); break;
    case 5: hipLaunchKernelGGL((wr<4, true, true>), g, b, 0, s, out, rows); break;
    case 6: hipLaunchKernelGGL((wr<16, true, true>), g, b, 0, s, out, rows); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}
