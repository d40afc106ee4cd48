// diag_stream.hip -- libsgdiag.so: what this box streams for the STFT path's traffic mix.
//
// Measurement aid for bench.py's roofline object only (not part of the C ABI in include/spectro.h,
// never on the product path): a grid-stride kernel in which every wave reads `rd` KiB and writes
// `wr` KiB per unit with 16-byte accesses, i.e. the best-shaped stream of the same read : write
// ratio as one STFT frame (hop * 4 bytes in, N/2 * 4 bytes out).  bench.py reports its rate as
// `measured_stream_GBps` beside the 8 TB/s spec peak.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <vector>

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

template <int RD_F4, int WR_F4>  // float4 per lane per unit
__global__ __launch_bounds__(256) void stream_mix(const f4* __restrict__ in, f4* __restrict__ out, long long units) {
  const int lane = threadIdx.x & 63;
  const long long w0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nw = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long u = w0; u < units; u += nw) {
    float acc = 0.f;
    const f4* s = in + u * (RD_F4 * 64);
#pragma unroll
    for (int i = 0; i < RD_F4; ++i) {
      const f4 v = __builtin_nontemporal_load(s + lane + 64 * i);
      acc += v.x + v.y + v.z + v.w;
    }
    f4* d = out + u * (WR_F4 * 64);
#pragma unroll
    for (int i = 0; i < WR_F4; ++i) __builtin_nontemporal_store(f4{acc, acc + 1.f, acc + 2.f, acc + 3.f}, d + lane + 64 * i);
  }
}

}  // namespace

extern "C" {

// Streams `units` x (1 KiB read + 2 KiB written) per launch over caller-sized device buffers it allocates
// itself (in_bytes + out_bytes must exceed the 256 MiB Infinity Cache to mean anything); median of
// `reps` launches after 3 warm-ups.  Returns 0 and the rate in *gbps, or a negative hipError_t.
int sg_diag_stream_rate(long long total_bytes, int reps, double* gbps) {
  if (!gbps || total_bytes < (3 << 20) || reps < 1) return -1;
  const long long units = total_bytes / 3072;
  f4 *in = nullptr, *out = nullptr;
  hipError_t e = hipMalloc((void**)&in, (size_t)units * 1024);
  if (e != hipSuccess) return -(int)e;
  e = hipMalloc((void**)&out, (size_t)units * 2048);
  if (e != hipSuccess) { (void)hipFree(in); return -(int)e; }
  (void)hipMemset(in, 0, (size_t)units * 1024);
  (void)hipMemset(out, 0, (size_t)units * 2048);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  const int blocks = 8192;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream_mix<1, 2>), dim3(blocks), dim3(256), 0, 0, in, out, units);
  std::vector<float> ms;
  for (int r = 0; r < reps; ++r) {
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL((stream_mix<1, 2>), dim3(blocks), dim3(256), 0, 0, in, out, units);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float t = 0.f;
    (void)hipEventElapsedTime(&t, a, b);
    ms.push_back(t);
  }
  e = hipGetLastError();
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  (void)hipFree(in);
  (void)hipFree(out);
  if (e != hipSuccess) return -(int)e;
  std::sort(ms.begin(), ms.end());
  *gbps = (double)units * 3072.0 / (double)ms[ms.size() / 2] / 1e6;
  return 0;
}

}  // extern "C"
