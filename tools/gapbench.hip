// gapbench.hip -- what does a launch boundary cost between two dependent kernels of one stream?
// The bench launch (config 2) keeps its waves busy for 65 us (in-kernel stamps) but back-to-back launches come every
// 73-84 us (HIP events, rocprofv3 kernel durations): this measures that difference for kernels that leave a known
// amount of dirty data in the L2s.  For each case: period = events over 200 back-to-back launches / 200;
// span = last wave's end - first wave's start of one launch (s_memrealtime, 100 MHz).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o tools/gapbench tools/gapbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f4 __attribute__((ext_vector_type(4)));

// mode 0: nothing; 1: writes `units` KiB-units (nontemporal), 2: plain stores, 3: reads, 4: reads + nontemporal writes 1:2
template <int MODE>
__global__ __launch_bounds__(256) void k(const f4* __restrict__ in, f4* __restrict__ out, long long units, int spin,
                                         unsigned long long* __restrict__ stamps) {
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const int lane = threadIdx.x & 63;
  const long long w0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nw = ((long long)gridDim.x * blockDim.x) >> 6;
  float acc = 0.f;
  for (long long u = w0; u < units; u += nw) {
    if (MODE == 3 || MODE == 4) {
      const f4 v = __builtin_nontemporal_load(in + u * 64 + lane);
      acc += v.x + v.y + v.z + v.w;
    }
    if (MODE == 1 || MODE == 4) {
      __builtin_nontemporal_store(f4{acc, 1.f, 2.f, 3.f}, out + u * 128 + lane);
      if (MODE == 4) __builtin_nontemporal_store(f4{acc, 1.f, 2.f, 3.f}, out + u * 128 + 64 + lane);
    }
    if (MODE == 2) out[u * 128 + lane] = f4{acc, 1.f, 2.f, 3.f};
  }
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
  if (MODE == 3 && acc == 12345.f) out[0] = f4{acc, 0, 0, 0};
  if (stamps && threadIdx.x == 0) {
    __builtin_amdgcn_s_waitcnt(0);
    stamps[2 * blockIdx.x] = r0;
    stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
  }
}

template <int MODE>
static void run(const char* tag, const f4* in, f4* out, long long units, int spin, int blocks, unsigned long long* st) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, units, spin, nullptr);
  hipEventRecord(a, 0);
  const int n = 200;
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, units, spin, nullptr);
  hipEventRecord(b, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, units, spin, st);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned long long> h(2 * (size_t)blocks);
  hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
  unsigned long long lo = ~0ULL, hi = 0;
  for (int i = 0; i < blocks; ++i) { lo = std::min(lo, h[2 * i]); hi = std::max(hi, h[2 * i + 1]); }
  const double span = (hi - lo) / 100.0, period = ms * 1e3 / n;
  printf("%-44s period %7.2f us  span %7.2f us  boundary %6.2f us\n", tag, period, span, period - span);
  fflush(stdout);
  hipEventDestroy(a); hipEventDestroy(b);
}

int main() {
  const int blocks = 1005;
  f4 *in, *out; unsigned long long* st;
  const long long max_units = 230LL << 10;  // KiB units
  hipMalloc((void**)&in, (size_t)max_units * 1024);
  hipMalloc((void**)&out, (size_t)max_units * 2048);
  hipMalloc((void**)&st, 8192 * 16);
  hipMemset(in, 0, (size_t)max_units * 1024);
  // settle: 60 ms of the heaviest case
  for (int i = 0; i < 800; ++i) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, in, out, 112LL << 10, 0, nullptr);
  hipDeviceSynchronize();
  run<0>("empty, 1005 x 256", in, out, 0, 0, blocks, st);
  run<0>("empty, 8192 x 256", in, out, 0, 0, 8192, st);
  run<0>("sleep ~60 us, no memory", in, out, 0, 48, blocks, st);
  char tag[96];
  for (long long mb : {2LL, 8LL, 32LL, 64LL, 230LL}) {
    snprintf(tag, sizeof tag, "nontemporal stores, %lld MB", mb);
    run<1>(tag, in, out, mb << 10, 0, blocks, st);   // a unit writes 1 KiB
  }
  for (long long mb : {8LL, 32LL, 230LL}) {
    snprintf(tag, sizeof tag, "plain stores, %lld MB", mb);
    run<2>(tag, in, out, mb << 10, 0, blocks, st);
  }
  for (long long mb : {32LL, 115LL}) {
    snprintf(tag, sizeof tag, "loads only, %lld MB", mb);
    run<3>(tag, in, out, mb << 10, 0, blocks, st);
  }
  run<4>("115 MB in + 230 MB out (the STFT's mix)", in, out, 115LL << 10, 0, blocks, st);
  run<4>("same, 8192 blocks", in, out, 115LL << 10, 0, 8192, st);
  return 0;
}
