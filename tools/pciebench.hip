// pciebench.hip -- how fast a kernel reads / writes page-locked host memory in place, by allocation flavour (round 5).
// The streaming path (sg_push / sg_push_poll_* with SG_IN_PINNED, polls with SG_OUT_PINNED) reads 2.1 MB and writes 1.05 MB of
// host memory per config-4 hop from inside kernels; sg_alloc_pinned uses hipHostMallocDefault.  usage: tools/pciebench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void rd(const f4* __restrict__ src, f4* __restrict__ dst, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
int main() {
  const size_t sizes[] = {131072, 1048576, 2097152, 8388608};
  struct { const char* name; unsigned flags; } kinds[] = {{"default", hipHostMallocDefault}, {"coherent", hipHostMallocCoherent},
      {"noncoherent", hipHostMallocNonCoherent}, {"writecombined", hipHostMallocWriteCombined}, {"mapped|portable", hipHostMallocMapped | hipHostMallocPortable}};
  f4* dev = nullptr;
  CK(hipMalloc((void**)&dev, 8388608));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (auto& k : kinds) {
    for (size_t sz : sizes) {
      void* h = nullptr;
      if (hipHostMalloc(&h, sz, k.flags) != hipSuccess) { (void)hipGetLastError(); printf("%-16s %8zu: allocation refused\n", k.name, sz); continue; }
      void* hd = nullptr;
      CK(hipHostGetDevicePointer(&hd, h, 0));
      for (size_t i = 0; i < sz / 4; ++i) ((float*)h)[i] = (float)i;
      const long n = (long)(sz / 16);
      const unsigned nb = (unsigned)((n + 255) / 256);
      double best[2] = {1e9, 1e9};
      for (int dir = 0; dir < 2; ++dir) {
        std::vector<float> ts;
        for (int it = 0; it < 30; ++it) {
          for (size_t i = 0; i < sz / 4; i += 1024) ((float*)h)[i] = (float)(i + it);   // the host touches the buffer between hops
          CK(hipEventRecord(a, 0));
          if (dir == 0) hipLaunchKernelGGL(rd, dim3(nb), dim3(256), 0, 0, (const f4*)hd, dev, n);
          else hipLaunchKernelGGL(rd, dim3(nb), dim3(256), 0, 0, (const f4*)dev, (f4*)hd, n);
          CK(hipEventRecord(b, 0));
          CK(hipEventSynchronize(b));
          float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
          if (it >= 5) ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        best[dir] = ts[ts.size() / 2];
      }
      // correctness of what the device saw on the last read pass: the host wrote (i + 29) at every 1024th float before it
      printf("%-16s %8zu B: kernel reads host %7.1f us (%5.1f GB/s)   kernel writes host %7.1f us (%5.1f GB/s)\n", k.name, sz,
             best[0] * 1e3, sz / best[0] / 1e6, best[1] * 1e3, sz / best[1] / 1e6);
      CK(hipHostFree(h));
    }
  }
  return 0;
}
