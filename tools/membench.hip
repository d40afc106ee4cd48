// membench.hip -- what HBM delivers for THIS kernel's traffic shape (diagnostic, not product).
// Each wave reads `rd` KiB and writes `wr` KiB per iteration, streaming, grid-stride over waves.
//   mode 0: dwordx4 loads + dwordx4 stores (ideal shape)
//   mode 1: dwordx2 loads (512 B/wave-instr) + dword stores (256 B/wave-instr): the STFT kernel's shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE, int RD_F, int WR_F>  // floats per lane per unit
__global__ __launch_bounds__(256) void mix(const float* __restrict__ in, float* __restrict__ out, long long units) {
  const int lane = threadIdx.x & 63;
  long long w = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nw = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long u = w; u < units; u += nw) {
    float acc = 0.f;
    if constexpr (MODE == 0) {
      const float4* s = reinterpret_cast<const float4*>(in + u * (RD_F * 64));
#pragma unroll
      for (int i = 0; i < RD_F / 4; ++i) { float4 v = s[lane + 64 * i]; acc += v.x + v.y + v.z + v.w; }
      float4* d = reinterpret_cast<float4*>(out + u * (WR_F * 64));
#pragma unroll
      for (int i = 0; i < WR_F / 4; ++i) d[lane + 64 * i] = make_float4(acc, acc + 1, acc + 2, acc + 3);
    } else {
      const float2* s = reinterpret_cast<const float2*>(in + u * (RD_F * 64));
#pragma unroll
      for (int i = 0; i < RD_F / 2; ++i) { float2 v = s[lane + 64 * i]; acc += v.x + v.y; }
      float* d = out + u * (WR_F * 64);
#pragma unroll
      for (int i = 0; i < WR_F; ++i) d[lane + 64 * i] = acc + i;
    }
  }
}

// mode "runs": wave w owns R consecutive units and walks them in order (the STFT kernel's pattern:
// 1 KiB in + 2 KiB out per unit, x2 loads, dword stores, 8 store instrs hitting 8 x 256 B segments)
__global__ __launch_bounds__(256) void runs(const float* __restrict__ in, float* __restrict__ out, long long units, int R, int xcd) {
  const int lane = threadIdx.x & 63;
  long long bid = blockIdx.x;
  if (xcd) { const long long nb = gridDim.x, per = nb >> 3; if (bid < (per << 3)) bid = (bid & 7) * per + (bid >> 3); }
  const long long w = bid * 4 + (threadIdx.x >> 6);
  const long long u0 = w * R;
  float acc = 0.f;
  float2 nx0, nx1;
  if (u0 < units) { const float2* s = reinterpret_cast<const float2*>(in + u0 * 256); nx0 = s[lane]; nx1 = s[lane + 64]; }
  for (int r = 0; r < R; ++r) {
    const long long u = u0 + r;
    if (u >= units) break;
    float2 c0 = nx0, c1 = nx1;
    if (r + 1 < R && u + 1 < units) { const float2* s = reinterpret_cast<const float2*>(in + (u + 1) * 256); nx0 = s[lane]; nx1 = s[lane + 64]; }
    acc += c0.x + c0.y + c1.x + c1.y;
    float* d = out + u * 512;
#pragma unroll
    for (int i = 0; i < 4; ++i) { d[lane + 64 * i] = acc + i; d[511 - lane - 64 * i] = acc - i; }
  }
}
void run_runs(float* in, float* out, size_t in_bytes, size_t out_bytes, int R, int xcd, long long units, int lds = 0) {
  const long long waves = (units + R - 1) / R;
  const int blocks = (int)((waves + 3) / 4);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  if (lds > 48 * 1024) CK(hipFuncSetAttribute((const void*)runs, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(runs, dim3(blocks), dim3(256), lds, 0, in, out, units, R, xcd);
  std::vector<float> ms;
  for (int r = 0; r < 10; ++r) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(runs, dim3(blocks), dim3(256), lds, 0, in, out, units, R, xcd);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float t; CK(hipEventElapsedTime(&t, a, b)); ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  double bytes = (double)units * 3072;
  printf("lds=%6d ", lds);
  printf("runs pattern R=%3d xcd=%d units=%8lld blocks=%6d  %8.1f GB/s (median %.4f ms min %.4f)\n", R, xcd, units, blocks, bytes / ms[5] / 1e6, ms[5], ms[0]);
}

template <int MODE, int RD_F, int WR_F>
void run(const char* name, float* in, float* out, size_t in_bytes, size_t out_bytes, int blocks) {
  long long units = std::min(in_bytes / (RD_F * 256), out_bytes / (WR_F * 256));
  if (RD_F == 0) units = out_bytes / (WR_F * 256);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((mix<MODE, RD_F ? RD_F : 4, WR_F>), dim3(blocks), dim3(256), 0, 0, in, out, units);
  std::vector<float> ms;
  for (int r = 0; r < 10; ++r) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((mix<MODE, RD_F ? RD_F : 4, WR_F>), dim3(blocks), dim3(256), 0, 0, in, out, units);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float t; CK(hipEventElapsedTime(&t, a, b)); ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  double bytes = (double)units * 256 * (RD_F + WR_F);
  printf("%-44s blocks=%5d  %8.1f GB/s (median %.3f ms, %.0f MB)\n", name, blocks, bytes / ms[5] / 1e6, ms[5], bytes / 1e6);
}

int main() {
  size_t in_bytes = 1ull << 30, out_bytes = 2ull << 30;
  float *in, *out; CK(hipMalloc(&in, in_bytes)); CK(hipMalloc(&out, out_bytes));
  CK(hipMemset(in, 0, in_bytes)); CK(hipMemset(out, 0, out_bytes));
  for (int blocks : {2048, 8192}) {
    run<0, 4, 8>("x4 loads 1 KiB + x4 stores 2 KiB per wave-unit", in, out, in_bytes, out_bytes, blocks);
    run<1, 4, 8>("x2 loads 1 KiB + dword stores 2 KiB", in, out, in_bytes, out_bytes, blocks);
    run<0, 8, 8>("x4 copy 1:1", in, out, in_bytes, out_bytes, blocks);
    run<0, 16, 4>("x4 read-heavy 4:1", in, out, in_bytes, out_bytes, blocks);
    run<0, 4, 16>("x4 write-heavy 1:4", in, out, in_bytes, out_bytes, blocks);
  }
  // occupancy throttle via dynamic LDS: 160 KB / lds blocks per CU (4 waves each)
  for (int lds : {0, 20000, 40000, 53000, 80000, 160000})
    for (int R : {8, 16}) run_runs(in, out, in_bytes, out_bytes, R, 1, 112497LL, lds);
  return 0;
}
