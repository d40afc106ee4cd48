// streamsbench.hip -- round 6 diagnostic (not product): N concurrent write streams, each a wave storing consecutive 2 KiB rows
// (eight 256-byte wave-stores per row, the STFT kernel's output pattern), the streams `stride` bytes apart -- into each of several
// separately allocated buffers.  Does a plain write pattern of that shape see the same fast / slow allocations the STFT launch
// sees (tools/addr_grid.py), and from how many streams on?
//   hipcc -O3 --offload-arch=gfx950 -o tools/streamsbench tools/streamsbench.hip && tools/streamsbench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

// one wave per workgroup; wave w writes rows [w * rows_per_stream, (w + 1) * rows_per_stream) of 2 KiB each
__global__ __launch_bounds__(64) void streams(float* out, long long stride_floats, int rows, int nt) {
  float* p = out + (long long)blockIdx.x * stride_floats + threadIdx.x;
  float v = (float)blockIdx.x;
  for (int r = 0; r < rows; ++r) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (nt) __builtin_nontemporal_store(v, p + i * 64);
      else p[i * 64] = v;
    }
    p += 512;
    v += 1.f;
  }
}

// the same bytes as a read + write mix closer to the STFT: every row also reads 1 KiB from a second buffer
__global__ __launch_bounds__(64) void streams_rw(const float* in, float* out, long long in_stride, long long out_stride, int rows) {
  const float* q = in + (long long)blockIdx.x * in_stride + threadIdx.x;
  float* p = out + (long long)blockIdx.x * out_stride + threadIdx.x;
  float acc = 0.f;
  for (int r = 0; r < rows; ++r) {
    const float a = q[0], b = q[64], c = q[128], d = q[192];
    acc += a + b + c + d;
#pragma unroll
    for (int i = 0; i < 8; ++i) __builtin_nontemporal_store(acc + (float)i, p + i * 64);
    p += 512;
    q += 256;
  }
}

int main() {
  const long long ROWS = 112497;                 // config 2's output: 112,497 rows of 2 KiB
  const size_t out_bytes = (size_t)ROWS * 2048, in_bytes = (size_t)ROWS * 1024 + 4096;
  const int NB = 10;
  std::vector<float*> outs(NB), ins(NB);
  for (int i = 0; i < NB; ++i) CK(hipMalloc((void**)&ins[i], in_bytes));
  for (int i = 0; i < NB; ++i) CK(hipMalloc((void**)&outs[i], out_bytes));
  for (int i = 0; i < NB; ++i) { CK(hipMemset(ins[i], 0, in_bytes)); CK(hipMemset(outs[i], 0, out_bytes)); }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](int b, int n_streams, int rows, int mode) -> float {   // us
    const long long stride = (long long)rows * 512;
    CK(hipEventRecord(e0, 0));
    if (mode == 2) hipLaunchKernelGGL(streams_rw, dim3(n_streams), dim3(64), 0, 0, ins[b], outs[b], (long long)rows * 256, stride, rows);
    else hipLaunchKernelGGL(streams, dim3(n_streams), dim3(64), 0, 0, outs[b], stride, rows, mode);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f;
  };
  auto survey = [&](const char* tag, int n_streams, int rows, int mode, int reps) {
    std::vector<std::vector<float>> t(NB);
    for (int r = 0; r < reps + 2; ++r)
      for (int b = 0; b < NB; ++b) {
        const float us = run(b, n_streams, rows, mode);
        if (r >= 2) t[b].push_back(us);
      }
    std::printf("%-58s", tag);
    float lo = 1e9f, hi = 0;
    for (int b = 0; b < NB; ++b) {
      std::sort(t[b].begin(), t[b].end());
      const float m = t[b][t[b].size() / 2];
      std::printf(" %6.1f", m);
      lo = std::min(lo, m); hi = std::max(hi, m);
    }
    std::printf("   spread %.1f %%\n", 100.f * (hi - lo) / lo);
  };
  std::printf("median us per launch into each of %d output allocations of %zu bytes (the launch writes n_streams x rows x 2 KiB)\n", NB, out_bytes);
  char tag[128];
  for (int mode = 1; mode <= 2; ++mode)
    for (int n : {4018, 8036, 2009, 1024, 512}) {
      const int rows = (int)(ROWS / n);
      std::snprintf(tag, sizeof tag, "%s %5d streams x %3d rows (stride %4d KiB)", mode == 2 ? "read+write" : "nt write  ", n, rows, rows * 2);
      survey(tag, n, rows, mode, 9);
    }
  // one sequential grid-stride fill for comparison (what torch's fill_ does)
  survey("plain stores  4018 streams x  28 rows", 4018, 28, 0, 9);
  return 0;
}
