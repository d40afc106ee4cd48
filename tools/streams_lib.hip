// streams_lib.hip -- round 6 diagnostic (not product): the write / read+write stream kernels of tools/streamsbench.hip behind a C
// entry point, so that tools/addr_synth.py can time them into the very buffers it times the STFT launch into.
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o /tmp/libstreams.so tools/streams_lib.hip
#include <hip/hip_runtime.h>

// mode 0: every wave stores `rows` consecutive 2 KiB rows (eight 256-byte non-temporal wave-stores per row), streams `rows` rows apart
// mode 1: ... and reads 1 KiB per row from `in` (four 256-byte wave-loads requested two rows ahead, as the STFT's register window does)
// mode 2: as 1, with ~`work` dependent FMAs per row between the loads and the stores (arithmetic the size of a 1024-point frame's)
template <int MODE>
__global__ __launch_bounds__(64) void streams(const float* __restrict__ in, float* __restrict__ out, int rows, int work) {
  const long long w = blockIdx.x;
  float* p = out + w * rows * 512 + threadIdx.x;
  const float* q = in + w * rows * 256 + threadIdx.x;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  float n0[4] = {0, 0, 0, 0}, n1[4] = {0, 0, 0, 0};
  if (MODE >= 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { n0[i] = q[i * 64]; n1[i] = q[256 + i * 64]; }
  }
  for (int r = 0; r < rows; ++r) {
    float c[4] = {n0[0], n0[1], n0[2], n0[3]};
    if (MODE >= 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { n0[i] = n1[i]; n1[i] = (r + 2 < rows) ? q[(r + 2) * 256 + i * 64] : 0.f; }
      a0 += c[0]; a1 += c[1]; a2 += c[2]; a3 += c[3];
    }
    if (MODE == 2) {
      for (int k = 0; k < work; ++k) {
        a0 = __builtin_fmaf(a0, 1.0001f, a1); a1 = __builtin_fmaf(a1, 0.9999f, a2);
        a2 = __builtin_fmaf(a2, 1.0001f, a3); a3 = __builtin_fmaf(a3, 0.9999f, a0);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) __builtin_nontemporal_store((i & 1 ? a1 : a0) + (i & 2 ? a2 : a3) + (float)r, p + i * 64);
    p += 512;
  }
}

extern "C" int sb_launch(int mode, const float* in, float* out, int n_streams, int rows, int work, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (mode == 0) hipLaunchKernelGGL(streams<0>, dim3(n_streams), dim3(64), 0, s, in, out, rows, work);
  else if (mode == 1) hipLaunchKernelGGL(streams<1>, dim3(n_streams), dim3(64), 0, s, in, out, rows, work);
  else hipLaunchKernelGGL(streams<2>, dim3(n_streams), dim3(64), 0, s, in, out, rows, work);
  return (int)hipGetLastError();
}
