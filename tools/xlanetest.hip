// xlanetest.hip -- checks csrc/xlane.hpp's register <-> lane transpositions on the GPU against the
// host model (what tests/test_xlane_model.py states in numpy).  Build:
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o tools/xlanetest tools/xlanetest.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../spectrogram_amd/csrc/xlane.hpp"

using sg::f2;

template <int RB, int LB>
__global__ void k_bit(f2* p, float) {
  f2 v[8];
  for (int i = 0; i < 8; ++i) v[i] = p[threadIdx.x * 8 + i];
  sg::xl_swap_bit<RB, LB>(v);
  for (int i = 0; i < 8; ++i) p[threadIdx.x * 8 + i] = v[i];
}
// z = 0 at run time: a VALU write of every operand right in front of the block and a VALU read right behind it,
// which is how the blocks sit in the FFT (the head's wait states are what this exercises)
template <int LB0>
__global__ void k_swap3(f2* p, float z) {
  f2 v[8];
  for (int i = 0; i < 8; ++i) v[i] = p[threadIdx.x * 8 + i];
  for (int i = 7; i >= 0; --i) v[i] += z;
  sg::xl_swap3<LB0>(v);
  for (int i = 7; i >= 0; --i) v[i] += z;
  for (int i = 0; i < 8; ++i) p[threadIdx.x * 8 + i] = v[i];
}

static int check(const char* what, const std::vector<float>& got, int nbits, const int* rb, const int* lb) {
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 8; ++r) {
      int l2 = l, r2 = r;  // element (l, r) moves to (l2, r2)
      for (int i = 0; i < nbits; ++i) {
        const int lv = (l >> lb[i]) & 1, rv = (r >> rb[i]) & 1;
        l2 = (l2 & ~(1 << lb[i])) | (rv << lb[i]);
        r2 = (r2 & ~(1 << rb[i])) | (lv << rb[i]);
      }
      const float ex = (float)(l * 8 + r), ey = -(float)(l * 8 + r) - 0.5f;
      const float gx = got[(l2 * 8 + r2) * 2], gy = got[(l2 * 8 + r2) * 2 + 1];
      if (gx != ex || gy != ey) {
        if (bad < 4) printf("  %s: (lane %d, reg %d) -> expected at (lane %d, reg %d), found %.1f %.1f\n", what, l, r, l2, r2, gx, gy);
        ++bad;
      }
    }
  printf("%-24s %s\n", what, bad ? "FAIL" : "ok");
  return bad;
}

int main() {
  std::vector<float> init(64 * 8 * 2), got(64 * 8 * 2);
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 8; ++r) { init[(l * 8 + r) * 2] = (float)(l * 8 + r); init[(l * 8 + r) * 2 + 1] = -(float)(l * 8 + r) - 0.5f; }
  f2* d;
  if (hipMalloc(&d, init.size() * 4) != hipSuccess) { printf("hipMalloc failed\n"); return 2; }
  int bad = 0;
  volatile float zero = 0.f;
#define RUN(KERN, NAME, NB, RBS, LBS)                                            \
  do {                                                                           \
    hipMemcpy(d, init.data(), init.size() * 4, hipMemcpyHostToDevice);           \
    hipLaunchKernelGGL(KERN, dim3(1), dim3(64), 0, 0, d, zero);                       \
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; } \
    hipMemcpy(got.data(), d, init.size() * 4, hipMemcpyDeviceToHost);            \
    const int rb_[] = RBS; const int lb_[] = LBS;                                \
    bad += check(NAME, got, NB, rb_, lb_);                                       \
  } while (0)
#define L(...) {__VA_ARGS__}
  RUN((k_bit<0, 0>), "reg0<->lane0", 1, L(0), L(0));
  RUN((k_bit<1, 1>), "reg1<->lane1", 1, L(1), L(1));
  RUN((k_bit<2, 2>), "reg2<->lane2", 1, L(2), L(2));
  RUN((k_bit<0, 3>), "reg0<->lane3", 1, L(0), L(3));
  RUN((k_bit<1, 4>), "reg1<->lane4", 1, L(1), L(4));
  RUN((k_bit<2, 5>), "reg2<->lane5", 1, L(2), L(5));
  RUN((k_bit<2, 0>), "reg2<->lane0", 1, L(2), L(0));
  RUN((k_bit<0, 5>), "reg0<->lane5", 1, L(0), L(5));
  RUN((k_swap3<0>), "swap3<0>", 3, L(0, 1, 2), L(0, 1, 2));
  RUN((k_swap3<3>), "swap3<3>", 3, L(0, 1, 2), L(3, 4, 5));
  printf(bad ? "xlanetest: FAILED\n" : "xlanetest: all ok\n");
  return bad ? 1 : 0;
}
