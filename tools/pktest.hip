// pktest.hip -- checks every helper of pk_math.hpp against its scalar formula on the GPU.
#include "../spectrogram_amd/csrc/pk_math.hpp"
#include <cstdio>
#include <cstdlib>
#include <cmath>
using sg::f2;
#define N_H 13
__global__ void k(const float* in, float* out) {
  const int i = threadIdx.x;
  f2 a = {in[6 * i], in[6 * i + 1]}, b = {in[6 * i + 2], in[6 * i + 3]}, c = {in[6 * i + 4], in[6 * i + 5]};
  f2 r[N_H] = {sg::pk_add_mi(a, b), sg::pk_add_pi(a, b), sg::pk_add_cj(a, b), sg::pk_sub_cj(a, b), sg::pk_mul_x(a, b),
               sg::pk_fma_x(a, b, c), sg::pk_fnma_x(a, b, c), sg::pk_fma_y(a, b, c), sg::pk_fma_iy(a, b, c),
               sg::pk_fma_mi_x(a, b, c), sg::pk_fma_pi_x(a, b, c), sg::pk_fms(a, b, c), sg::pk_cfma(a, b, c)};
  for (int h = 0; h < N_H; ++h) { out[(i * N_H + h) * 2] = r[h].x; out[(i * N_H + h) * 2 + 1] = r[h].y; }
}
int main() {
  const int n = 64;
  float hin[6 * n], hout[2 * N_H * n];
  srand(1);
  for (int i = 0; i < 6 * n; ++i) hin[i] = (float)rand() / RAND_MAX * 2 - 1;
  float *din, *dout;
  if (hipMalloc(&din, sizeof hin) != hipSuccess || hipMalloc(&dout, sizeof hout) != hipSuccess) return 2;
  hipMemcpy(din, hin, sizeof hin, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(n), 0, 0, din, dout);
  if (hipMemcpy(hout, dout, sizeof hout, hipMemcpyDeviceToHost) != hipSuccess) return 2;
  const char* names[N_H] = {"add_mi", "add_pi", "add_cj", "sub_cj", "mul_x", "fma_x", "fnma_x", "fma_y", "fma_iy", "fma_mi_x", "fma_pi_x", "fms", "cfma"};
  int bad = 0;
  for (int h = 0; h < N_H; ++h) {
    double worst = 0;
    for (int i = 0; i < n; ++i) {
      double ax = hin[6 * i], ay = hin[6 * i + 1], bx = hin[6 * i + 2], by = hin[6 * i + 3], cx = hin[6 * i + 4], cy = hin[6 * i + 5];
      double ex, ey;
      switch (h) {
        case 0: ex = ax + by; ey = ay - bx; break;
        case 1: ex = ax - by; ey = ay + bx; break;
        case 2: ex = ax + bx; ey = ay - by; break;
        case 3: ex = ax - bx; ey = ay + by; break;
        case 4: ex = ax * bx; ey = ay * bx; break;
        case 5: ex = ax * bx + cx; ey = ay * bx + cy; break;
        case 6: ex = -ax * bx + cx; ey = -ay * bx + cy; break;
        case 7: ex = ax * by + cx; ey = ay * by + cy; break;
        case 8: ex = -ay * by + cx; ey = ax * by + cy; break;
        case 9: ex = ay * bx + cx; ey = -ax * bx + cy; break;
        case 10: ex = -ay * bx + cx; ey = ax * bx + cy; break;
        case 11: ex = ax * bx - cx; ey = ay * by - cy; break;
        default: ex = ax * bx - ay * by + cx; ey = ax * by + ay * bx + cy; break;
      }
      worst = fmax(worst, fmax(fabs(hout[(i * N_H + h) * 2] - ex), fabs(hout[(i * N_H + h) * 2 + 1] - ey)));
    }
    printf("%-9s max err %.2e %s\n", names[h], worst, worst < 1e-6 ? "ok" : "WRONG");
    bad += worst >= 1e-6;
  }
  return bad ? 1 : 0;
}
