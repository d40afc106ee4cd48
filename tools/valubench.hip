// valubench.hip -- VALU issue rates on gfx950: packed-f32 vs scalar-f32 ops (diagnostic only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
#define BODY(asmstr, ...) \
  for (int it = 0; it < iters; ++it) { REP8(REP8(asm volatile(asmstr : __VA_ARGS__);)) }

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f2 a = {1.0f + threadIdx.x * 1e-9f, 1.0f}, b = {1.0000001f, 0.9999999f}, c = {0.f, 1e-9f};
  f2 a1 = a + 1.f, a2 = a + 2.f, a3 = a + 3.f;  // 4 independent chains
  if (OP == 0) BODY("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5", "+v"(a), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c))
  if (OP == 1) BODY("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4", "+v"(a), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c))
  if (OP == 2) BODY("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4", "+v"(a), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b))
  if (OP == 3) BODY("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5", "+v"(a.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x) : "v"(b.x), "v"(c.x))
  if (OP == 4) BODY("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4", "+v"(a.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x) : "v"(c.x))
  if (OP == 5) BODY("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4", "+v"(a.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x) : "v"(c.x))
  if (OP == 6) BODY("v_mov_b64 %0, %4\n v_mov_b64 %1, %4\n v_mov_b64 %2, %4\n v_mov_b64 %3, %4", "+v"(a), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c))
  if (OP == 7) BODY("v_pk_mov_b32 %0, %4, %4\n v_pk_mov_b32 %1, %4, %4\n v_pk_mov_b32 %2, %4, %4\n v_pk_mov_b32 %3, %4, %4", "+v"(a), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c))
  if (OP == 8) BODY("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5", "+v"(a.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x) : "v"(b.x), "v"(c.x))
  if (OP == 9) BODY("v_pk_fma_f32 %0, %0, %4, %5 op_sel:[0,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %1, %1, %4, %5 op_sel:[0,1,0] op_sel_hi:[1,0,1]\n v_pk_fma_f32 %2, %2, %4, %5 op_sel:[0,1,0] op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %3, %4, %5 op_sel:[0,1,0] op_sel_hi:[1,0,1]", "+v"(a), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c))
  if (OP == 10) BODY("v_cndmask_b32_dpp %0, %4, %0, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_cndmask_b32_dpp %1, %4, %1, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_cndmask_b32_dpp %2, %4, %2, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_cndmask_b32_dpp %3, %4, %3, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1", "+v"(a.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x) : "v"(c.x) : "vcc")
  if (OP == 11) BODY("v_cndmask_b32_dpp %0, %4, %0, vcc row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_cndmask_b32_dpp %1, %4, %1, vcc row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_cndmask_b32_dpp %2, %4, %2, vcc row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_cndmask_b32_dpp %3, %4, %3, vcc row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1", "+v"(a.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x) : "v"(c.x) : "vcc")
  if (OP == 12) BODY("v_cndmask_b32_dpp %0, %4, %0, vcc row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_cndmask_b32_dpp %1, %4, %1, vcc row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_cndmask_b32_dpp %2, %4, %2, vcc row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_cndmask_b32_dpp %3, %4, %3, vcc row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1", "+v"(a.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x) : "v"(c.x) : "vcc")
  if (OP == 13) BODY("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7", "+v"(a.x), "+v"(a.y), "+v"(a1.x), "+v"(a1.y), "+v"(a2.x), "+v"(a2.y), "+v"(a3.x), "+v"(a3.y))
  if (OP == 14) BODY("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7", "+v"(a.x), "+v"(a.y), "+v"(a1.x), "+v"(a1.y), "+v"(a2.x), "+v"(a2.y), "+v"(a3.x), "+v"(a3.y))
  if (OP == 15) BODY("v_cndmask_b32 %0, %4, %0, vcc\n v_cndmask_b32 %1, %4, %1, vcc\n v_cndmask_b32 %2, %4, %2, vcc\n v_cndmask_b32 %3, %4, %3, vcc", "+v"(a.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x) : "v"(c.x) : "vcc")
  if (OP == 16) BODY("v_mov_b32_dpp %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1", "+v"(a.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x) : "v"(c.x))
  if (OP == 17) BODY("v_add_f32_dpp %0, %4, %0 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %4, %1 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %4, %2 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %4, %3 row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1", "+v"(a.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x) : "v"(c.x))
  if (OP == 18) BODY("v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3", "+v"(a.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x) : "v"(c.x))
  if (OP == 19) BODY("s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1", "+v"(a.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x) : "v"(c.x))
  f2 r = a + a1 + a2 + a3;
  if (r.x == 12345.678f) out[threadIdx.x] = r.x + r.y;
}

template <int OP>
void run(const char* name, float* d, int waves_per_simd) {
  const int iters = 2000, instr = iters * 64 * 4;
  dim3 grid(256 * waves_per_simd), block(256);  // 256 CUs x (4 waves per block = 1 per SIMD) x wps
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, d, iters);
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, d, iters);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  // wave-instructions per SIMD = instr * waves_per_simd ; cycles at f GHz = ms*1e6*f
  double per = ms * 1e6 / ((double)instr * waves_per_simd);  // ns per wave-instr per SIMD
  printf("%-34s waves/SIMD=%d  %.3f ms  %.3f ns/wave-instr/SIMD (= %.2f cyc @2.0GHz, %.2f @2.4GHz)\n", name, waves_per_simd, ms, per, per * 2.0, per * 2.4);
}

int main() {
  float* d; CK(hipMalloc(&d, 4096));
  for (int w : {1, 3, 4}) {
    run<0>("v_pk_fma_f32", d, w); run<1>("v_pk_add_f32", d, w); run<2>("v_pk_mul_f32", d, w);
    run<9>("v_pk_fma_f32 op_sel/neg", d, w);
    run<3>("v_fma_f32", d, w); run<8>("v_fmac_f32", d, w); run<4>("v_add_f32", d, w);
    run<5>("v_mov_b32", d, w); run<6>("v_mov_b64", d, w); run<7>("v_pk_mov_b32", d, w);
    run<10>("v_cndmask_b32_dpp quad_perm", d, w); run<11>("v_cndmask_b32_dpp row_shr:4", d, w);
    run<12>("v_cndmask_b32_dpp row_ror:8", d, w); run<13>("v_permlane32_swap_b32", d, w);
    run<14>("v_permlane16_swap_b32", d, w); run<15>("v_cndmask_b32", d, w); run<16>("v_mov_b32_dpp quad_perm", d, w);
    run<17>("v_add_f32_dpp row_ror:8", d, w); run<18>("v_log_f32", d, w); run<19>("s_nop 1", d, w);
  }
  return 0;
}
