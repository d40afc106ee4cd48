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
  if (OP == 20) {  // LDS: one 8-byte write and one 8-byte read per lane, lane-linear (conflict-free)
    __shared__ f2 sh[256 * 2];
    unsigned addr = threadIdx.x * 8;
    sh[threadIdx.x] = a;
    BODY("ds_write_b64 %4, %0\n ds_read_b64 %1, %4 offset:2048\n ds_write_b64 %4, %2 offset:2048\n ds_read_b64 %3, %4\n s_waitcnt lgkmcnt(0)", "+v"(a), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(addr) : "memory")
  }
  // round 4: the bank-masked move tried in place of the DPP selects of csrc/xlane.hpp (masked lanes keep their register), and a
  // v_cndmask_b32 whose condition is an SGPR pair other than vcc
  if (OP == 22) BODY("v_mov_b32_dpp %0, %4 row_ror:8 row_mask:0xf bank_mask:0xc\n v_mov_b32_dpp %1, %4 row_ror:8 row_mask:0xf bank_mask:0xc\n v_mov_b32_dpp %2, %4 row_ror:8 row_mask:0xf bank_mask:0x3\n v_mov_b32_dpp %3, %4 row_ror:8 row_mask:0xf bank_mask:0x3", "+v"(a.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x) : "v"(c.x))
  if (OP == 23) BODY("v_cndmask_b32_e64 %0, %4, %0, s[20:21]\n v_cndmask_b32_e64 %1, %4, %1, s[20:21]\n v_cndmask_b32_e64 %2, %4, %2, s[20:21]\n v_cndmask_b32_e64 %3, %4, %3, s[20:21]", "+v"(a.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x) : "v"(c.x) : "s20", "s21")
  if (OP == 21) {  // LDS: the paired forms the kernel's exchanges use
    __shared__ f2 sh2[256 * 4];
    unsigned addr = threadIdx.x * 16;
    sh2[threadIdx.x] = a;
    float __attribute__((ext_vector_type(4))) rd;
    BODY("ds_write2_b64 %5, %0, %1 offset1:1\n ds_read2_b64 %4, %5 offset1:1\n ds_write2_b64 %5, %2, %3 offset0:2 offset1:3\n ds_read2_b64 %4, %5 offset0:2 offset1:3\n s_waitcnt lgkmcnt(0)", "+v"(a), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(rd) : "v"(addr) : "memory")
  }
  f2 r = a + a1 + a2 + a3;
  if (r.x == 12345.678f) out[threadIdx.x] = r.x + r.y;
}

// energy mode: one op back to back for `secs` seconds on every SIMD (4 waves each) while the caller samples the
// board power (tools/energy_probe.py); prints the wave-instruction rate
template <int OP>
void run_for(const char* name, float* d, double secs) {
  const int iters = 20000, wps = 4;
  dim3 grid(256 * wps), block(256);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, d, iters);
  CK(hipDeviceSynchronize());
  double total_ms = 0; long long launches = 0;
  while (total_ms < secs * 1e3) {
    CK(hipEventRecord(a));
    for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, d, iters);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    total_ms += ms; launches += 8;
  }
  const double instr = (double)launches * iters * 64 * (OP >= 20 ? 4 : 4) * (256.0 * 4 * wps);  // wave-instructions, whole chip
  printf("%s: %.3f s, %.4g wave-instr/s chip-wide, %.2f ns per wave-instr per SIMD\n", name, total_ms / 1e3, instr / (total_ms / 1e3),
         total_ms * 1e6 / ((double)launches * iters * 64 * 4 * wps));
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

int main(int argc, char** argv) {
  float* d; CK(hipMalloc(&d, 4096));
  if (argc >= 3) {
    const int op = atoi(argv[1]); const double secs = atof(argv[2]);
    switch (op) {
      case 0: run_for<0>("v_pk_fma_f32", d, secs); break;
      case 1: run_for<1>("v_pk_add_f32", d, secs); break;
      case 3: run_for<3>("v_fma_f32", d, secs); break;
      case 4: run_for<4>("v_add_f32", d, secs); break;
      case 5: run_for<5>("v_mov_b32", d, secs); break;
      case 9: run_for<9>("v_pk_fma_f32 op_sel/neg", d, secs); break;
      case 12: run_for<12>("v_cndmask_b32_dpp row_ror:8", d, secs); break;
      case 13: run_for<13>("v_permlane32_swap_b32", d, secs); break;
      case 18: run_for<18>("v_log_f32", d, secs); break;
      case 19: run_for<19>("s_nop 1", d, secs); break;
      case 20: run_for<20>("ds_write_b64 + ds_read_b64", d, secs); break;
      case 21: run_for<21>("ds_write2_b64 + ds_read2_b64", d, secs); break;
      default: printf("no such op\n"); return 1;
    }
    return 0;
  }
  for (int w : {1, 3, 4}) {
    run<0>("v_pk_fma_f32", d, w); run<1>("v_pk_add_f32", d, w); run<2>("v_pk_mul_f32", d, w);
    run<9>("v_pk_fma_f32 op_sel/neg", d, w);
    run<3>("v_fma_f32", d, w); run<8>("v_fmac_f32", d, w); run<4>("v_add_f32", d, w);
    run<5>("v_mov_b32", d, w); run<6>("v_mov_b64", d, w); run<7>("v_pk_mov_b32", d, w);
    run<10>("v_cndmask_b32_dpp quad_perm", d, w); run<11>("v_cndmask_b32_dpp row_shr:4", d, w);
    run<12>("v_cndmask_b32_dpp row_ror:8", d, w); run<13>("v_permlane32_swap_b32", d, w);
    run<14>("v_permlane16_swap_b32", d, w); run<15>("v_cndmask_b32", d, w); run<16>("v_mov_b32_dpp quad_perm", d, w);
    run<17>("v_add_f32_dpp row_ror:8", d, w); run<18>("v_log_f32", d, w); run<19>("s_nop 1", d, w);
    run<22>("v_mov_b32_dpp row_ror:8 bank_mask", d, w); run<23>("v_cndmask_b32 sgpr-pair cond", d, w);
  }
  return 0;
}
