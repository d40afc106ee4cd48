// xlane.hpp -- register <-> lane transpositions inside one wavefront, without LDS.
//
// Between two radix-8 passes of the FFT the three index bits a lane has just produced (held in
// the register index r of v[r]) have to trade places with three index bits that are spread over
// lanes.  That is a transposition of one REG bit with one LANE bit, three times.  For REG bit B
// and lane bit b, with (a, c) = (v[r], v[r | 1 << B]) for every r whose bit B is clear:
//     lanes whose bit b is 0 keep a and receive the partner lane's a into c,
//     lanes whose bit b is 1 keep c and receive the partner lane's c into a      (partner = lane ^ (1 << b)).
// gfx950 has a one-instruction form of exactly this for b = 4 and b = 5 (v_permlane16_swap_b32,
// v_permlane32_swap_b32: "lanes [31:16] / [63:32] of vdst swap with lanes [15:0] / [31:0] of src");
// for b = 0..3 it is two v_cndmask_b32_dpp per dword (DPP brings the partner's value in as src0).
// Per frame of N = 1024 that is 80 plain VALU instructions in place of 16 ds_write_b64 +
// 16 ds_read_b64 and two LDS round trips on the wave's critical path.
//
// Hazards (LLVM gfx950 rules): a VALU write of a VGPR needs 2 wait states before a DPP or a
// v_permlane*_swap reads it -> every block opens with s_nop 1 and never reads what it has written.
// tools/xlanetest.hip checks every stage against the host model on the GPU.
#pragma once
#include <hip/hip_runtime.h>

#include "pk_math.hpp"

namespace sg {

#define SG_XL_DPP(dst, src0, src1, ctl) \
  "v_cndmask_b32_dpp %[" #dst "], %[" #src0 "], %[" #src1 "], vcc " ctl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"

// The lane masks arrive as 64-bit SGPR constants (loop-invariant: materialised once per kernel, one s_mov_b64 to vcc
// per half-stage).  v_cndmask_b32: D = vcc ? src1 : dpp(src0).  First half (vcc = lanes with bit 0): a' = a on those
// lanes, partner's c elsewhere.  Second half (vcc = lanes with bit 1): c' = c there, partner's a elsewhere.
#define SG_XL_STAGE(NAME, DPP_FROM_ABOVE, DPP_FROM_BELOW, MASK_LO, MASK_HI)                                   \
  __device__ __forceinline__ void NAME(f2& a0, f2& c0, f2& a1, f2& c1, f2& a2, f2& c2, f2& a3, f2& c3) {     \
    f2 na0, nc0, na1, nc1, na2, nc2, na3, nc3;                                                                \
    const unsigned long long mlo = MASK_LO##ull * 0x100000001ull, mhi = MASK_HI##ull * 0x100000001ull;       \
    asm("s_mov_b64 vcc, %[mlo]\n\ts_nop 1\n\t"                                                               \
        SG_XL_DPP(na0x, c0x, a0x, DPP_FROM_BELOW) SG_XL_DPP(na0y, c0y, a0y, DPP_FROM_BELOW)                    \
        SG_XL_DPP(na1x, c1x, a1x, DPP_FROM_BELOW) SG_XL_DPP(na1y, c1y, a1y, DPP_FROM_BELOW)                    \
        SG_XL_DPP(na2x, c2x, a2x, DPP_FROM_BELOW) SG_XL_DPP(na2y, c2y, a2y, DPP_FROM_BELOW)                    \
        SG_XL_DPP(na3x, c3x, a3x, DPP_FROM_BELOW) SG_XL_DPP(na3y, c3y, a3y, DPP_FROM_BELOW)                    \
        "s_mov_b64 vcc, %[mhi]\n\t"                                                                         \
        SG_XL_DPP(nc0x, a0x, c0x, DPP_FROM_ABOVE) SG_XL_DPP(nc0y, a0y, c0y, DPP_FROM_ABOVE)                    \
        SG_XL_DPP(nc1x, a1x, c1x, DPP_FROM_ABOVE) SG_XL_DPP(nc1y, a1y, c1y, DPP_FROM_ABOVE)                    \
        SG_XL_DPP(nc2x, a2x, c2x, DPP_FROM_ABOVE) SG_XL_DPP(nc2y, a2y, c2y, DPP_FROM_ABOVE)                    \
        SG_XL_DPP(nc3x, a3x, c3x, DPP_FROM_ABOVE) SG_XL_DPP(nc3y, a3y, c3y, DPP_FROM_ABOVE) "s_nop 0"          \
        : [na0x] "=&v"(na0.x), [na0y] "=&v"(na0.y), [na1x] "=&v"(na1.x), [na1y] "=&v"(na1.y),                 \
          [na2x] "=&v"(na2.x), [na2y] "=&v"(na2.y), [na3x] "=&v"(na3.x), [na3y] "=&v"(na3.y),                 \
          [nc0x] "=&v"(nc0.x), [nc0y] "=&v"(nc0.y), [nc1x] "=&v"(nc1.x), [nc1y] "=&v"(nc1.y),                 \
          [nc2x] "=&v"(nc2.x), [nc2y] "=&v"(nc2.y), [nc3x] "=&v"(nc3.x), [nc3y] "=&v"(nc3.y)                  \
        : [a0x] "v"(a0.x), [a0y] "v"(a0.y), [a1x] "v"(a1.x), [a1y] "v"(a1.y), [a2x] "v"(a2.x),                \
          [a2y] "v"(a2.y), [a3x] "v"(a3.x), [a3y] "v"(a3.y), [c0x] "v"(c0.x), [c0y] "v"(c0.y),                \
          [c1x] "v"(c1.x), [c1y] "v"(c1.y), [c2x] "v"(c2.x), [c2y] "v"(c2.y), [c3x] "v"(c3.x),                \
          [c3y] "v"(c3.y), [mlo] "s"(mlo), [mhi] "s"(mhi)                                                      \
        : "vcc");                                                                                              \
    a0 = na0; c0 = nc0; a1 = na1; c1 = nc1; a2 = na2; c2 = nc2; a3 = na3; c3 = nc3;                            \
  }

// lanes with bit = 1 read the lane below them (lane - 2^b), lanes with bit = 0 the lane above
SG_XL_STAGE(xl_stage_b0, "quad_perm:[1,0,3,2]", "quad_perm:[1,0,3,2]", 0x55555555, 0xaaaaaaaa)
SG_XL_STAGE(xl_stage_b1, "quad_perm:[2,3,0,1]", "quad_perm:[2,3,0,1]", 0x33333333, 0xcccccccc)
SG_XL_STAGE(xl_stage_b2, "row_shl:4", "row_shr:4", 0x0f0f0f0f, 0xf0f0f0f0)
SG_XL_STAGE(xl_stage_b3, "row_ror:8", "row_ror:8", 0x00ff00ff, 0xff00ff00)

#define SG_XL_PERMSTAGE(NAME, INSN)                                                                            \
  __device__ __forceinline__ void NAME(f2& a0, f2& c0, f2& a1, f2& c1, f2& a2, f2& c2, f2& a3, f2& c3) {      \
    asm("s_nop 1\n\t" INSN " %0, %8\n\t" INSN " %1, %9\n\t" INSN " %2, %10\n\t" INSN " %3, %11\n\t"           \
        INSN " %4, %12\n\t" INSN " %5, %13\n\t" INSN " %6, %14\n\t" INSN " %7, %15\n\ts_nop 0"                \
        : "+v"(a0.x), "+v"(a0.y), "+v"(a1.x), "+v"(a1.y), "+v"(a2.x), "+v"(a2.y), "+v"(a3.x), "+v"(a3.y),      \
          "+v"(c0.x), "+v"(c0.y), "+v"(c1.x), "+v"(c1.y), "+v"(c2.x), "+v"(c2.y), "+v"(c3.x), "+v"(c3.y));     \
  }
SG_XL_PERMSTAGE(xl_stage_b4, "v_permlane16_swap_b32")
SG_XL_PERMSTAGE(xl_stage_b5, "v_permlane32_swap_b32")

#undef SG_XL_DPP
#undef SG_XL_STAGE
#undef SG_XL_PERMSTAGE

template <int B>
__device__ __forceinline__ void xl_stage(f2& a0, f2& c0, f2& a1, f2& c1, f2& a2, f2& c2, f2& a3, f2& c3) {
  if constexpr (B == 0) xl_stage_b0(a0, c0, a1, c1, a2, c2, a3, c3);
  else if constexpr (B == 1) xl_stage_b1(a0, c0, a1, c1, a2, c2, a3, c3);
  else if constexpr (B == 2) xl_stage_b2(a0, c0, a1, c1, a2, c2, a3, c3);
  else if constexpr (B == 3) xl_stage_b3(a0, c0, a1, c1, a2, c2, a3, c3);
  else if constexpr (B == 4) xl_stage_b4(a0, c0, a1, c1, a2, c2, a3, c3);
  else xl_stage_b5(a0, c0, a1, c1, a2, c2, a3, c3);
}

// REG bit RB of v[0..7] <-> lane bit LB
template <int RB, int LB>
__device__ __forceinline__ void xl_swap_bit(f2 (&v)[8]) {
  if constexpr (RB == 0) xl_stage<LB>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
  else if constexpr (RB == 1) xl_stage<LB>(v[0], v[2], v[1], v[3], v[4], v[6], v[5], v[7]);
  else xl_stage<LB>(v[0], v[4], v[1], v[5], v[2], v[6], v[3], v[7]);
}

// REG bits 0, 1, 2 <-> lane bits LB0, LB0 + 1, LB0 + 2: afterwards v[r] of lane l holds what was
// v[(l >> LB0) & 7] of the lane whose bits LB0.. are r and whose other bits are l's.
template <int LB0>
__device__ __forceinline__ void xl_swap3(f2 (&v)[8]) {
  xl_swap_bit<0, LB0>(v);
  xl_swap_bit<1, LB0 + 1>(v);
  xl_swap_bit<2, LB0 + 2>(v);
}

}  // namespace sg
