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

#ifndef SG_XL_FUSED
#define SG_XL_FUSED 1  // lane bits 3..5: one asm block for the three stages (tools/xlanetest.hip checks both forms)
#endif

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


// REG bits 0, 1, 2 <-> lane bits 3, 4, 5 as ONE block (the exchange after radix-8 pass 1): the row_ror:8 stage into
// fresh registers n0..n7, then the two permlane-swap stages in place on them.  Fused, only the head needs the two
// wait states (the compiler's last VALU write -> first DPP read); inside, every read is at least 3 instructions
// after the write it depends on: the first half-stage writes n0 n2 n4 n6, the second n1 n3 n5 n7, the
// v_permlane16 order below starts with the pairs of the first half, and the v_permlane32 order with the registers
// the v_permlane16 stage wrote first.  6 s_nop / s_mov fewer per frame than three separate stages.
__device__ __forceinline__ void xl_swap3_hi(f2 (&v)[8]) {
  f2 n0, n1, n2, n3, n4, n5, n6, n7;
  const unsigned long long mlo = 0x00ff00ffull * 0x100000001ull, mhi = 0xff00ff00ull * 0x100000001ull;
#define SG_XL_R8 "row_ror:8"
#define SG_XL_P16(a, b) "v_permlane16_swap_b32 %[" #a "], %[" #b "]\n\t"
#define SG_XL_P32(a, b) "v_permlane32_swap_b32 %[" #a "], %[" #b "]\n\t"
  asm("s_mov_b64 vcc, %[mlo]\n\ts_nop 1\n\t"
      SG_XL_DPP(n0x, v1x, v0x, SG_XL_R8) SG_XL_DPP(n0y, v1y, v0y, SG_XL_R8)
      SG_XL_DPP(n2x, v3x, v2x, SG_XL_R8) SG_XL_DPP(n2y, v3y, v2y, SG_XL_R8)
      SG_XL_DPP(n4x, v5x, v4x, SG_XL_R8) SG_XL_DPP(n4y, v5y, v4y, SG_XL_R8)
      SG_XL_DPP(n6x, v7x, v6x, SG_XL_R8) SG_XL_DPP(n6y, v7y, v6y, SG_XL_R8)
      "s_mov_b64 vcc, %[mhi]\n\t"
      SG_XL_DPP(n1x, v0x, v1x, SG_XL_R8) SG_XL_DPP(n1y, v0y, v1y, SG_XL_R8)
      SG_XL_DPP(n3x, v2x, v3x, SG_XL_R8) SG_XL_DPP(n3y, v2y, v3y, SG_XL_R8)
      SG_XL_DPP(n5x, v4x, v5x, SG_XL_R8) SG_XL_DPP(n5y, v4y, v5y, SG_XL_R8)
      SG_XL_DPP(n7x, v6x, v7x, SG_XL_R8) SG_XL_DPP(n7y, v6y, v7y, SG_XL_R8)
      // REG bit 1 <-> lane bit 4: (n0, n2) (n4, n6) (n1, n3) (n5, n7)
      SG_XL_P16(n0x, n2x) SG_XL_P16(n0y, n2y) SG_XL_P16(n4x, n6x) SG_XL_P16(n4y, n6y)
      SG_XL_P16(n1x, n3x) SG_XL_P16(n1y, n3y) SG_XL_P16(n5x, n7x) SG_XL_P16(n5y, n7y)
      // REG bit 2 <-> lane bit 5: (n0, n4) (n2, n6) (n1, n5) (n3, n7)
      SG_XL_P32(n0x, n4x) SG_XL_P32(n0y, n4y) SG_XL_P32(n2x, n6x) SG_XL_P32(n2y, n6y)
      SG_XL_P32(n1x, n5x) SG_XL_P32(n1y, n5y) SG_XL_P32(n3x, n7x) "v_permlane32_swap_b32 %[n3y], %[n7y]"
      : [n0x] "=&v"(n0.x), [n0y] "=&v"(n0.y), [n1x] "=&v"(n1.x), [n1y] "=&v"(n1.y), [n2x] "=&v"(n2.x),
        [n2y] "=&v"(n2.y), [n3x] "=&v"(n3.x), [n3y] "=&v"(n3.y), [n4x] "=&v"(n4.x), [n4y] "=&v"(n4.y),
        [n5x] "=&v"(n5.x), [n5y] "=&v"(n5.y), [n6x] "=&v"(n6.x), [n6y] "=&v"(n6.y), [n7x] "=&v"(n7.x),
        [n7y] "=&v"(n7.y)
      : [v0x] "v"(v[0].x), [v0y] "v"(v[0].y), [v1x] "v"(v[1].x), [v1y] "v"(v[1].y), [v2x] "v"(v[2].x),
        [v2y] "v"(v[2].y), [v3x] "v"(v[3].x), [v3y] "v"(v[3].y), [v4x] "v"(v[4].x), [v4y] "v"(v[4].y),
        [v5x] "v"(v[5].x), [v5y] "v"(v[5].y), [v6x] "v"(v[6].x), [v6y] "v"(v[6].y), [v7x] "v"(v[7].x),
        [v7y] "v"(v[7].y), [mlo] "s"(mlo), [mhi] "s"(mhi)
      : "vcc");
#undef SG_XL_R8
#undef SG_XL_P16
#undef SG_XL_P32
  v[0] = n0; v[1] = n1; v[2] = n2; v[3] = n3; v[4] = n4; v[5] = n5; v[6] = n6; v[7] = n7;
}

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
  if constexpr (LB0 == 3 && SG_XL_FUSED) { xl_swap3_hi(v); return; }
  xl_swap_bit<0, LB0>(v);
  xl_swap_bit<1, LB0 + 1>(v);
  xl_swap_bit<2, LB0 + 2>(v);
}

}  // namespace sg
