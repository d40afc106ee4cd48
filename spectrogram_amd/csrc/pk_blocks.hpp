// pk_blocks.hpp -- whole butterfly stages as single asm blocks on packed-f32 VALU ops.
//
// hipcc pads an s_nop between two dependent one-instruction asm statements and cannot
// interleave their chains; inside one block the hardware interlocks VALU->VALU dependencies
// and the instruction order below keeps dependent instructions at least two slots apart.
// Operands are arranged so that every result lands either in place or in a fresh
// early-clobber register and is then re-bound by name: no register moves.
// Modifier semantics: pk_math.hpp (each form there is verified by tools/pktest.hip).
#pragma once
#include "pk_math.hpp"

namespace sg {

#define SG_ADD_MI "op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]"
#define SG_ADD_PI "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]"
#define SG_SUB "neg_lo:[0,1] neg_hi:[0,1]"
#define SG_FMA_X "op_sel:[0,0,0] op_sel_hi:[1,0,1]"
#define SG_FNMA_X "op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]"
#define SG_FMA_Y "op_sel:[0,1,0] op_sel_hi:[1,1,1]"
#define SG_FMA_IY "op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
#define SG_FMA_MI_X "op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]"
#define SG_FMA_PI_X "op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]"
#define SG_FMS "neg_lo:[0,0,1] neg_hi:[0,0,1]"
#define SG_MUL_X "op_sel:[0,0] op_sel_hi:[1,0]"

// Stages 2+3 of the 8-point DFT over operands named v0..v7 (stage-1 output in, natural-order DFT out, in place) with
// temporaries t0..t7; cc.x = 1/sqrt(2).  18 instructions.
#define SG_O(n) "%[" #n "]"
#define SG_S23_BODY(v0, v1, v2, v3, v4, v5, v6, v7, t0, t1, t2, t3, t4, t5, t6, t7)                \
  "v_pk_add_f32 " SG_O(t0) ", " SG_O(v0) ", " SG_O(v2) "\n\t"                         /* b0 */      \
  "v_pk_add_f32 " SG_O(t1) ", " SG_O(v1) ", " SG_O(v3) "\n\t"                         /* b1 */      \
  "v_pk_add_f32 " SG_O(t2) ", " SG_O(v0) ", " SG_O(v2) " " SG_SUB "\n\t"              /* b2 */      \
  "v_pk_add_f32 " SG_O(t3) ", " SG_O(v1) ", " SG_O(v3) " " SG_SUB "\n\t"              /* d13 */     \
  "v_pk_add_f32 " SG_O(t4) ", " SG_O(v4) ", " SG_O(v6) " " SG_ADD_MI "\n\t"           /* b4 */      \
  "v_pk_add_f32 " SG_O(t6) ", " SG_O(v5) ", " SG_O(v5) " " SG_ADD_MI "\n\t"           /* s5 */      \
  "v_pk_add_f32 " SG_O(t7) ", " SG_O(v7) ", " SG_O(v7) " " SG_ADD_PI "\n\t"           /* q7 */      \
  "v_pk_add_f32 " SG_O(t5) ", " SG_O(v4) ", " SG_O(v6) " " SG_ADD_PI "\n\t"           /* b6 */      \
  "v_pk_add_f32 " SG_O(v0) ", " SG_O(t0) ", " SG_O(t1) "\n\t"                         /* X0 */      \
  "v_pk_add_f32 " SG_O(v4) ", " SG_O(t0) ", " SG_O(t1) " " SG_SUB "\n\t"              /* X4 */      \
  "v_pk_add_f32 " SG_O(t0) ", " SG_O(t6) ", " SG_O(t7) " " SG_SUB "\n\t"              /* u = s5 - q7 */ \
  "v_pk_add_f32 " SG_O(t1) ", " SG_O(t6) ", " SG_O(t7) "\n\t"                         /* g = s5 + q7 */ \
  "v_pk_add_f32 " SG_O(v2) ", " SG_O(t2) ", " SG_O(t3) " " SG_ADD_MI "\n\t"           /* X2 */      \
  "v_pk_add_f32 " SG_O(v6) ", " SG_O(t2) ", " SG_O(t3) " " SG_ADD_PI "\n\t"           /* X6 */      \
  "v_pk_fma_f32 " SG_O(v1) ", " SG_O(t0) ", %[cc], " SG_O(t4) " " SG_FMA_X "\n\t"     /* X1 */      \
  "v_pk_fma_f32 " SG_O(v3) ", " SG_O(t1) ", %[cc], " SG_O(t5) " " SG_FMA_MI_X "\n\t"  /* X3 */      \
  "v_pk_fma_f32 " SG_O(v5) ", " SG_O(t0) ", %[cc], " SG_O(t4) " " SG_FNMA_X "\n\t"    /* X5 */      \
  "v_pk_fma_f32 " SG_O(v7) ", " SG_O(t1) ", %[cc], " SG_O(t5) " " SG_FMA_PI_X           /* X7 */

__device__ __forceinline__ void dft8_stage23_blk(f2 (&v)[8], const f2 cc) {
  f2 t0, t1, t2, t3, t4, t5, t6, t7;
  asm(SG_S23_BODY(v0, v1, v2, v3, v4, v5, v6, v7, t0, t1, t2, t3, t4, t5, t6, t7)
      : [v0] "+v"(v[0]), [v1] "+v"(v[1]), [v2] "+v"(v[2]), [v3] "+v"(v[3]), [v4] "+v"(v[4]), [v5] "+v"(v[5]),
        [v6] "+v"(v[6]), [v7] "+v"(v[7]), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3),
        [t4] "=&v"(t4), [t5] "=&v"(t5), [t6] "=&v"(t6), [t7] "=&v"(t7)
      : [cc] "v"(cc));
}

// Stage 1 with the inter-pass twiddles fused: a_r = w_r v_r + w_{r+4} v_{r+4},
// a_{r+4} = 2 w_r v_r - a_r (w[r-1] = W^{kr}, W^0 = 1).  18 instructions.  Results are
// re-bound, not moved: X0 lands in a fresh register, X4 where v0 was, X1..3 in place,
// X5..7 in fresh registers.
#define SG_S1TW_BODY                                                                               \
  "v_pk_fma_f32 %[a0], %[v4], %[w4], %[r0] " SG_FMA_X "\n\t"                                       \
  "v_pk_mul_f32 %[m1], %[r1], %[w1] " SG_MUL_X "\n\t"                                              \
  "v_pk_mul_f32 %[m2], %[r2], %[w2] " SG_MUL_X "\n\t"                                              \
  "v_pk_mul_f32 %[m3], %[r3], %[w3] " SG_MUL_X "\n\t"                                              \
  "v_pk_fma_f32 %[a0], %[v4], %[w4], %[a0] " SG_FMA_IY "\n\t" /* a0 = v0 + w4 v4        (X0) */    \
  "v_pk_fma_f32 %[m1], %[r1], %[w1], %[m1] " SG_FMA_IY "\n\t" /* m1 = w1 v1 */                     \
  "v_pk_fma_f32 %[m2], %[r2], %[w2], %[m2] " SG_FMA_IY "\n\t"                                      \
  "v_pk_fma_f32 %[m3], %[r3], %[w3], %[m3] " SG_FMA_IY "\n\t"                                      \
  "v_pk_fma_f32 %[r0], %[r0], %[two], %[a0] " SG_FMS "\n\t"   /* r0 = 2 v0 - a0         (X4) */    \
  "v_pk_fma_f32 %[r1], %[v5], %[w5], %[m1] " SG_FMA_X "\n\t"                                       \
  "v_pk_fma_f32 %[r2], %[v6], %[w6], %[m2] " SG_FMA_X "\n\t"                                       \
  "v_pk_fma_f32 %[r3], %[v7], %[w7], %[m3] " SG_FMA_X "\n\t"                                       \
  "v_pk_fma_f32 %[r1], %[v5], %[w5], %[r1] " SG_FMA_IY "\n\t" /* r1 = w1 v1 + w5 v5     (X1) */    \
  "v_pk_fma_f32 %[r2], %[v6], %[w6], %[r2] " SG_FMA_IY "\n\t"                                      \
  "v_pk_fma_f32 %[r3], %[v7], %[w7], %[r3] " SG_FMA_IY "\n\t"                                      \
  "v_pk_fma_f32 %[m1], %[m1], %[two], %[r1] " SG_FMS "\n\t"   /* m1 = 2 w1 v1 - X1      (X5) */    \
  "v_pk_fma_f32 %[m2], %[m2], %[two], %[r2] " SG_FMS "\n\t"                                        \
  "v_pk_fma_f32 %[m3], %[m3], %[two], %[r3] " SG_FMS

__device__ __forceinline__ void dft8_stage1_tw_blk(f2 (&v)[8], const f2* w, const f2 two) {
  f2 a0, m1, m2, m3;
  f2 r0 = v[0], r1 = v[1], r2 = v[2], r3 = v[3];
  asm(SG_S1TW_BODY
      : [r0] "+v"(r0), [r1] "+v"(r1), [r2] "+v"(r2), [r3] "+v"(r3), [a0] "=&v"(a0), [m1] "=&v"(m1),
        [m2] "=&v"(m2), [m3] "=&v"(m3)
      : [v4] "v"(v[4]), [v5] "v"(v[5]), [v6] "v"(v[6]), [v7] "v"(v[7]), [w1] "v"(w[0]), [w2] "v"(w[1]),
        [w3] "v"(w[2]), [w4] "v"(w[3]), [w5] "v"(w[4]), [w6] "v"(w[5]), [w7] "v"(w[6]), [two] "v"(two));
  v[0] = a0; v[4] = r0;
  v[1] = r1; v[5] = m1;
  v[2] = r2; v[6] = m2;
  v[3] = r3; v[7] = m3;
}

// A whole twiddled radix-8 pass (the two blocks above) as ONE statement: between two asm statements the compiler
// must assume a hazard and pads an s_nop.  After stage 1, X0 = a0, X4 = r0, X1..3 = r1..3, X5..7 = m1..3 and the
// registers of v4..v7 are dead: they serve as four of the stage-2/3 temporaries.
__device__ __forceinline__ void dft8_tw_pass_blk(f2 (&v)[8], const f2* w, const f2 two, const f2 cc) {
  f2 a0, m1, m2, m3, t4, t5, t6, t7;
  f2 r0 = v[0], r1 = v[1], r2 = v[2], r3 = v[3], q4 = v[4], q5 = v[5], q6 = v[6], q7 = v[7];
  asm(SG_S1TW_BODY "\n\t"
      SG_S23_BODY(a0, r1, r2, r3, r0, m1, m2, m3, v4, v5, v6, v7, t4, t5, t6, t7)
      : [r0] "+v"(r0), [r1] "+v"(r1), [r2] "+v"(r2), [r3] "+v"(r3), [v4] "+v"(q4), [v5] "+v"(q5), [v6] "+v"(q6),
        [v7] "+v"(q7), [a0] "=&v"(a0), [m1] "=&v"(m1), [m2] "=&v"(m2), [m3] "=&v"(m3), [t4] "=&v"(t4),
        [t5] "=&v"(t5), [t6] "=&v"(t6), [t7] "=&v"(t7)
      : [w1] "v"(w[0]), [w2] "v"(w[1]), [w3] "v"(w[2]), [w4] "v"(w[3]), [w5] "v"(w[4]), [w6] "v"(w[5]),
        [w7] "v"(w[6]), [two] "v"(two), [cc] "v"(cc));
  v[0] = a0; v[4] = r0;
  v[1] = r1; v[5] = m1;
  v[2] = r2; v[6] = m2;
  v[3] = r3; v[7] = m3;
}

// Real-input post-process for the four bin pairs of a thread: on entry z[i] = Z_k,
// zp[i] = Z_{M-k}; on exit z[i] = X_k and zp[i] = conj X_{M-k}.  20 instructions.
//   E = Z_k + conj Z_{M-k}, D = Z_k - conj Z_{M-k}, X_k = E - i W D, conj X_{M-k} = 2E - X_k.
__device__ __forceinline__ void real_post_blk(f2 (&z)[8], f2 (&zp)[4], const f2 (&w)[4], const f2 two) {
  f2 e0, e1, e2, e3;
  asm("v_pk_add_f32 %[e0], %[z0], %[p0] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[e1], %[z1], %[p1] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[e2], %[z2], %[p2] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[e3], %[z3], %[p3] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %[p0], %[z0], %[p0] neg_lo:[0,1]\n\t"             // D (in place on zp)
      "v_pk_add_f32 %[p1], %[z1], %[p1] neg_lo:[0,1]\n\t"
      "v_pk_add_f32 %[p2], %[z2], %[p2] neg_lo:[0,1]\n\t"
      "v_pk_add_f32 %[p3], %[z3], %[p3] neg_lo:[0,1]\n\t"
      "v_pk_fma_f32 %[z0], %[p0], %[w0], %[e0] " SG_FMA_MI_X "\n\t"   // E + (-i D) Re W
      "v_pk_fma_f32 %[z1], %[p1], %[w1], %[e1] " SG_FMA_MI_X "\n\t"
      "v_pk_fma_f32 %[z2], %[p2], %[w2], %[e2] " SG_FMA_MI_X "\n\t"
      "v_pk_fma_f32 %[z3], %[p3], %[w3], %[e3] " SG_FMA_MI_X "\n\t"
      "v_pk_fma_f32 %[z0], %[p0], %[w0], %[z0] " SG_FMA_Y "\n\t"      // ... + D Im W = X_k
      "v_pk_fma_f32 %[z1], %[p1], %[w1], %[z1] " SG_FMA_Y "\n\t"
      "v_pk_fma_f32 %[z2], %[p2], %[w2], %[z2] " SG_FMA_Y "\n\t"
      "v_pk_fma_f32 %[z3], %[p3], %[w3], %[z3] " SG_FMA_Y "\n\t"
      "v_pk_fma_f32 %[p0], %[e0], %[two], %[z0] " SG_FMS "\n\t"       // conj X_{M-k} = 2E - X_k
      "v_pk_fma_f32 %[p1], %[e1], %[two], %[z1] " SG_FMS "\n\t"
      "v_pk_fma_f32 %[p2], %[e2], %[two], %[z2] " SG_FMS "\n\t"
      "v_pk_fma_f32 %[p3], %[e3], %[two], %[z3] " SG_FMS
      : [z0] "+v"(z[0]), [z1] "+v"(z[1]), [z2] "+v"(z[2]), [z3] "+v"(z[3]), [p0] "+v"(zp[0]), [p1] "+v"(zp[1]),
        [p2] "+v"(zp[2]), [p3] "+v"(zp[3]), [e0] "=&v"(e0), [e1] "=&v"(e1), [e2] "=&v"(e2), [e3] "=&v"(e3)
      : [w0] "v"(w[0]), [w1] "v"(w[1]), [w2] "v"(w[2]), [w3] "v"(w[3]), [two] "v"(two));
}

}  // namespace sg
