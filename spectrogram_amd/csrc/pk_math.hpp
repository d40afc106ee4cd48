// pk_math.hpp -- complex arithmetic on gfx950 packed-f32 VALU ops (v_pk_{add,mul,fma}_f32).
//
// A complex number is an f2 in an even-aligned VGPR pair (.x = re, .y = im).  VOP3P source
// modifiers do the component swaps and sign flips for free:
//   op_sel[i]    which half of source i feeds the LOW  result (0 = .x, 1 = .y), default 0
//   op_sel_hi[i] which half of source i feeds the HIGH result,                 default 1
//   neg_lo[i] / neg_hi[i] negate source i for the low / high result.
// hipcc does not fold swaps into these modifiers (it emits v_mov / v_xor around the packed
// op), and per-wave issue cadence is ~4-5 cycles per instruction packed or not, so one
// packed op with modifiers instead of 2-4 scalar ops is what shortens a frame's critical
// path.  Each helper is ONE instruction, non-volatile asm: the compiler still schedules,
// CSEs and dead-strips them.  tools/pktest.hip checks every helper against the scalar
// formula on the GPU.
#pragma once
#include <hip/hip_runtime.h>

namespace sg {

typedef float f2 __attribute__((ext_vector_type(2)));

#define SG_PK2(name, insn, mods)                                   \
  __device__ __forceinline__ f2 name(f2 a, f2 b) {                 \
    f2 d;                                                          \
    asm(insn " %0, %1, %2 " mods : "=v"(d) : "v"(a), "v"(b));      \
    return d;                                                      \
  }
#define SG_PK3(name, mods)                                                     \
  __device__ __forceinline__ f2 name(f2 a, f2 b, f2 c) {                       \
    f2 d;                                                                      \
    asm("v_pk_fma_f32 %0, %1, %2, %3 " mods : "=v"(d) : "v"(a), "v"(b), "v"(c)); \
    return d;                                                                  \
  }

// a + (-i) b = (a.x + b.y, a.y - b.x)
SG_PK2(pk_add_mi, "v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")
// a + (+i) b = (a.x - b.y, a.y + b.x)
SG_PK2(pk_add_pi, "v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")
// a + conj b = (a.x + b.x, a.y - b.y)
SG_PK2(pk_add_cj, "v_pk_add_f32", "neg_hi:[0,1]")
// a - conj b = (a.x - b.x, a.y + b.y)
SG_PK2(pk_sub_cj, "v_pk_add_f32", "neg_lo:[0,1]")
// a * Re(w) = (a.x w.x, a.y w.x)
SG_PK2(pk_mul_x, "v_pk_mul_f32", "op_sel:[0,0] op_sel_hi:[1,0]")
// a Re(w) + c
SG_PK3(pk_fma_x, "op_sel:[0,0,0] op_sel_hi:[1,0,1]")
// -a Re(w) + c
SG_PK3(pk_fnma_x, "op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]")
// a Im(w) + c = (a.x w.y + c.x, a.y w.y + c.y)
SG_PK3(pk_fma_y, "op_sel:[0,1,0] op_sel_hi:[1,1,1]")
// i a Im(w) + c = (-a.y w.y + c.x, a.x w.y + c.y)
SG_PK3(pk_fma_iy, "op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]")
// (-i a) Re(w) + c = (a.y w.x + c.x, -a.x w.x + c.y)
SG_PK3(pk_fma_mi_x, "op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[1,0,0]")
// (+i a) Re(w) + c = (-a.y w.x + c.x, a.x w.x + c.y)
SG_PK3(pk_fma_pi_x, "op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0]")
// a b - c (component-wise): used as 2 t - a with b = (2, 2)
SG_PK3(pk_fms, "neg_lo:[0,0,1] neg_hi:[0,0,1]")

// a w (complex) and a w + c: the two dependent instructions as ONE asm statement.  Left as two statements the
// compiler's hazard recognizer puts an s_nop between them whenever they end up adjacent (it must assume an inline-asm
// producer wrote with a dst_sel, which needs a wait state before the next reader; packed f32 ops have no such hazard).
__device__ __forceinline__ f2 pk_cmul(f2 a, f2 w) {
  f2 d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
      : "=&v"(d) : "v"(a), "v"(w));
  return d;
}
__device__ __forceinline__ f2 pk_cfma(f2 a, f2 w, f2 c) {
  f2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
      : "=&v"(d) : "v"(a), "v"(w), "v"(c));
  return d;
}

#undef SG_PK2
#undef SG_PK3

}  // namespace sg
