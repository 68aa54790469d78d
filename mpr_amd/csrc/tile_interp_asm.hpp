/*
 * tile_interp_asm.hpp — forward walk of eval_tiles_i (reference src/context.cu:236-279, the
 * interval clause loop) as a threaded interpreter in gfx950 assembly, for k_eval_tiles.
 *
 * Same construction as the float pass's interpreter (kernels_voxel_asm.hip, read that header
 * first): 63-clause blocks, clause words rewritten per block to {out slot, handler index, lhs,
 * rhs}, handler address computed on the scalar unit, three handler tables (operands from LDS /
 * lhs forwarded / rhs forwarded from the previous clause's result registers).  Differences:
 *   - a slot holds an interval: lo plane at s * 512 + lane * 4, hi plane 256 bytes further
 *     (ds_read2st64_b32 / ds_write2st64_b32 move both with one instruction; the slot bytes of the
 *     rewritten clause word are pre-doubled so that one v_perm_b32 still forms the address:
 *     nslots <= 128, the reference's own limit);
 *   - arithmetic is the interval arithmetic of device_math.hpp in round-up mode (lower bounds by
 *     negation), instruction for instruction what the compiler makes of those functions;
 *   - min / max record their choice masks (lanes that chose lhs / rhs) in LDS and in the running
 *     "any choice" mask, exactly like the compiled loop;
 *   - sqrt, division, exp, log and the trigonometric intervals (double-precision libm inside) leave
 *     the block, are evaluated by device_math.hpp and re-enter.
 * The wave must be in round-up mode and have all 64 lanes enabled.
 */
#pragma once
#include "kernel_common.hpp"

namespace mprk {

template <uint32_t OP>
__device__ __noinline__ float2 rare_interval_op(float2 l, float2 r, float imm)
{
    int c = 0;
    const ival o = interval_clause(OP, iv(l.x, l.y), iv(r.x, r.y), imm, c);
    return make_float2(o.lo, o.hi);
}
/* op is wave-uniform: the switch runs on the scalar unit, each case is one out-of-line routine */
DEV float2 rare_interval(uint32_t op, float2 l, float2 r, float imm)
{
    switch (op) {
        case MPR_OP_SQRT_LHS: return rare_interval_op<MPR_OP_SQRT_LHS>(l, r, imm);
        case MPR_OP_SIN_LHS: return rare_interval_op<MPR_OP_SIN_LHS>(l, r, imm);
        case MPR_OP_COS_LHS: return rare_interval_op<MPR_OP_COS_LHS>(l, r, imm);
        case MPR_OP_ASIN_LHS: return rare_interval_op<MPR_OP_ASIN_LHS>(l, r, imm);
        case MPR_OP_ACOS_LHS: return rare_interval_op<MPR_OP_ACOS_LHS>(l, r, imm);
        case MPR_OP_ATAN_LHS: return rare_interval_op<MPR_OP_ATAN_LHS>(l, r, imm);
        case MPR_OP_EXP_LHS: return rare_interval_op<MPR_OP_EXP_LHS>(l, r, imm);
        case MPR_OP_LOG_LHS: return rare_interval_op<MPR_OP_LOG_LHS>(l, r, imm);
        case MPR_OP_DIV_LHS_IMM: return rare_interval_op<MPR_OP_DIV_LHS_IMM>(l, r, imm);
        case MPR_OP_DIV_IMM_RHS: return rare_interval_op<MPR_OP_DIV_IMM_RHS>(l, r, imm);
        case MPR_OP_DIV_LHS_RHS: return rare_interval_op<MPR_OP_DIV_LHS_RHS>(l, r, imm);
        default: return rare_interval_op<MPR_OP_COUNT>(l, r, imm);      /* not an opcode: NaN */
    }
}

/* The same routines under fixed symbol names: the assembly walk calls them with s_swappc_b64 (TI_CALL)
 * instead of leaving to C++ and coming back, which cost ~650 cycles per clause (10% of the clauses of
 * the involute gears are acos / atan, 7% of bear's exp / log).  Leaf functions of the AMDGPU calling
 * convention: operand in v0 (lo), v1 (hi), result in v0, v1, return address s[30:31]; they use
 * v0..v39, v48..v55, v64..v69 and s0..s31 at most (checked against the compiler's resource remarks) and
 * keep v40..v47, v56..v63 and s34 and up, where the walk's own state lives. */
#define MPR_TI_NAMED(NAME, OP)                                                                        \
    static __device__ __attribute__((noinline, used)) float2 ti_named_##NAME(float2 l) __asm__("mpr_ti_" #NAME);   \
    static __device__ float2 ti_named_##NAME(float2 l)                                                \
    {                                                                                                 \
        int c = 0;                                                                                    \
        const ival o = interval_clause(OP, iv(l.x, l.y), iv(0.0f, 0.0f), 0.0f, c);                    \
        return make_float2(o.lo, o.hi);                                                               \
    }
MPR_TI_NAMED(asin, MPR_OP_ASIN_LHS)
MPR_TI_NAMED(acos, MPR_OP_ACOS_LHS)
MPR_TI_NAMED(atan, MPR_OP_ATAN_LHS)
MPR_TI_NAMED(exp, MPR_OP_EXP_LHS)
MPR_TI_NAMED(log, MPR_OP_LOG_LHS)
#undef MPR_TI_NAMED
/* the general directed quotients / roots (device_math.hpp: round-to-nearest sandwich + residual, double
 * precision for operands below 2^-90), called by the assembly walk when some lane is at the edges of the
 * format: {RD(a1 / b1), RU(a2 / b2)} and {RD(sqrt(a)), RU(sqrt(b))}; arguments in v0..v3, result in v0, v1 */
static __device__ __attribute__((noinline, used)) float2 ti_named_divx(float a1, float b1, float a2, float b2) __asm__("mpr_ti_divx");
static __device__ float2 ti_named_divx(float a1, float b1, float a2, float b2)
{
    const ival o = div_dir(a1, b1, a2, b2);
    return make_float2(o.lo, o.hi);
}
static __device__ __attribute__((noinline, used)) float2 ti_named_sqrtx(float a, float b) __asm__("mpr_ti_sqrtx");
static __device__ float2 ti_named_sqrtx(float a, float b)
{
    const ival o = sqrt_dir(a, b);
    return make_float2(o.lo, o.hi);
}

/* Fixed registers (declared as clobbers):
 *   s[80:81] handler address   s[82:83] table base   s[84:85] block address   s86 clause word
 *   s87 immediate   s88 clause counter in block   s89 block base   s96 0xff00
 *   s[72:73] alive lanes   s74 next choice entry (LDS address)   s75 end of the choice array
 *   s[76:77] any-choice mask   s78 choice count   s79 words fetched
 *   s[40:59], s[92:95] scratch masks
 *   v32 aA  v33 aB  v34 aO   v[36:37] A (lo, hi)   v[38:39] B   v[40:41] result (and previous result)
 *   v42..v49 temporaries                                                                        */
/* next clause: word, handler address ... */
#define TI_PREP                                        \
    "s_add_u32 s88, s88, 1\n"                          \
    "v_readlane_b32 s86, %[blo], s88\n"                \
    "s_and_b32 s80, s86, s96\n"                        \
    "s_add_u32 s80, s80, s82\n"                        \
    "s_addc_u32 s81, s83, 0\n"
/* ... and go.  Handlers run TI_PREP right after issuing their LDS reads, under the reads' latency
 * (this walk is bound by the latency of its dependent instruction chain, not by issue) */
#define TI_GO "s_setpc_b64 s[80:81]\n"
#define TI_DISPATCH TI_PREP TI_GO
#define TI_IMM "v_readlane_b32 s87, %[bhi], s88\n"
#define TI_AL "v_perm_b32 v32, s86, %[lb], %[selL]\n ds_read2st64_b32 v[36:37], v32 offset1:1\n"
#define TI_AR "v_perm_b32 v33, s86, %[lb], %[selR]\n ds_read2st64_b32 v[38:39], v33 offset1:1\n"
#define TI_FL "v_mov_b32 v36, v40\n v_mov_b32 v37, v41\n"      /* lhs = previous result */
#define TI_FR "v_mov_b32 v38, v40\n v_mov_b32 v39, v41\n"      /* rhs = previous result */
#define TI_AO "v_perm_b32 v34, s86, %[lb], %[selO]\n"
#define TI_W "s_waitcnt lgkmcnt(0)\n"
#define TI_ST "ds_write2st64_b32 v34, v40, v41 offset1:1\n"
#define TI_H(v, n) ".p2align 8\nL_t" #v "_" #n "_%=:\n"
#define TI_EXIT TI_IMM "s_branch L_exit_%=\n"
#define TI_H30 TI_EXIT                /* a word that is not an opcode: evaluated (to NaN) outside the block */
#define TI_VS_ENTER
#define TI_VS_LEAVE
#define TI_END TI_ST TI_GO
#define TI_NEGLO "v_xor_b32 v40, 0x80000000, v40\n"
/* call a compiled routine on v[36:37]; v34 (address of the out slot) survives in v42 */
#define TI_CALL(sym)                                                                             \
    "v_mov_b32 v42, v34\n v_mov_b32 v0, v36\n v_mov_b32 v1, v37\n"                               \
    "s_getpc_b64 s[40:41]\n"                                                                     \
    "s_add_u32 s40, s40, " sym "@rel32@lo+4\n"                                                   \
    "s_addc_u32 s41, s41, " sym "@rel32@hi+12\n"                                                 \
    "s_swappc_b64 s[30:31], s[40:41]\n"                                                          \
    "v_mov_b32 v40, v0\n v_mov_b32 v41, v1\n v_mov_b32 v34, v42\n" TI_END

/* LDL / LDR: bring lhs into v[36:37] / rhs into v[38:39] (load, or copy of the previous result);
 * WL / WR / WLR: s_waitcnt when lhs / rhs / either was loaded from LDS */
#define TI_TABLE(v, LDL, LDR, WL, WR, WLR)                                                                  \
    TI_H(v, 0) "s_add_u32 s79, s79, s88\n s_add_u32 s79, s79, 1\n s_branch L_exit_%=\n"    /* end of tape */ \
    TI_H(v, 1) TI_IMM                                                  /* JUMP: base += j + imm + 1 */     \
    "s_add_u32 s79, s79, s88\n s_add_u32 s79, s79, 1\n"                                                     \
    "s_add_u32 s89, s89, s88\n s_add_u32 s89, s89, s87\n s_add_u32 s89, s89, 1\n s_branch L_load_%=\n"     \
    TI_H(v, 2) LDL TI_AO TI_PREP WL "s_branch L_square_%=\n"                                                        \
    TI_H(v, 3) LDL TI_AO TI_PREP WL "s_branch L_isqrt_%=\n"                                                        \
    TI_H(v, 4) LDL TI_AO TI_PREP WL "v_xor_b32 v40, 0x80000000, v37\n v_xor_b32 v41, 0x80000000, v36\n" TI_END     \
    /* i_sin / i_cos are the constant interval [-1, 1] (inc/gpu_interval.hpp:346-380) */                     \
    TI_H(v, 5) TI_AO TI_PREP "v_mov_b32 v40, -1.0\n v_mov_b32 v41, 1.0\n" TI_END                            \
    TI_H(v, 6) TI_AO TI_PREP "v_mov_b32 v40, -1.0\n v_mov_b32 v41, 1.0\n" TI_END                            \
    TI_H(v, 7) LDL TI_AO TI_PREP WL "s_branch L_casin_%=\n"                                                \
    TI_H(v, 8) LDL TI_AO TI_PREP WL "s_branch L_cacos_%=\n"                                                \
    TI_H(v, 9) LDL TI_AO TI_PREP WL "s_branch L_catan_%=\n"                                                \
    TI_H(v, 10) LDL TI_AO TI_PREP WL "s_branch L_cexp_%=\n"                                                \
    TI_H(v, 11) LDL TI_AO TI_PREP WL "s_branch L_abs_%=\n"                                                          \
    TI_H(v, 12) LDL TI_AO TI_PREP WL "s_branch L_clog_%=\n"                                                \
    TI_H(v, 13) TI_IMM LDL TI_AO TI_PREP WL                                                    /* ADD_LHS_IMM */    \
    "v_add_f32_e64 v40, -v36, -s87\n v_add_f32 v41, s87, v37\n" TI_NEGLO TI_END                            \
    TI_H(v, 14) LDL LDR TI_AO TI_PREP WLR                                                      /* ADD_LHS_RHS */    \
    "v_add_f32_e64 v40, -v36, -v38\n v_add_f32 v41, v37, v39\n" TI_NEGLO TI_END                            \
    TI_H(v, 15) TI_IMM LDL TI_AO TI_PREP WL                                                    /* MUL_LHS_IMM */    \
    "v_mov_b32 v42, s87\n v_cmp_gt_f32 vcc, 0, v42\n s_nop 1\n"                                            \
    "v_cndmask_b32 v43, v36, v37, vcc\n v_cndmask_b32 v44, v37, v36, vcc\n"                                \
    "v_mul_f32_e64 v40, -v43, v42\n v_mul_f32 v41, v44, v42\n" TI_NEGLO TI_END                             \
    TI_H(v, 16) LDL LDR TI_AO TI_PREP WLR "s_branch L_mul_%=\n"                                                     \
    TI_H(v, 17) TI_IMM LDL TI_AO TI_PREP WL "v_mov_b32 v38, s87\n v_mov_b32 v39, s87\n s_branch L_min_%=\n"        \
    TI_H(v, 18) LDL LDR TI_AO TI_PREP WLR "s_branch L_min_%=\n"                                                     \
    TI_H(v, 19) TI_IMM LDL TI_AO TI_PREP WL "v_mov_b32 v38, s87\n v_mov_b32 v39, s87\n s_branch L_max_%=\n"        \
    TI_H(v, 20) LDL LDR TI_AO TI_PREP WLR "s_branch L_max_%=\n"                                                     \
    TI_H(v, 21) TI_IMM LDL TI_AO TI_PREP WL                                                    /* lhs - imm */      \
    "v_sub_f32 v40, s87, v36\n v_subrev_f32 v41, s87, v37\n" TI_NEGLO TI_END                               \
    TI_H(v, 22) TI_IMM LDR TI_AO TI_PREP WR                                                    /* imm - rhs */      \
    "v_subrev_f32 v40, s87, v39\n v_sub_f32 v41, s87, v38\n" TI_NEGLO TI_END                               \
    TI_H(v, 23) LDL LDR TI_AO TI_PREP WLR                                                      /* lhs - rhs */      \
    "v_sub_f32 v40, v39, v36\n v_sub_f32 v41, v37, v38\n" TI_NEGLO TI_END                                  \
    /* division: s[58:59] = lanes whose divisor contains zero (result [-inf, inf]) */                         \
    TI_H(v, 24) TI_IMM LDL TI_AO TI_PREP WL                                            /* lhs / imm */      \
    "v_mov_b32 v38, s87\n v_mov_b32 v39, s87\n v_cmp_lg_f32 s[58:59], 0, v38\n s_nop 0\n"                   \
    "s_not_b64 s[58:59], s[58:59]\n s_branch L_idiv_%=\n"                                                   \
    TI_H(v, 25) TI_IMM LDR TI_AO TI_PREP WR                                            /* imm / rhs */      \
    "v_mov_b32 v36, s87\n v_mov_b32 v37, s87\n v_cmp_ge_f32 s[58:59], 0, v38\n v_cmp_le_f32 vcc, 0, v39\n"  \
    "s_and_b64 s[58:59], s[58:59], vcc\n s_branch L_idiv_%=\n"                                              \
    TI_H(v, 26) LDL LDR TI_AO TI_PREP WLR                                              /* lhs / rhs */      \
    "v_cmp_ge_f32 s[58:59], 0, v38\n v_cmp_le_f32 vcc, 0, v39\n"                                            \
    "s_and_b64 s[58:59], s[58:59], vcc\n s_branch L_idiv_%=\n"                                              \
    TI_H(v, 27) TI_IMM TI_AO TI_PREP "s_nop 0\n v_mov_b32 v40, s87\n v_mov_b32 v41, s87\n" TI_END                  \
    TI_H(v, 28) LDL TI_AO TI_PREP WL "v_mov_b32 v40, v36\n v_mov_b32 v41, v37\n" TI_END                            \
    TI_H(v, 29) LDR TI_AO TI_PREP WR "v_mov_b32 v40, v38\n v_mov_b32 v41, v39\n" TI_END                            \
    TI_H(v, 30) TI_H30                                                                                      \
    TI_H(v, 31) "s_add_u32 s79, s79, 63\n s_add_u32 s89, s89, 63\n s_branch L_load_%=\n"   /* lane 63: next block */


/* The interval routines the handlers branch to (operands v[36:37], v[38:39], result v[40:41], ending in TI_END): shared by the two
 * interpreters below and, as subroutines of generated code, by tile_gen_asm.hpp (TI_END = a return there). */
#define TI_BODIES_TEXT \
    /* ---- i_square(v[36:37]) (device_math.hpp) ---- */ \
    ".p2align 8\n" \
    "L_square_%=:\n" \
    "v_cmp_lt_f32 s[92:93], v37, -v36\n"             /* big: -lo > hi */ \
    "v_mul_f32 v42, v36, v36\n"                      /* a = RU(lo * lo) */ \
    "v_mul_f32 v43, v37, v37\n"                      /* b = RU(hi * hi) */ \
    "v_mul_f32_e64 v44, -v36, v36\n"                 /* -c, c = RD(lo * lo) */ \
    "v_cmp_lt_f32 vcc, 0, v36\n"                     /* pos */ \
    "v_cmp_gt_f32 s[94:95], 0, v37\n"                /* neg */ \
    "v_cndmask_b32 v45, v43, v42, s[92:93]\n" \
    "v_cndmask_b32 v40, 0, -v44, vcc\n" \
    "v_cndmask_b32 v45, v45, v43, vcc\n" \
    "v_mul_f32_e64 v46, -v37, v37\n"                 /* -d, d = RD(hi * hi) */ \
    "v_cndmask_b32 v41, v45, v42, s[94:95]\n" \
    "v_cndmask_b32 v40, v40, -v46, s[94:95]\n" \
    TI_END \
    /* ---- i_abs(v[36:37]) ---- */ \
    "L_abs_%=:\n" \
    "v_max_f32 v42, v37, v37\n" \
    "v_max_f32_e64 v43, -v36, -v36\n" \
    "v_max_f32 v43, v43, v42\n"                      /* m = fmax(-lo, hi) */ \
    "v_cmp_gt_f32 vcc, 0, v37\n"                     /* neg: hi < 0 */ \
    "v_cmp_le_f32 s[92:93], 0, v36\n"                /* nonneg: lo >= 0 */ \
    "s_nop 0\n" \
    "v_cndmask_b32 v42, 0, -v37, vcc\n" \
    "v_cndmask_b32 v43, v43, -v36, vcc\n" \
    "v_cndmask_b32 v40, v42, v36, s[92:93]\n" \
    "v_cndmask_b32 v41, v43, v37, s[92:93]\n" \
    TI_END \
    /* ---- i_mul(v[36:37], v[38:39]): sign-case table without branches ---- */ \
    "L_mul_%=:\n" \
    "v_cmp_gt_f32 s[40:41], 0, v36\n"                /* xn */ \
    "v_cmp_nlt_f32 s[42:43], 0, v37\n"               /* !xp */ \
    "v_cmp_ngt_f32 s[46:47], 0, v38\n"               /* !yn */ \
    "v_cmp_lt_f32 s[50:51], 0, v39\n"                /* yp */ \
    "v_cmp_lt_f32 s[44:45], 0, v37\n"                /* xp */ \
    "s_and_b64 s[42:43], s[40:41], s[42:43]\n"       /* xN */ \
    "s_and_b64 s[58:59], s[46:47], s[50:51]\n"       /* yP */ \
    "s_or_b64 s[46:47], s[46:47], s[50:51]\n"        /* !yN */ \
    "v_cmp_ngt_f32 vcc, 0, v36\n"                    /* !xn */ \
    "v_cmp_gt_f32 s[48:49], 0, v38\n"                /* yn */ \
    "s_and_b64 s[52:53], s[40:41], s[44:45]\n"       /* xM */ \
    "s_and_b64 s[46:47], s[46:47], s[42:43]\n"       /* !yN & xN */ \
    "s_and_b64 s[54:55], vcc, s[44:45]\n"            /* xP */ \
    "s_and_b64 s[56:57], s[48:49], s[50:51]\n"       /* yM */ \
    "s_or_b64 vcc, s[58:59], s[46:47]\n"             /* p is x.lo */ \
    "s_and_b64 s[46:47], s[52:53], s[58:59]\n"       /* xM & yP */ \
    "v_cndmask_b32 v42, v37, v36, vcc\n"             /* p */ \
    "s_or_b64 vcc, s[42:43], s[46:47]\n"             /* q is y.hi */ \
    "s_and_b64 s[42:43], s[54:55], s[56:57]\n"       /* xP & yM */ \
    "v_cndmask_b32 v43, v38, v39, vcc\n"             /* q */ \
    "s_or_b64 vcc, s[58:59], s[42:43]\n"             /* r is x.hi */ \
    "v_cndmask_b32 v44, v36, v37, vcc\n"             /* r */ \
    "s_or_b64 vcc, s[54:55], s[46:47]\n"             /* s is y.hi */ \
    "v_cndmask_b32 v45, v38, v39, vcc\n"             /* s */ \
    "v_mul_f32_e64 v42, -v42, v43\n"                 /* -lo = RU(-p * q) */ \
    "v_mul_f32 v43, v44, v45\n"                      /* hi = RU(r * s) */ \
    "v_mul_f32_e64 v44, -v36, v39\n"                 /* -lo2 = RU(-x.lo * y.hi) */ \
    "v_mul_f32 v45, v37, v39\n"                      /* hi2 = RU(x.hi * y.hi) */ \
    "s_and_b64 vcc, s[52:53], s[56:57]\n"            /* M * M */ \
    "v_max_f32 v46, v42, v42\n" \
    "v_max_f32 v44, v44, v44\n" \
    "v_max_f32 v44, v44, v46\n"                      /* -min(lo2, lo) */ \
    "v_cndmask_b32 v40, v42, v44, vcc\n" \
    "v_max_f32 v44, v43, v43\n" \
    "v_max_f32 v45, v45, v45\n" \
    "v_max_f32 v45, v44, v45\n"                      /* max(hi, hi2) */ \
    "s_or_b64 s[40:41], s[40:41], s[44:45]\n"        /* xn | xp */ \
    "s_or_b64 s[42:43], s[48:49], s[50:51]\n"        /* yn | yp */ \
    "v_xor_b32 v40, 0x80000000, v40\n" \
    "v_cndmask_b32 v41, v43, v45, vcc\n" \
    "s_and_b64 vcc, s[40:41], s[42:43]\n"            /* neither operand is the zero class */ \
    "v_cndmask_b32 v40, 0, v40, vcc\n" \
    "v_cndmask_b32 v41, 0, v41, vcc\n" \
    TI_END \
    /* ---- i_min(v[36:37], v[38:39]) + choice ---- */ \
    "L_min_%=:\n" \
    "v_max_f32 v42, v38, v38\n" \
    "v_max_f32 v43, v36, v36\n" \
    "v_max_f32 v44, v39, v39\n" \
    "v_cmp_nlt_f32 vcc, v37, v38\n"                  /* !c1, c1: x.hi < y.lo */ \
    "v_cmp_gt_f32 s[92:93], v36, v39\n"              /* y.hi < x.lo */ \
    "s_and_b64 s[92:93], vcc, s[92:93]\n"            /* c2 */ \
    "v_min_f32 v42, v43, v42\n" \
    "v_max_f32 v43, v37, v37\n" \
    "v_min_f32 v43, v43, v44\n" \
    "s_branch L_choice_%=\n" \
    /* ---- i_max ---- */ \
    "L_max_%=:\n" \
    "v_max_f32 v42, v38, v38\n" \
    "v_max_f32 v43, v36, v36\n" \
    "v_max_f32 v44, v39, v39\n" \
    "v_cmp_ngt_f32 vcc, v36, v39\n"                  /* !c1, c1: x.lo > y.hi */ \
    "v_cmp_lt_f32 s[92:93], v37, v38\n"              /* y.lo > x.hi */ \
    "s_and_b64 s[92:93], vcc, s[92:93]\n"            /* c2 */ \
    "v_max_f32 v42, v43, v42\n" \
    "v_max_f32 v43, v37, v37\n" \
    "v_max_f32 v43, v43, v44\n" \
    /* result = c1 ? x : c2 ? y : (v42, v43); record {lanes that chose lhs, lanes that chose rhs} */ \
    "L_choice_%=:\n" \
    "v_cndmask_b32 v40, v42, v38, s[92:93]\n" \
    "v_cndmask_b32 v41, v43, v39, s[92:93]\n" \
    "v_cndmask_b32 v40, v36, v40, vcc\n" \
    "v_cndmask_b32 v41, v37, v41, vcc\n" \
    "s_andn2_b64 s[94:95], s[72:73], vcc\n"          /* chose lhs */ \
    "s_and_b64 s[92:93], s[92:93], s[72:73]\n"       /* chose rhs */ \
    "s_or_b64 s[76:77], s[76:77], s[94:95]\n" \
    "s_or_b64 s[76:77], s[76:77], s[92:93]\n" \
    "s_cmp_lt_u32 s74, s75\n" \
    "s_cbranch_scc0 L_nochoice_%=\n" \
    "s_mov_b64 exec, 1\n" \
    "v_mov_b32 v42, s94\n" \
    "v_mov_b32 v43, s95\n" \
    "v_mov_b32 v44, s92\n" \
    "v_mov_b32 v45, s93\n" \
    "v_mov_b32 v46, s74\n" \
    "ds_write_b128 v46, v[42:45]\n" \
    "s_mov_b64 exec, -1\n" \
    "L_nochoice_%=:\n" \
    "s_add_u32 s74, s74, 16\n" \
    "s_add_u32 s78, s78, 1\n" \
    TI_END \
    /* ---- i_div(v[36:37], v[38:39]) (device_math.hpp: operands of the two directed quotients by \
    *      selects, quotients in a round-to-nearest sandwich, moved one ulp by the residual's sign); \
    *      s[58:59]: lanes whose divisor contains zero ---- */ \
    "L_idiv_%=:\n" \
    "s_movk_i32 s56, 0x1f8\n" \
    "s_movk_i32 s57, 0x198\n" \
    "v_cmp_gt_f32 s[40:41], 0, v37\n"                /* xn: x.hi < 0 */ \
    "v_cmp_gt_f32 s[42:43], 0, v36\n"                /* x.lo < 0 */ \
    "v_cmp_gt_f32 vcc, 0, v39\n"                     /* yn: y.hi < 0 */ \
    "s_andn2_b64 s[42:43], s[42:43], s[40:41]\n"     /* xm */ \
    "s_nop 0\n" \
    "v_cndmask_b32 v42, v36, v37, vcc\n"             /* a1 = yn ? x.hi : x.lo */ \
    "v_cndmask_b32 v43, v37, v36, vcc\n"             /* a2 = yn ? x.lo : x.hi */ \
    "v_cndmask_b32 v44, v38, v39, vcc\n"             /* ym = yn ? y.hi : y.lo */ \
    "v_cndmask_b32 v45, v39, v44, s[42:43]\n" \
    "v_cndmask_b32 v45, v45, v38, s[40:41]\n"        /* b1 = xn ? y.lo : xm ? ym : y.hi */ \
    "v_cndmask_b32 v46, v38, v44, s[42:43]\n" \
    "v_cndmask_b32 v46, v46, v39, s[40:41]\n"        /* b2 = xn ? y.hi : xm ? ym : y.lo */ \
    /* Fast path, taken when every lane is away from the edges of the format — numerators zero or \
    * 2^-60 <= |a| <= 2^60, divisors 2^-30 <= |b| <= 2^30, so that quotients and residuals stay normal: \
    * reciprocal by v_rcp + one Newton step, quotient by one correction (rounded up for the lower bound, \
    * down for the upper one), then the EXACT residual r = a - q b says on which side of q the true \
    * quotient lies, and the directed result is q or its neighbour.  No switch to \
    * round-to-nearest, a third of the instructions of the general sequence below. */ \
    "v_and_b32 v51, 0x7fffffff, v45\n"               /* |b1| */ \
    "v_and_b32 v52, 0x7fffffff, v46\n"               /* |b2| */ \
    "v_and_b32 v53, 0x7fffffff, v42\n"               /* |a1| */ \
    "v_and_b32 v54, 0x7fffffff, v43\n"               /* |a2| */ \
    "v_min_u32 v55, v51, v52\n" \
    "v_max_u32 v51, v51, v52\n" \
    "s_mov_b32 s44, 0x30800000\n"                    /* 2^-30 */ \
    "s_mov_b32 s45, 0x4e800000\n"                    /* 2^30 */ \
    "v_cmp_le_u32 s[46:47], s44, v55\n" \
    "v_cmp_ge_u32 vcc, s45, v51\n" \
    "s_and_b64 s[46:47], s[46:47], vcc\n" \
    "s_mov_b32 s44, 0x21800000\n"                    /* 2^-60 */ \
    "s_mov_b32 s45, 0x5d800000\n"                    /* 2^60 */ \
    "v_max_u32 v51, v53, v54\n" \
    "v_cmp_ge_u32 vcc, s45, v51\n" \
    "s_and_b64 s[46:47], s[46:47], vcc\n" \
    "v_cmp_le_u32 s[48:49], s44, v53\n" \
    "v_cmp_eq_u32 s[50:51], 0, v53\n"               /* a1 is a zero */ \
    "s_or_b64 s[48:49], s[48:49], s[50:51]\n" \
    "s_and_b64 s[46:47], s[46:47], s[48:49]\n" \
    "v_cmp_le_u32 s[48:49], s44, v54\n" \
    "v_cmp_eq_u32 s[52:53], 0, v54\n"               /* a2 is a zero */ \
    "s_or_b64 s[48:49], s[48:49], s[52:53]\n" \
    "s_and_b64 s[46:47], s[46:47], s[48:49]\n" \
    "s_cmp_eq_u64 s[46:47], exec\n" \
    "s_cbranch_scc0 L_idiv_slow_%=\n" \
    "v_rcp_f32 v51, v45\n" \
    "v_rcp_f32 v52, v46\n" \
    "s_nop 0\n" \
    "v_fma_f32 v53, -v45, v51, 1.0\n" \
    "v_fma_f32 v54, -v46, v52, 1.0\n" \
    "v_fmac_f32 v51, v53, v51\n"                     /* 1 / b1 */ \
    "v_fmac_f32 v52, v54, v52\n"                     /* 1 / b2 */ \
    "v_mul_f32 v47, v42, v51\n" \
    "v_mul_f32 v49, v43, v52\n" \
    "v_fma_f32 v48, -v47, v45, v42\n" \
    "v_fma_f32 v50, -v49, v46, v43\n" \
    "v_fma_f32 v53, v48, v51, v47\n"                 /* q1 = RU(S1), S1 = a1 / b1 up to a fraction of an ulp */ \
    "v_fma_f32 v54, -v50, v52, -v49\n"               /* -q2 = RU(-S2): q2 = RD(S2) — each quotient errs to */ \
    "v_xor_b32 v54, 0x80000000, v54\n"               /* the side its single correction step can undo */ \
    "v_cndmask_b32 v47, v53, v47, s[50:51]\n"        /* a zero numerator: the product already is the */ \
    "v_cndmask_b32 v49, v54, v49, s[52:53]\n"        /* signed zero (the sum above would lose its sign) */ \
    "v_fma_f32 v48, -v47, v45, v42\n"                /* exact residuals */ \
    "v_fma_f32 v50, -v49, v46, v43\n" \
    /* lower bound: one ulp down when the true quotient is below q1 (r1 and b1 of opposite sign) */ \
    "v_xor_b32 v51, v48, v45\n" \
    "v_ashrrev_i32 v52, 31, v47\n" \
    "v_cmp_gt_i32 vcc, 0, v51\n" \
    "v_cmp_neq_f32 s[44:45], 0, v48\n" \
    "v_or_b32 v52, 1, v52\n"                         /* -1 for a negative q1, else 1 */ \
    "s_and_b64 vcc, vcc, s[44:45]\n" \
    "v_sub_u32 v52, v47, v52\n"                      /* next_down(q1) */ \
    "v_cndmask_b32 v40, v47, v52, vcc\n" \
    /* upper bound: one ulp up when the true quotient is above q2 (r2 and b2 of the same sign) */ \
    "v_xor_b32 v51, v50, v46\n" \
    "v_ashrrev_i32 v52, 31, v49\n" \
    "v_cmp_lt_i32 vcc, -1, v51\n" \
    "v_cmp_neq_f32 s[44:45], 0, v50\n" \
    "v_or_b32 v52, 1, v52\n" \
    "s_and_b64 vcc, vcc, s[44:45]\n" \
    "v_add_u32 v52, v49, v52\n"                      /* next_up(q2) */ \
    "v_cndmask_b32 v41, v49, v52, vcc\n" \
    "s_branch L_idiv_tail_%=\n" \
    "L_idiv_slow_%=:\n"                              /* some lane at the edges of the format: compiled routine */ \
    "v_mov_b32 v47, v34\n" \
    "v_mov_b32 v0, v42\n v_mov_b32 v1, v45\n v_mov_b32 v2, v43\n v_mov_b32 v3, v46\n" \
    "s_getpc_b64 s[40:41]\n" \
    "s_add_u32 s40, s40, mpr_ti_divx@rel32@lo+4\n" \
    "s_addc_u32 s41, s41, mpr_ti_divx@rel32@hi+12\n" \
    "s_swappc_b64 s[30:31], s[40:41]\n" \
    "v_mov_b32 v40, v0\n v_mov_b32 v41, v1\n v_mov_b32 v34, v47\n" \
    "L_idiv_tail_%=:\n" \
    "v_mov_b32 v51, 0xff800000\n" \
    "v_mov_b32 v52, 0x7f800000\n" \
    "v_cndmask_b32 v40, v40, v51, s[58:59]\n" \
    "v_cndmask_b32 v41, v41, v52, s[58:59]\n" \
    TI_END \
    /* ---- i_sqrt(v[36:37]): x.hi < 0: NaN; lower bound 0 when x.lo <= 0 ---- */ \
    "L_isqrt_%=:\n" \
    "s_movk_i32 s55, 0x260\n" \
    "s_movk_i32 s56, 0x1f8\n" \
    "v_cmp_ge_f32 vcc, 0, v36\n"                     /* x.lo <= 0 */ \
    "v_cmp_gt_f32 s[58:59], 0, v37\n"                /* x.hi < 0 */ \
    "s_nop 0\n" \
    "v_cndmask_b32 v42, v36, 0, vcc\n"               /* a */ \
    /* Fast path when every lane's operands are zero or 2^-60 <= x <= 2^60 (negative upper ends pass: \
    * their lanes are overwritten with NaN below): v_sqrt_f32 is within one ulp, so the directed \
    * roots are among y - 1 ulp, y, y + 1 ulp, told apart by the signs of x - c^2 (fma).  No switch \
    * to round-to-nearest, half the instructions of the general sequence. */ \
    "v_and_b32 v52, 0x7fffffff, v37\n" \
    "s_mov_b32 s44, 0x21800000\n"                    /* 2^-60 */ \
    "s_mov_b32 s45, 0x5d800000\n"                    /* 2^60 */ \
    "v_cmp_le_u32 s[46:47], s44, v42\n" \
    "v_cmp_eq_u32 s[50:51], 0, v42\n"               /* a is zero */ \
    "v_cmp_ge_u32 vcc, s45, v42\n" \
    "s_or_b64 s[46:47], s[46:47], s[50:51]\n" \
    "s_and_b64 s[46:47], s[46:47], vcc\n" \
    "v_cmp_le_u32 s[48:49], s44, v52\n" \
    "v_cmp_eq_u32 s[52:53], 0, v52\n"               /* x.hi is a zero */ \
    "v_cmp_ge_u32 vcc, s45, v52\n" \
    "s_or_b64 s[48:49], s[48:49], s[52:53]\n" \
    "s_and_b64 s[48:49], s[48:49], vcc\n" \
    "s_and_b64 s[46:47], s[46:47], s[48:49]\n" \
    "s_cmp_eq_u64 s[46:47], exec\n" \
    "s_cbranch_scc0 L_isqrt_slow_%=\n" \
    "v_sqrt_f32 v47, v42\n" \
    "v_sqrt_f32 v49, v37\n" \
    "s_nop 0\n" \
    "v_add_u32 v51, 1, v47\n"                        /* lower: y + 1 ulp, y - 1 ulp */ \
    "v_add_u32 v52, -1, v47\n" \
    "v_add_u32 v54, 1, v49\n"                        /* upper */ \
    "v_add_u32 v55, -1, v49\n" \
    "v_fma_f32 v48, -v47, v47, v42\n"                /* x - y^2 */ \
    "v_fma_f32 v53, -v51, v51, v42\n"                /* x - (y + 1)^2 */ \
    "v_fma_f32 v50, -v49, v49, v37\n" \
    "v_fma_f32 v43, -v55, v55, v37\n"                /* x - (y - 1)^2 */ \
    "v_cmp_le_f32 s[44:45], 0, v53\n"               /* y + 1 still not above the root */ \
    "v_cmp_gt_f32 s[46:47], 0, v48\n"               /* y above the root */ \
    "v_cmp_ge_f32 s[48:49], 0, v43\n"               /* y - 1 still not below the root */ \
    "v_cmp_lt_f32 vcc, 0, v50\n"                    /* y below the root */ \
    "v_cndmask_b32 v51, v47, v51, s[44:45]\n" \
    "v_cndmask_b32 v55, v49, v55, s[48:49]\n" \
    "v_cndmask_b32 v40, v51, v52, s[46:47]\n"        /* largest c with c^2 <= x */ \
    "v_cndmask_b32 v41, v55, v54, vcc\n"             /* smallest c with c^2 >= x */ \
    "v_cndmask_b32 v40, v40, v42, s[50:51]\n"        /* the root of a zero is that zero */ \
    "v_cndmask_b32 v41, v41, v37, s[52:53]\n" \
    "s_branch L_isqrt_tail_%=\n" \
    "L_isqrt_slow_%=:\n" \
    "v_mov_b32 v47, v34\n" \
    "v_mov_b32 v0, v42\n v_mov_b32 v1, v37\n" \
    "s_getpc_b64 s[40:41]\n" \
    "s_add_u32 s40, s40, mpr_ti_sqrtx@rel32@lo+4\n" \
    "s_addc_u32 s41, s41, mpr_ti_sqrtx@rel32@hi+12\n" \
    "s_swappc_b64 s[30:31], s[40:41]\n" \
    "v_mov_b32 v40, v0\n v_mov_b32 v41, v1\n v_mov_b32 v34, v47\n" \
    "L_isqrt_tail_%=:\n" \
    "v_mov_b32 v51, 0x7fc00000\n" \
    "v_cndmask_b32 v40, v40, v51, s[58:59]\n" \
    "v_cndmask_b32 v41, v41, v51, s[58:59]\n" \
    TI_END \
    /* ---- compiled routines (double precision inside): called, not left for ---- */ \
    "L_casin_%=:\n" TI_CALL("mpr_ti_asin") \
    "L_cacos_%=:\n" TI_CALL("mpr_ti_acos") \
    "L_catan_%=:\n" TI_CALL("mpr_ti_atan") \
    "L_cexp_%=:\n" TI_CALL("mpr_ti_exp") \
    "L_clog_%=:\n" TI_CALL("mpr_ti_log")

/* The walk itself, as text: expanded in tile_interp_asm (slots in LDS) and in tile_interp_asm_vgpr (slots in VGPRs),
 * each time with that variant's definitions of TI_AL / TI_AR / TI_AO / TI_ST / TI_H30 / TI_VS_ENTER / TI_VS_LEAVE. */
#define TI_ASM_TEXT \
    TI_VS_ENTER \
    "s_mov_b32 s89, %[base]\n" \
    "s_mov_b32 s88, %[sj]\n" \
    "s_mov_b32 s96, 0xff00\n" \
    "s_mov_b32 s72, %[alo]\n" \
    "s_mov_b32 s73, %[ahi]\n" \
    "s_mov_b32 s74, %[caddr]\n" \
    "s_mov_b32 s75, %[cend]\n" \
    "s_mov_b32 s76, %[anylo]\n" \
    "s_mov_b32 s77, %[anyhi]\n" \
    "s_mov_b32 s78, %[ci]\n" \
    "s_mov_b32 s79, %[words]\n" \
    "v_mov_b32 v40, %[plo]\n" \
    "v_mov_b32 v41, %[phi]\n" \
    "s_getpc_b64 s[82:83]\n" \
    "L_pc_%=:\n" \
    "s_add_u32 s82, s82, L_t0_0_%=-L_pc_%=\n" \
    "s_addc_u32 s83, s83, 0\n" \
    "s_cmp_eq_u32 %[mode], 0\n" \
    "s_cbranch_scc1 L_load_%=\n" \
    "s_cmp_eq_u32 %[mode], 2\n" \
    "s_cbranch_scc1 L_loaded_%=\n" \
    TI_DISPATCH \
    /* ---- fetch 63 clauses at s89, rewrite the clause words ---- */ \
    "L_load_%=:\n" \
    "s_mov_b32 s84, s89\n" \
    "s_mov_b32 s85, 0\n" \
    "s_lshl_b64 s[84:85], s[84:85], 3\n" \
    "s_add_u32 s84, s84, %[tlo]\n" \
    "s_addc_u32 s85, s85, %[thi]\n" \
    "global_load_dword %[blo], %[lane8], s[84:85]\n" \
    "global_load_dword %[bhi], %[lane8], s[84:85] offset:4\n" \
    "L_loaded_%=:\n" \
    "s_mov_b32 s88, -1\n" \
    "v_mov_b32 v45, 0\n" \
    "v_mov_b32 v47, 32\n" \
    "v_mov_b32 v48, 64\n" \
    "s_waitcnt vmcnt(0)\n" \
    "v_bfe_u32 v44, %[blo], 8, 8\n"                 /* out slot */ \
    "v_and_b32 v42, 0xff, %[blo]\n" \
    "v_min_u32 v42, 30, v42\n"                       /* opcode; unknown ones -> handler 30 */ \
    "v_mov_b32_dpp v45, v44 wave_shr:1 row_mask:0xf bank_mask:0xf\n"   /* out slot of the previous clause (lane 0: none) */ \
    "v_bfe_u32 v46, %[blo], 16, 8\n"                /* lhs slot */ \
    "v_lshrrev_b32 v43, 24, %[blo]\n"               /* rhs slot */ \
    "v_cmp_eq_u32 s[92:93], v43, v45\n" \
    "v_cmp_eq_u32 vcc, v46, v45\n" \
    "v_cmp_ne_u32 s[94:95], 0, v45\n" \
    "v_cndmask_b32 v46, 0, v48, s[92:93]\n"          /* rhs forwarded: table 2 */ \
    "v_cndmask_b32 v46, v46, v47, vcc\n"             /* lhs forwarded: table 1 */ \
    "v_cndmask_b32 v46, 0, v46, s[94:95]\n" \
    "v_cmp_eq_u32 vcc, 0x1f8, %[lane8]\n"           /* lane 63 -> handler 31 of table 0 */ \
    "v_mov_b32 v43, 31\n" \
    "v_add_u32 v42, v42, v46\n" \
    "v_cndmask_b32 v42, v42, v43, vcc\n"             /* handler index = table * 32 + opcode */ \
    /* clause word as the handlers see it: byte 0 = 2 * out slot, byte 1 = handler index, \
    * bytes 2, 3 = 2 * lhs, 2 * rhs (slots < 128) */ \
    "v_lshlrev_b32 v44, 1, v44\n" \
    "v_lshl_or_b32 v42, v42, 8, v44\n" \
    "v_and_b32 %[blo], 0xffff0000, %[blo]\n" \
    "v_lshlrev_b32 %[blo], 1, %[blo]\n" \
    "v_or_b32 %[blo], %[blo], v42\n" \
    "s_nop 0\n" \
    TI_DISPATCH \
    /* ---- handlers: three tables of 32 x 256 bytes ---- */ \
    TI_TABLE(0, TI_AL, TI_AR, TI_W, TI_W, TI_W) \
    TI_TABLE(1, TI_FL, TI_AR, "", TI_W, TI_W) \
    TI_TABLE(2, TI_AL, TI_FR, TI_W, "", TI_W) \
    TI_BODIES_TEXT \
    /* ---- leave: end of tape, or an opcode evaluated in C++ ---- */ \
    "L_exit_%=:\n" \
    "s_waitcnt lgkmcnt(0)\n" \
    TI_VS_LEAVE \
    "s_mov_b32 %[dlo], s86\n" \
    "s_mov_b32 %[dhi], s87\n" \
    "s_mov_b32 %[base], s89\n" \
    "s_mov_b32 %[sj], s88\n" \
    "s_mov_b32 %[caddr], s74\n" \
    "s_mov_b32 %[anylo], s76\n" \
    "s_mov_b32 %[anyhi], s77\n" \
    "s_mov_b32 %[ci], s78\n" \
    "s_mov_b32 %[words], s79\n"

struct TileInterpResult {
    uint32_t result_slot;     /* slot named by the end clause */
    int nchoices;             /* min/max clauses met (= entries of the choice array, capped by the caller) */
    uint64_t any_choice;      /* lanes that chose a side at least once */
    int words;                /* clause words fetched, jumps and end included */
    int end_index;            /* pool index of the end clause */
};

/* smem: slot planes at LDS offset 0 (see the header); choices: ulonglong2[choice_cap] at LDS byte
 * offset choice_off */
DEV TileInterpResult tile_interp_asm(const uint64_t* __restrict__ tro, uint32_t first, unsigned char* smem, int lane,
                                     uint64_t alive_mask, uint32_t choice_off, int choice_cap,
                                     const uint64_t* first_block = nullptr)
{
    float* const plane = reinterpret_cast<float*>(smem);                  /* slot s: plane[s * 128 + lane], + 64 */
    uint32_t blo = 0, bhi = 0;
    uint32_t base = first, sj = 0, dlo = 0, dhi = 0;
    const uint32_t lb = (uint32_t)(uintptr_t)smem + (uint32_t)lane * 4u;
    const uint32_t lane8 = (uint32_t)lane * 8u;
    const uint32_t selO = to_vgpr(0x0c0c0400u), selL = to_vgpr(0x0c0c0600u), selR = to_vgpr(0x0c0c0700u);
    const uint32_t tlo = (uint32_t)(uintptr_t)tro, thi = (uint32_t)((uintptr_t)tro >> 32);
    const uint32_t alo = (uint32_t)alive_mask, ahi = (uint32_t)(alive_mask >> 32);
    uint32_t caddr = (uint32_t)(uintptr_t)smem + choice_off;
    const uint32_t cend = caddr + (uint32_t)choice_cap * 16u;
    uint32_t ci = 0, words = 0, anylo = 0, anyhi = 0;
    float plo = 0.0f, phi = 0.0f;
    uint32_t mode = 0;
    if (first_block) {                 /* the caller fetched tro[first + lane] under its own prologue */
        blo = (uint32_t)*first_block;
        bhi = (uint32_t)(*first_block >> 32);
        mode = 2;
    }

    for (;;) {
        base = rdfirst(base);
        sj = rdfirst(sj);
        mode = rdfirst(mode);
        caddr = rdfirst(caddr);
        ci = rdfirst(ci);
        words = rdfirst(words);
        anylo = rdfirst(anylo);
        anyhi = rdfirst(anyhi);
        asm volatile(
            TI_ASM_TEXT
            : [blo] "+&v"(blo), [bhi] "+&v"(bhi), [base] "+&s"(base), [sj] "+&s"(sj), [dlo] "=&s"(dlo), [dhi] "=&s"(dhi),
              [caddr] "+&s"(caddr), [anylo] "+&s"(anylo), [anyhi] "+&s"(anyhi), [ci] "+&s"(ci), [words] "+&s"(words)
            : [lb] "v"(lb), [selL] "v"(selL), [selR] "v"(selR), [selO] "v"(selO), [lane8] "v"(lane8),
              [tlo] "s"(tlo), [thi] "s"(thi), [mode] "s"(mode), [plo] "v"(plo), [phi] "v"(phi),
              [alo] "s"(alo), [ahi] "s"(ahi), [cend] "s"(cend)
            : "memory", "vcc", "scc",
              "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54",
              "s55", "s56", "s57", "s58", "s59",
              "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86",
              "s87", "s88", "s89", "s92", "s93", "s94", "s95", "s96",
              "v32", "v33", "v34", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47",
              "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55",
              /* what the called routines may use on top (TI_CALL) */
              "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15",
              "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30",
              "v31", "v35", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71",
              "s0", "s1", "s2", "s3", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17", "s18", "s19",
              "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31");
        const uint32_t op = (dlo >> 8) & 31;
        if (op == 0) break;
        /* sqrt, division, exp, log, trigonometry (and anything that is not an opcode) */
        const uint32_t l2 = (dlo >> 16) & 0xFF, r2 = dlo >> 24, o2 = dlo & 0xFF;      /* 2 * slot */
        const float imm = mpr_u2f(dhi);
        const float2 A = make_float2(plane[l2 * 64 + lane], plane[l2 * 64 + 64 + lane]);
        const float2 B = r2 ? make_float2(plane[r2 * 64 + lane], plane[r2 * 64 + 64 + lane]) : make_float2(imm, imm);
        const float2 R = rare_interval(op, A, B, imm);
        plane[o2 * 64 + lane] = R.x;
        plane[o2 * 64 + 64 + lane] = R.y;
        plo = R.x;
        phi = R.y;
        mode = 1;
    }
    TileInterpResult r;
    r.result_slot = (dlo & 0xFF) >> 1;
    r.nchoices = (int)ci;
    r.any_choice = ((uint64_t)anyhi << 32) | anylo;
    r.words = (int)words;
    r.end_index = (int)(base + sj);
    return r;
}


/* ====================================================================================== */
/* The same walk with the slot file in VECTOR REGISTERS, for tapes with many slots.           */
/*                                                                                          */
/* The LDS planes cost 512 bytes per slot and wavefront: a tape with 84 slots (the involute   */
/* gears, prospero) leaves room for 3 wavefronts per CU, and a walk that is a dependent chain */
/* needs many more to keep a CU busy.  Here slot s (1..92: slot 0 is nobody's) is the register */
/* pair v[68 + 2 s], v[69 + 2 s] — v70..v253 —, addressed through the scalar GPR index          */
/* (s_set_gpr_idx_on: scripts/ubench/gpr_idx_probe.hip — gfx950 has no v_movrel): the           */
/* pre-doubled slot byte of the rewritten clause word IS the index.  256 registers per lane are */
/* 2 wavefronts per SIMD, 8 per CU, and an operand arrives in ~15 cycles instead of an LDS      */
/* round trip.  Everything else — handlers, arithmetic, choices, the compiled routines (which   */
/* stay below v70) — is the text above.  The registers have to survive from the axis intervals' */
/* arrival to the result's departure, so both happen inside the one asm statement (TI_VS_ENTER / */
/* TI_VS_LEAVE, through a little LDS: the statement names all but ten VGPRs, and its operands    */
/* have to live in those), and a word that is not an opcode gets its NaN inside the block too   */
/* (TI_H30): nothing re-enters.                                                                */
/* ====================================================================================== */
#undef TI_AL
#undef TI_AR
#undef TI_AO
#undef TI_ST
#undef TI_H30
#undef TI_VS_ENTER
#undef TI_VS_LEAVE
#define TI_VS_BASE "v68"
#define TI_VS_BASE1 "v69"
#define TI_AL "s_bfe_u32 s60, s86, 0x80010\n s_set_gpr_idx_on s60, gpr_idx(SRC0)\n v_mov_b32 v36, " TI_VS_BASE "\n v_mov_b32 v37, " TI_VS_BASE1 "\n s_set_gpr_idx_off\n"
#define TI_AR "s_lshr_b32 s60, s86, 24\n s_set_gpr_idx_on s60, gpr_idx(SRC0)\n v_mov_b32 v38, " TI_VS_BASE "\n v_mov_b32 v39, " TI_VS_BASE1 "\n s_set_gpr_idx_off\n"
#define TI_AO "s_and_b32 s61, s86, 0xff\n"                       /* 2 * out slot, kept for TI_ST (TI_PREP replaces s86) */
#define TI_ST "s_set_gpr_idx_on s61, gpr_idx(DST)\n v_mov_b32 " TI_VS_BASE ", v40\n v_mov_b32 " TI_VS_BASE1 ", v41\n s_set_gpr_idx_off\n"
#define TI_H30 TI_AO TI_PREP "v_mov_b32 v40, 0x7fc00000\n v_mov_b32 v41, 0x7fc00000\n" TI_END
/* the axis intervals arrive, and the result leaves, through 2 KB of LDS behind the choices ([8][64] floats at %[io]):
 * every VGPR operand of this statement has to live in the ten registers it does not name */
#define TI_VS_ENTER                                                                                          \
    "v_lshrrev_b32 v32, 1, %[lane8]\n v_add_u32 v32, %[io], v32\n"                                           \
    "ds_read_b32 v36, v32\n ds_read_b32 v37, v32 offset:256\n ds_read_b32 v38, v32 offset:512\n"            \
    "ds_read_b32 v39, v32 offset:768\n ds_read_b32 v42, v32 offset:1024\n ds_read_b32 v43, v32 offset:1280\n" \
    "s_waitcnt lgkmcnt(0)\n"                                                                                 \
    "s_set_gpr_idx_on %[ax], gpr_idx(DST)\n v_mov_b32 " TI_VS_BASE ", v36\n v_mov_b32 " TI_VS_BASE1 ", v37\n s_set_gpr_idx_off\n" \
    "s_set_gpr_idx_on %[ay], gpr_idx(DST)\n v_mov_b32 " TI_VS_BASE ", v38\n v_mov_b32 " TI_VS_BASE1 ", v39\n s_set_gpr_idx_off\n" \
    "s_set_gpr_idx_on %[az], gpr_idx(DST)\n v_mov_b32 " TI_VS_BASE ", v42\n v_mov_b32 " TI_VS_BASE1 ", v43\n s_set_gpr_idx_off\n"
/* at the end clause: byte 0 of the rewritten word = 2 * the result's slot */
#define TI_VS_LEAVE                                                                                          \
    "s_and_b32 s60, s86, 0xff\n s_set_gpr_idx_on s60, gpr_idx(SRC0)\n v_mov_b32 v36, " TI_VS_BASE "\n v_mov_b32 v37, " TI_VS_BASE1 "\n s_set_gpr_idx_off\n" \
    "v_lshrrev_b32 v32, 1, %[lane8]\n v_add_u32 v32, %[io], v32\n"                                           \
    "ds_write_b32 v32, v36 offset:1536\n ds_write_b32 v32, v37 offset:1792\n s_waitcnt lgkmcnt(0)\n"
#define TI_V10(a) "v" #a "0", "v" #a "1", "v" #a "2", "v" #a "3", "v" #a "4", "v" #a "5", "v" #a "6", "v" #a "7", "v" #a "8", "v" #a "9"
constexpr int TI_VS_MAX_SLOTS = 93, TI_VS_SMALL_SLOTS = 24;

/* smem: ulonglong2[choice_cap] at LDS offset 0, then [8][64] floats of scratch.  ax / ay / az: 2 * the slots of the axes
 * (head clause); x / y / z: their intervals; res: the result interval. */
template <int NS>          /* registers for this many slots: 24 (v70..v117) or TI_VS_MAX_SLOTS (v70..v253) */
DEV TileInterpResult tile_interp_asm_vgpr(const uint64_t* __restrict__ tro, uint32_t first, unsigned char* smem, int lane,
                                          uint64_t alive_mask, int choice_cap, const uint64_t* first_block,
                                          uint32_t ax, uint32_t ay, uint32_t az, float2 x, float2 y, float2 z, float2* res)
{
    uint32_t blo = (uint32_t)*first_block, bhi = (uint32_t)(*first_block >> 32);
    uint32_t base = rdfirst(first), sj = 0, dlo = 0, dhi = 0;
    const uint32_t lane8 = (uint32_t)lane * 8u;
    const uint32_t tlo = (uint32_t)(uintptr_t)tro, thi = (uint32_t)((uintptr_t)tro >> 32);
    const uint32_t alo = (uint32_t)alive_mask, ahi = (uint32_t)(alive_mask >> 32);
    uint32_t caddr = rdfirst((uint32_t)(uintptr_t)smem);
    const uint32_t cend = caddr + (uint32_t)choice_cap * 16u;
    uint32_t ci = 0, words = 0, anylo = 0, anyhi = 0;
    const uint32_t mode = 2;                                   /* the caller fetched the first block */
    const uint32_t zero = 0;
    float* const io = reinterpret_cast<float*>(smem + (size_t)choice_cap * 16);
    io[lane] = x.x; io[64 + lane] = x.y; io[128 + lane] = y.x; io[192 + lane] = y.y; io[256 + lane] = z.x; io[320 + lane] = z.y;
    const uint32_t ioaddr = rdfirst((uint32_t)(uintptr_t)io);
    ax = rdfirst(ax);
    ay = rdfirst(ay);
    az = rdfirst(az);
#define TI_VS_OPERANDS \
        : [blo] "+&v"(blo), [bhi] "+&v"(bhi), [base] "+&s"(base), [sj] "+&s"(sj), [dlo] "=&s"(dlo), [dhi] "=&s"(dhi),\
          [caddr] "+&s"(caddr), [anylo] "+&s"(anylo), [anyhi] "+&s"(anyhi), [ci] "+&s"(ci), [words] "+&s"(words)\
        : [lane8] "v"(lane8), [tlo] "s"(tlo), [thi] "s"(thi), [mode] "s"(mode), [plo] "s"(zero), [phi] "s"(zero),\
          [alo] "s"(alo), [ahi] "s"(ahi), [cend] "s"(cend), [ax] "s"(ax), [ay] "s"(ay), [az] "s"(az), [io] "s"(ioaddr)
#define TI_VS_CLOBBERS \
        : "memory", "vcc", "scc",\
          "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54",\
          "s55", "s56", "s57", "s58", "s59", "s60", "s61",\
          "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86",\
          "s87", "s88", "s89", "s92", "s93", "s94", "s95", "s96",\
          "v32", "v33", "v34", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47",\
          "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55",\
          "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15",\
          "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30",\
          "v31", "v35", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71",\
          "s0", "s1", "s2", "s3", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17", "s18", "s19",\
          "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31"
    if constexpr (NS <= 24) {
        asm volatile(TI_ASM_TEXT TI_VS_OPERANDS TI_VS_CLOBBERS,
                     "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", TI_V10(8), TI_V10(9), TI_V10(10),
                     "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117");
    } else {
        asm volatile(TI_ASM_TEXT TI_VS_OPERANDS TI_VS_CLOBBERS,
                     "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", TI_V10(8), TI_V10(9), TI_V10(10), TI_V10(11),
                     TI_V10(12), TI_V10(13), TI_V10(14), TI_V10(15), TI_V10(16), TI_V10(17), TI_V10(18), TI_V10(19), TI_V10(20),
                     TI_V10(21), TI_V10(22), TI_V10(23), TI_V10(24), "v250", "v251", "v252", "v253");
    }
#undef TI_VS_OPERANDS
#undef TI_VS_CLOBBERS
    res->x = io[384 + lane];
    res->y = io[448 + lane];
    TileInterpResult r;
    r.result_slot = (dlo & 0xFF) >> 1;
    r.nchoices = (int)ci;
    r.any_choice = ((uint64_t)anyhi << 32) | anylo;
    r.words = (int)words;
    r.end_index = (int)(base + sj);
    return r;
}

/* ====================================================================================== */
/* Backward walk of tape pushing (reference src/context.cu:351-458) in assembly.            */
/*                                                                                          */
/* State: the active set is transposed as in the compiled walk — slot s is live for the     */
/* lanes (tiles) in the 64-bit mask held by lane s & 63 of VGPR pair s >> 6 — and is        */
/* updated in place with exec = 1 << lane instead of v_readlane / v_writelane round trips.  */
/* Slot 0 is nobody's output (tape_builder.cpp reserves it), so its entry absorbs the       */
/* updates of absent operands.  Words are fetched 63 at a time (lane 0 of a block is the    */
/* "fetch the previous block" handler) and classified per block on the VALU: handler index  */
/* = kind (plain / min-max / jump / head) x which VGPR pair out, lhs and rhs live in, so a  */
/* handler is straight-line code.  Per word the scalar unit does the bookkeeping on 64-bit  */
/* masks; only the offset arithmetic and the 8-byte store of a surviving clause are per     */
/* lane.  Chunks come from the lane's pre-claimed run (kernels.hip).                         */
/* ====================================================================================== */
/* s[80:81] pc  s[82:83] table  s86 classified word  s[70:71] clause  s88 j  s89 block base
 * s[72:73] live lanes  s74 choice entry address  s75 end of recorded choices  s78 choice count
 * s79 words  s[76:77] pool  s98 last pool index a chunk may start at
 * s[60:61] am  s[62:63] a1  s[64:65] a2  s[66:67] a0 / x  s[68:69] x  s[90:95] scratch
 * v32..v37 temporaries  v38 1  v[40:43] choice entry / jump words v[44:45] v[46:47]            */
/* next word: classified word, handler address ... */
#define TP_PREP                                        \
    "s_sub_u32 s88, s88, 1\n"                          \
    "v_readlane_b32 s86, %[bw], s88\n"                 \
    "s_and_b32 s80, s86, s96\n"                        \
    "s_add_u32 s80, s80, s82\n"                        \
    "s_addc_u32 s81, s83, 0\n"
/* ... and go.  The clause handlers run TP_PREP at their very start, under the latency of the active-set
 * lookup, and keep their own word and index in s99 / s97 (the walk is bound by the latency of its
 * dependent chain, like the forward one) */
#define TP_GO "s_setpc_b64 s[80:81]\n"
#define TP_DISPATCH TP_PREP TP_GO
#define TP_H(n) ".p2align 8\nL_p" #n "_%=:\n"
#define TP_H4(n) ".p2align 10\nL_p" #n "_%=:\n"        /* handlers of up to 1 KiB: index multiples of 4 */
/* am = live lanes whose out slot is active; none: next word */
#define TP_AM(POL, POH, tag)                                                    \
    "s_bfe_u32 s92, s86, 0x60000\n"                                            \
    "v_readlane_b32 s60, " POL ", s92\n"                                       \
    "v_readlane_b32 s61, " POH ", s92\n"                                       \
    "s_mov_b32 s97, s88\n"                                                     \
    "s_mov_b32 s99, s86\n"                                                     \
    TP_PREP                                                                     \
    "s_and_b64 s[60:61], s[60:61], s[72:73]\n"                                 \
    "s_cbranch_scc1 L_act" tag "_%=\n"                                         \
    TP_GO                                                                       \
    "L_act" tag "_%=:\n"                                                       \
    "v_readlane_b32 s70, %[blo], s97\n"                                        \
    "v_readlane_b32 s71, %[bhi], s97\n"
/* every lane of am takes the next word of its chunk; lanes that fill it move to the next chunk of
 * their run and write the two links (reference :384-413) or, out of room, stop pushing */
#define TP_OFFSET(tag)                                                          \
    "v_add_u32 v32, -1, %[oo]\n"                                               \
    "v_cndmask_b32 %[oo], %[oo], v32, s[60:61]\n"                              \
    "v_cmp_eq_u32 s[90:91], 0, %[oo]\n"                                        \
    "s_and_b64 s[90:91], s[90:91], s[60:61]\n"                                 \
    "s_cbranch_scc0 L_nosw" tag "_%=\n"                                        \
    "s_mov_b64 exec, s[90:91]\n"                                               \
    "v_mov_b32 v33, %[oi]\n"                                                   \
    "v_add_u32 %[oi], 64, %[oi]\n"                                             \
    "v_cmp_ge_u32 vcc, %[oi], %[rend]\n"                                       \
    "v_cmp_gt_u32 s[92:93], %[oi], s98\n"                                      \
    "s_or_b64 vcc, vcc, s[92:93]\n"                                            \
    "v_cndmask_b32 %[ovf], %[ovf], v38, vcc\n"                                 \
    "s_andn2_b64 s[72:73], s[72:73], vcc\n"                                    \
    "s_andn2_b64 s[60:61], s[60:61], vcc\n"                                    \
    "s_andn2_b64 exec, exec, vcc\n"                                            \
    "v_add_lshl_u32 v34, %[oi], 63, 3\n"                                       \
    "v_lshlrev_b32 v35, 3, v33\n"                                              \
    "global_store_dwordx2 v34, v[44:45], s[76:77]\n"                           \
    "global_store_dwordx2 v35, v[46:47], s[76:77]\n"                           \
    "v_mov_b32 %[oo], 62\n"                                                    \
    "s_mov_b64 exec, -1\n"                                                     \
    "L_nosw" tag "_%=:\n"
/* active[o] = 0; active[l] |= s[XL]; active[r] |= s[XR] */
#define TP_UPDATE(POL, POH, PLL, PLH, PRL, PRH, XL0, XL1, XR0, XR1)             \
    "s_lshl_b64 exec, 1, s92\n"                                                \
    "v_mov_b32 " POL ", 0\n v_mov_b32 " POH ", 0\n"                            \
    "s_bfe_u32 s93, s99, 0x60010\n"                                            \
    "s_lshl_b64 exec, 1, s93\n"                                                \
    "v_or_b32 " PLL ", " XL0 ", " PLL "\n v_or_b32 " PLH ", " XL1 ", " PLH "\n"  \
    "s_bfe_u32 s93, s99, 0x60018\n"                                            \
    "s_lshl_b64 exec, 1, s93\n"                                                \
    "v_or_b32 " PRL ", " XR0 ", " PRL "\n v_or_b32 " PRH ", " XR1 ", " PRH "\n"
/* a clause that is neither min nor max: live lanes keep it as it is, both operands become live */
#define TP_PLAIN(n, POL, POH, PLL, PLH, PRL, PRH)                               \
    TP_H4(n) TP_AM(POL, POH, #n) TP_OFFSET(#n)                                   \
    "s_bfe_u32 s92, s99, 0x60000\n"                                            \
    TP_UPDATE(POL, POH, PLL, PLH, PRL, PRH, "s60", "s61", "s60", "s61")        \
    "s_mov_b64 exec, s[60:61]\n"                                               \
    "v_add_lshl_u32 v34, %[oi], %[oo], 3\n"                                    \
    "v_mov_b32 v36, s70\n v_mov_b32 v37, s71\n"                                \
    "global_store_dwordx2 v34, v[36:37], s[76:77]\n"                           \
    "s_mov_b64 exec, -1\n"                                                     \
    TP_GO
/* min / max: lanes that chose a side keep only that operand and get a COPY (or nothing, when the
 * copy would be onto itself); bit 6 / 7 of byte 0: lhs == out / rhs == out, bit 6 of byte 2: no rhs */
#define TP_MINMAX(n, POL, POH, PLL, PLH, PRL, PRH)                              \
    TP_H4(n) "s_sub_u32 s78, s78, 1\n s_sub_u32 s74, s74, 16\n"                  \
    TP_AM(POL, POH, #n)                                                         \
    "s_mov_b64 s[62:63], 0\n s_mov_b64 s[64:65], 0\n"                          \
    "s_cmp_lt_u32 s74, s75\n"                                                  \
    "s_cbranch_scc0 L_noch" #n "_%=\n"                                         \
    "v_mov_b32 v32, s74\n"                                                     \
    "ds_read_b128 v[40:43], v32\n"                                             \
    "s_waitcnt lgkmcnt(0)\n"                                                   \
    "v_readfirstlane_b32 s62, v40\n v_readfirstlane_b32 s63, v41\n"            \
    "v_readfirstlane_b32 s64, v42\n v_readfirstlane_b32 s65, v43\n"            \
    "L_noch" #n "_%=:\n"                                                       \
    TP_OFFSET(#n)                                                               \
    "s_or_b64 s[66:67], s[62:63], s[64:65]\n"                                  \
    "s_andn2_b64 s[66:67], s[60:61], s[66:67]\n"          /* a0 */             \
    "s_addc_u32 s84, s84, 0\n"                          /* + (a0 != 0): a min / max some lane keeps */ \
    "s_and_b64 s[62:63], s[62:63], s[60:61]\n"            /* a1 */             \
    "s_and_b64 s[64:65], s[64:65], s[60:61]\n"            /* a2 */             \
    "s_or_b64 s[68:69], s[66:67], s[62:63]\n"             /* lhs live for a0 | a1 */ \
    "s_or_b64 s[66:67], s[66:67], s[64:65]\n"             /* rhs live for a0 | a2 */ \
    "s_bfe_u32 s92, s99, 0x60000\n"                                            \
    TP_UPDATE(POL, POH, PLL, PLH, PRL, PRH, "s68", "s69", "s66", "s67")        \
    "s_mov_b64 exec, -1\n"                                                     \
    "s_andn2_b32 s92, s70, 0xff\n"                                             \
    "s_or_b32 s93, s92, 28\n"                              /* COPY_LHS */      \
    "s_bitcmp1_b32 s99, 22\n"                                                  \
    "s_cselect_b32 s94, 27, 29\n"                          /* COPY_IMM : COPY_RHS */ \
    "s_or_b32 s94, s92, s94\n"                                                 \
    "v_mov_b32 v36, s70\n v_mov_b32 v33, s93\n v_mov_b32 v35, s94\n"           \
    "v_cndmask_b32 v36, v36, v33, s[62:63]\n"                                  \
    "v_cndmask_b32 v36, v36, v35, s[64:65]\n"                                  \
    "s_bitcmp1_b32 s99, 6\n"                                                   \
    "s_cselect_b64 s[90:91], s[62:63], 0\n"                                    \
    "s_bitcmp1_b32 s99, 7\n"                                                   \
    "s_cselect_b64 s[92:93], s[64:65], 0\n"                                    \
    "s_or_b64 s[90:91], s[90:91], s[92:93]\n"             /* lanes whose copy is dropped */ \
    "v_add_lshl_u32 v34, %[oi], %[oo], 3\n"                                    \
    "v_add_u32 v32, 1, %[oo]\n"                                                \
    "v_cndmask_b32 %[oo], %[oo], v32, s[90:91]\n"                              \
    "s_andn2_b64 exec, s[60:61], s[90:91]\n"                                   \
    "v_mov_b32 v37, s71\n"                                                     \
    "global_store_dwordx2 v34, v[36:37], s[76:77]\n"                           \
    "s_mov_b64 exec, -1\n"                                                     \
    TP_GO

struct TilePushState {
    uint32_t a0l, a0h, a1l, a1h;      /* active set: slots 0..63 in lanes of (a0l, a0h), 64..127 in (a1l, a1h) */
    uint32_t out_index, out_offset;   /* per lane: current chunk and next free word + 1 */
    uint32_t overflow;                /* per lane: ran out of chunks */
    uint64_t live;                    /* lanes still pushing */
    int head_index;                   /* out: pool index of the head clause the walk ended on */
    int words;
    int kept_minmax;                  /* out: min / max words at least one lane kept undecided (bounds the choices of every pushed tape) */
};

/* Walks backward from pool index `cur` (the word before the end clause).  `ci` = number of choices
 * the forward walk met; choices at LDS byte offset choice_off.  pool_limit = last pool index at
 * which a chunk may start. */
DEV void tile_push_asm(const uint64_t* __restrict__ pool, int cur, unsigned char* smem, int lane, TilePushState& st,
                       uint32_t run_end, int ci, uint32_t choice_off, int choice_cap, uint32_t pool_limit)
{
    uint32_t blo = 0, bhi = 0, bw = 0;
    uint32_t a0l = st.a0l, a0h = st.a0h, a1l = st.a1l, a1h = st.a1h, oi = st.out_index, oo = st.out_offset, ovf = st.overflow;
    uint32_t livelo = rdfirst((uint32_t)st.live), livehi = rdfirst((uint32_t)(st.live >> 32));
    const uint32_t plo = rdfirst((uint32_t)(uintptr_t)pool), phi = rdfirst((uint32_t)((uintptr_t)pool >> 32));
    const uint32_t cbase = (uint32_t)(uintptr_t)smem + choice_off;
    uint32_t caddr = rdfirst(cbase + (uint32_t)ci * 16u), uci = rdfirst((uint32_t)ci);
    const uint32_t cend = rdfirst(cbase + (uint32_t)choice_cap * 16u);
    uint32_t bbase = rdfirst((uint32_t)(cur - 63)), sj = 0, words = 0, kept = 0;
    const uint32_t lane1 = (uint32_t)lane;
    const uint32_t plim = rdfirst(pool_limit);
    asm volatile(
        "s_mov_b32 s89, %[bbase]\n"
        "s_mov_b32 s96, 0xff00\n"
        "s_mov_b32 s72, %[livelo]\n"
        "s_mov_b32 s73, %[livehi]\n"
        "s_mov_b32 s74, %[caddr]\n"
        "s_mov_b32 s75, %[cend]\n"
        "s_mov_b32 s76, %[plo]\n"
        "s_mov_b32 s77, %[phi]\n"
        "s_mov_b32 s78, %[ci]\n"
        "s_mov_b32 s79, 0\n"
        "s_mov_b32 s84, 0\n"
        "s_mov_b32 s98, %[plim]\n"
        "v_mov_b32 v38, 1\n"
        "v_mov_b32 v44, 1\n"                      /* word 63 of a new chunk: JUMP back to the previous one (-127) */
        "v_mov_b32 v45, 0xffffff81\n"
        "v_mov_b32 v46, 1\n"                      /* word 0 of the previous chunk: JUMP forward (+127) */
        "v_mov_b32 v47, 127\n"
        "s_getpc_b64 s[82:83]\n"
        "L_pc_%=:\n"
        "s_add_u32 s82, s82, L_p0_%=-L_pc_%=\n"
        "s_addc_u32 s83, s83, 0\n"
        /* ---- fetch words s89 .. s89 + 63 (lane 0: sentinel), classify ---- */
        "L_load_%=:\n"
        "v_add_u32 v32, s89, %[lane]\n"
        "v_max_i32 v32, 0, v32\n"
        "v_lshlrev_b32 v32, 3, v32\n"
        "global_load_dword %[blo], v32, s[76:77]\n"
        "global_load_dword %[bhi], v32, s[76:77] offset:4\n"
        "s_mov_b32 s88, 64\n"
        "s_waitcnt vmcnt(0)\n"
        "v_and_b32 v33, 0xff, %[blo]\n"                 /* op */
        "v_bfe_u32 v34, %[blo], 8, 8\n"                 /* out */
        "v_bfe_u32 v35, %[blo], 16, 8\n"                /* lhs */
        "v_lshrrev_b32 v36, 24, %[blo]\n"               /* rhs */
        /* variant = 4 * (out >= 64) + 2 * (lhs >= 64) + (rhs >= 64) */
        "v_lshrrev_b32 v37, 6, v34\n"
        "v_lshrrev_b32 v32, 6, v35\n"
        "v_lshl_or_b32 v37, v37, 1, v32\n"
        "v_lshrrev_b32 v32, 6, v36\n"
        "v_lshl_or_b32 v37, v37, 1, v32\n"
        /* handler index (x 256 bytes; a handler may be up to 1 KiB): plain 4 * variant, min / max
         * (17..20) 32 + 4 * variant, jump 64, head 65, lane 0: 66 */
        "v_add_u32 v32, -17, v33\n"
        "v_cmp_gt_u32 vcc, 4, v32\n"
        "v_lshlrev_b32 v37, 2, v37\n"
        "v_add_u32 v32, 32, v37\n"
        "v_cndmask_b32 v37, v37, v32, vcc\n"
        "v_cmp_eq_u32 vcc, 1, v33\n"
        "v_mov_b32 v32, 64\n"
        "s_nop 0\n"
        "v_cndmask_b32 v37, v37, v32, vcc\n"
        "v_cmp_eq_u32 vcc, 0, v33\n"
        "v_mov_b32 v32, 65\n"
        "s_nop 0\n"
        "v_cndmask_b32 v37, v37, v32, vcc\n"
        "v_cmp_eq_u32 vcc, 0, %[lane]\n"
        "v_mov_b32 v32, 66\n"
        "s_nop 0\n"
        "v_cndmask_b32 v37, v37, v32, vcc\n"
        /* byte 0: out & 63, bit 6 lhs == out, bit 7 rhs != 0 && rhs == out */
        "v_and_b32 v32, 63, v34\n"
        "v_cmp_eq_u32 vcc, v35, v34\n"
        "v_or_b32 v33, 64, v32\n"
        "s_nop 0\n"
        "v_cndmask_b32 v32, v32, v33, vcc\n"
        "v_cmp_eq_u32 vcc, v36, v34\n"
        "v_cmp_ne_u32 s[92:93], 0, v36\n"
        "s_and_b64 vcc, vcc, s[92:93]\n"
        "v_or_b32 v33, 0x80, v32\n"
        "v_cndmask_b32 v32, v32, v33, vcc\n"
        "v_lshl_or_b32 v32, v37, 8, v32\n"              /* byte 1: handler index */
        /* byte 2: lhs & 63, bit 6: no rhs; byte 3: rhs & 63 */
        "v_and_b32 v35, 63, v35\n"
        "v_cmp_eq_u32 vcc, 0, v36\n"
        "v_or_b32 v33, 64, v35\n"
        "s_nop 0\n"
        "v_cndmask_b32 v35, v35, v33, vcc\n"
        "v_lshl_or_b32 v32, v35, 16, v32\n"
        "v_and_b32 v36, 63, v36\n"
        "v_lshl_or_b32 %[bw], v36, 24, v32\n"
        "s_nop 0\n"
        TP_DISPATCH
        TP_PLAIN(0, "%[a0l]", "%[a0h]", "%[a0l]", "%[a0h]", "%[a0l]", "%[a0h]")
        TP_PLAIN(4, "%[a0l]", "%[a0h]", "%[a0l]", "%[a0h]", "%[a1l]", "%[a1h]")
        TP_PLAIN(8, "%[a0l]", "%[a0h]", "%[a1l]", "%[a1h]", "%[a0l]", "%[a0h]")
        TP_PLAIN(12, "%[a0l]", "%[a0h]", "%[a1l]", "%[a1h]", "%[a1l]", "%[a1h]")
        TP_PLAIN(16, "%[a1l]", "%[a1h]", "%[a0l]", "%[a0h]", "%[a0l]", "%[a0h]")
        TP_PLAIN(20, "%[a1l]", "%[a1h]", "%[a0l]", "%[a0h]", "%[a1l]", "%[a1h]")
        TP_PLAIN(24, "%[a1l]", "%[a1h]", "%[a1l]", "%[a1h]", "%[a0l]", "%[a0h]")
        TP_PLAIN(28, "%[a1l]", "%[a1h]", "%[a1l]", "%[a1h]", "%[a1l]", "%[a1h]")
        TP_MINMAX(32, "%[a0l]", "%[a0h]", "%[a0l]", "%[a0h]", "%[a0l]", "%[a0h]")
        TP_MINMAX(36, "%[a0l]", "%[a0h]", "%[a0l]", "%[a0h]", "%[a1l]", "%[a1h]")
        TP_MINMAX(40, "%[a0l]", "%[a0h]", "%[a1l]", "%[a1h]", "%[a0l]", "%[a0h]")
        TP_MINMAX(44, "%[a0l]", "%[a0h]", "%[a1l]", "%[a1h]", "%[a1l]", "%[a1h]")
        TP_MINMAX(48, "%[a1l]", "%[a1h]", "%[a0l]", "%[a0h]", "%[a0l]", "%[a0h]")
        TP_MINMAX(52, "%[a1l]", "%[a1h]", "%[a0l]", "%[a0h]", "%[a1l]", "%[a1h]")
        TP_MINMAX(56, "%[a1l]", "%[a1h]", "%[a1l]", "%[a1h]", "%[a0l]", "%[a0h]")
        TP_MINMAX(60, "%[a1l]", "%[a1h]", "%[a1l]", "%[a1h]", "%[a1l]", "%[a1h]")
        TP_H4(64)                                        /* JUMP: cur += imm - 1, then fetch around the new cur */
        "v_readlane_b32 s71, %[bhi], s88\n"
        "s_sub_u32 s92, 64, s88\n s_add_u32 s79, s79, s92\n"
        "s_add_u32 s89, s89, s88\n"
        "s_add_u32 s89, s89, s71\n"
        "s_sub_u32 s89, s89, 64\n"                       /* block base = new cur - 63 */
        "s_branch L_load_%=\n"
        TP_H(65)                                         /* head clause: done */
        "s_sub_u32 s92, 64, s88\n s_add_u32 s79, s79, s92\n"
        "s_branch L_exit_%=\n"
        TP_H(66)                                         /* lane 0: the previous 63 words */
        "s_add_u32 s79, s79, 63\n"
        "s_sub_u32 s89, s89, 63\n"
        "s_branch L_load_%=\n"
        "L_exit_%=:\n"
        "s_mov_b32 %[bbase], s89\n"
        "s_mov_b32 %[sj], s88\n"
        "s_mov_b32 %[livelo], s72\n"
        "s_mov_b32 %[livehi], s73\n"
        "s_mov_b32 %[words], s79\n"
        "s_mov_b32 %[kept], s84\n"
        : [blo] "+&v"(blo), [bhi] "+&v"(bhi), [bw] "+&v"(bw), [a0l] "+&v"(a0l), [a0h] "+&v"(a0h), [a1l] "+&v"(a1l), [a1h] "+&v"(a1h),
          [oi] "+&v"(oi), [oo] "+&v"(oo), [ovf] "+&v"(ovf), [bbase] "+&s"(bbase), [sj] "=&s"(sj), [livelo] "+&s"(livelo),
          [livehi] "+&s"(livehi), [words] "=&s"(words), [kept] "=&s"(kept)
        : [lane] "v"(lane1), [rend] "v"(run_end), [caddr] "s"(caddr), [cend] "s"(cend), [plo] "s"(plo), [phi] "s"(phi),
          [ci] "s"(uci), [plim] "s"(plim)
        : "memory", "vcc", "scc",
          "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75",
          "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s86", "s88", "s89", "s90", "s91", "s92", "s93", "s94",
          "s84", "s95", "s96", "s97", "s98", "s99",
          "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
    st.a0l = a0l; st.a0h = a0h; st.a1l = a1l; st.a1h = a1h;
    st.out_index = oi;
    st.out_offset = oo;
    st.overflow = ovf;
    st.live = ((uint64_t)livehi << 32) | livelo;
    st.head_index = (int)(bbase + sj);
    st.words = (int)words;
    st.kept_minmax = (int)kept;
}

}  // namespace mprk
