/*
 * kernels_normals_asm.hip — the normals pass (reference src/context.cu:978-1132, eval_pixels_d)
 * with the tape interpreter in gfx950 assembly.
 *
 * Layout as in k_eval_normals_q (kernels_float.hip): a wavefront is a 4x4 pixel footprint, lane =
 * pixel * 4 + component of the forward-mode derivative (dx, dy, dz, value), slot file float per
 * lane in LDS; the value of an operand reaches the three partials of its quad through DPP
 * quad_perm:[3,3,3,3], folded into the consuming instruction where the ISA allows it.  The
 * interpreter itself is the float pass's (kernels_voxel_asm.hip — read that header first):
 * 63-clause blocks rewritten per block to {out, handler index, lhs, rhs}, handler address on the
 * scalar unit, three handler tables with operand forwarding.  Every operation evaluates exactly
 * the expression of the compiled kernel (inc/gpu_deriv.hpp order of operations, no contraction),
 * with division, square root, exp and log in line through the shared routines of
 * asm_float_bodies.hpp; the inverse trigonometric functions and sin / cos leave the block.
 * One wave per workgroup (LDS base 0 for the v_perm_b32 address trick).
 *
 * Decisions.  When the last tile stage pushed no tapes (TileStageArgs::no_push) a smallest tile carries its
 * GROUP's tape, and its own tape — the one the reference walks here — is that tape with the tile's recorded
 * min / max decisions applied (clauses the shortening dropped are walked for nothing).  The kernel then keeps
 * the decisions of every pixel's tile as bit sets in LDS ({chose lhs, chose rhs} x 192 bits behind the slot
 * file) and the min / max handlers go through L_dec, which overrides the comparison's outcome for the lanes
 * whose tile decided that clause: the partials then come from the operand the tile's own tape would copy.
 */
#include "asm_float_bodies.hpp"
#include "kernel_common.hpp"

namespace mprk {

DEV float quad_bcast_a(float x, int which)
{
    const int v = (int)mpr_f2u(x);
    int r;
    switch (which) {
        case 0: r = __builtin_amdgcn_update_dpp(0, v, 0x00, 0xF, 0xF, true); break;
        case 1: r = __builtin_amdgcn_update_dpp(0, v, 0x55, 0xF, 0xF, true); break;
        case 2: r = __builtin_amdgcn_update_dpp(0, v, 0xAA, 0xF, 0xF, true); break;
        default: r = __builtin_amdgcn_update_dpp(0, v, 0xFF, 0xF, 0xF, true); break;
    }
    return mpr_u2f((uint32_t)r);
}
__device__ __noinline__ float deriv_trig_q(uint32_t op, float a, float av, bool isv)
{
    switch (op) {
        case MPR_OP_SIN_LHS: { const float c = mpr_cosf(av); return isv ? mpr_sinf(av) : c * a; }
        case MPR_OP_COS_LHS: { const float s = -mpr_sinf(av); return isv ? mpr_cosf(av) : s * a; }
        case MPR_OP_ASIN_LHS: { const float d = __builtin_sqrtf(1 - av * av); return isv ? mpr_asinf(av) : a / d; }
        case MPR_OP_ACOS_LHS: { const float d = -__builtin_sqrtf(1 - av * av); return isv ? mpr_acosf(av) : a / d; }
        case MPR_OP_ATAN_LHS: { const float d = av * av + 1; return isv ? mpr_atanf(av) : a / d; }
        default: return mpr_u2f(0x7FC00000u);
    }
}

/* asin / acos / atan as routines the assembly loop calls (NQ_CALLC): leaf functions of the AMDGPU
 * calling convention — a in v0, "this lane holds the value component" in v1, result in v0, return
 * address s[30:31] — with the very expressions of deriv_trig_q.  They keep v40..v47 and every SGPR from
 * s34 up, where the interpreter's state lives. */
__device__ __attribute__((noinline, used)) float nq_asin(float a, int isv) __asm__("mpr_nq_asin");
__device__ __attribute__((noinline, used)) float nq_acos(float a, int isv) __asm__("mpr_nq_acos");
__device__ __attribute__((noinline, used)) float nq_atan(float a, int isv) __asm__("mpr_nq_atan");
__device__ float nq_asin(float a, int isv)
{
    const float av = quad_bcast_a(a, 3);
    const float d = __builtin_sqrtf(1 - av * av);
    return isv ? mpr_asinf(av) : a / d;
}
__device__ float nq_acos(float a, int isv)
{
    const float av = quad_bcast_a(a, 3);
    const float d = -__builtin_sqrtf(1 - av * av);
    return isv ? mpr_acosf(av) : a / d;
}
__device__ float nq_atan(float a, int isv)
{
    const float av = quad_bcast_a(a, 3);
    const float d = av * av + 1;
    return isv ? mpr_atanf(av) : a / d;
}

/* Fixed registers (clobbers):
 *   s[80:81] handler address  s[82:83] table base  s[84:85] block address  s86 clause word  s87 immediate
 *   s88 clause counter  s89 block base  s90 0x260  s96 0xff00  s[98:99] lanes holding the value (comp 3)
 *   s[70:71] return address of the shared routines, s[72:79] their entry points (div, sqrt, exp, log)
 *   s64 decisions present  s65 min / max clauses met so far  s[66:67] L_dec  s[46:47] its return address
 *   v32 aA  v33 aB  v34 aO  v35 A  v36 B  v37 result (and previous result)  v38..v47 temporaries      */
#define NQ_Q3 " quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n"
/* next clause: word, handler address ... */
#define NQ_PREP                                        \
    "s_add_u32 s88, s88, 1\n"                          \
    "v_readlane_b32 s86, %[blo], s88\n"                \
    "s_and_b32 s80, s86, s96\n"                        \
    "s_add_u32 s80, s80, s82\n"                        \
    "s_addc_u32 s81, s83, 0\n"
/* ... and go.  Handlers run NQ_PREP right after forming their addresses (the last use of the clause word) and issuing
 * their LDS reads, under the reads' latency: a wavefront's walk is a dependent chain, and this link need not be on it */
#define NQ_DISPATCH NQ_PREP "s_setpc_b64 s[80:81]\n"
#define NQ_IMM "v_readlane_b32 s87, %[bhi], s88\n"
#define NQ_AL "v_perm_b32 v32, s86, %[lb], %[selL]\n ds_read_b32 v35, v32\n"
#define NQ_AR "v_perm_b32 v33, s86, %[lb], %[selR]\n ds_read_b32 v36, v33\n"
#define NQ_FL "v_mov_b32 v35, v37\n"                   /* lhs = previous result */
#define NQ_FR "v_mov_b32 v36, v37\n"
#define NQ_AO "v_perm_b32 v34, s86, %[lb], %[selO]\n"
#define NQ_W "s_waitcnt lgkmcnt(0)\n"
#define NQ_ST "ds_write_b32 v34, v37\n"
#define NQ_END NQ_ST "s_setpc_b64 s[80:81]\n"
#define NQ_H(v, n) ".p2align 8\nL_n" #v "_" #n "_%=:\n"
#define NQ_EXIT NQ_IMM "s_branch L_exit_%=\n"
#define NQ_H30 NQ_EXIT               /* a word that is not an opcode: evaluated (to NaN) outside the block */
#define NQ_VS_ENTER
#define NQ_VS_LEAVE
#define NQ_CALL(pair) "s_swappc_b64 s[70:71], " pair "\n"
/* v37 = sym(v35) by a compiled routine; v34 (address of the out slot) survives in v44 */
#define NQ_CALLC(sym)                                                                            \
    "v_mov_b32 v44, v34\n v_mov_b32 v0, v35\n v_cndmask_b32_e64 v1, 0, 1, s[98:99]\n"            \
    "s_getpc_b64 s[40:41]\n"                                                                     \
    "s_add_u32 s40, s40, " sym "@rel32@lo+4\n"                                                   \
    "s_addc_u32 s41, s41, " sym "@rel32@hi+12\n"                                                 \
    "s_swappc_b64 s[30:31], s[40:41]\n"                                                          \
    "v_mov_b32 v37, v0\n v_mov_b32 v34, v44\n" NQ_END
#define NQ_DIV NQ_CALL("s[72:73]")
#define NQ_SQRT NQ_CALL("s[74:75]")
#define NQ_EXP NQ_CALL("s[76:77]")
#define NQ_LOG NQ_CALL("s[78:79]")

/* Handlers read lhs from v35 and rhs from v36 (LDL / LDR bring them there: LDS load, or a copy of
 * the previous result) and leave the result in v37.  a = lhs, b = rhs as Derivs, av / bv their
 * values broadcast over the quad; isv = lanes of component 3. */
#define NQ_TABLE(v, LDL, LDR, WL, WR, WLR)                                                                   \
    NQ_H(v, 0) "s_branch L_exit_%=\n"                                                     /* end of tape */  \
    NQ_H(v, 1) NQ_IMM "s_add_u32 s89, s89, s88\n s_add_u32 s89, s89, s87\n s_add_u32 s89, s89, 1\n s_branch L_load_%=\n" \
    NQ_H(v, 2) LDL NQ_AO NQ_PREP WL                                  /* SQUARE: isv ? a*a : a*av + a*av */           \
    "v_mul_f32_dpp v38, v35, v35" NQ_Q3 "v_mul_f32 v39, v35, v35\n v_add_f32 v38, v38, v38\n"               \
    "v_cndmask_b32 v37, v38, v39, s[98:99]\n" NQ_END                                                         \
    NQ_H(v, 3) LDL NQ_AO NQ_PREP WL                                  /* SQRT: s = sqrt(av); isv ? s : a / (2 s) */   \
    "v_mov_b32 v43, v35\n v_mov_b32_dpp v35, v35" NQ_Q3 NQ_SQRT                                              \
    "v_mov_b32 v44, v37\n v_mul_f32 v36, 2.0, v37\n v_mov_b32 v35, v43\n" NQ_DIV                             \
    "v_cndmask_b32 v37, v37, v44, s[98:99]\n" NQ_END                                                         \
    NQ_H(v, 4) LDL NQ_AO NQ_PREP WL "v_xor_b32 v37, 0x80000000, v35\n" NQ_END                                        \
    NQ_H(v, 5) LDL NQ_AO NQ_PREP WL                                  /* SIN: isv ? sin(av) : cos(av) * a */          \
    "v_mov_b32 v48, v35\n v_mov_b32_dpp v35, v35" NQ_Q3 NQ_CALL("s[68:69]")                                  \
    "v_mul_f32 v38, v36, v48\n v_cndmask_b32 v37, v38, v37, s[98:99]\n" NQ_END                              \
    NQ_H(v, 6) LDL NQ_AO NQ_PREP WL                                  /* COS: isv ? cos(av) : -sin(av) * a */         \
    "v_mov_b32 v48, v35\n v_mov_b32_dpp v35, v35" NQ_Q3 NQ_CALL("s[68:69]")                                  \
    "v_mul_f32_e64 v38, -v37, v48\n v_cndmask_b32 v37, v38, v36, s[98:99]\n" NQ_END                         \
    NQ_H(v, 7) LDL NQ_AO NQ_PREP WL "s_branch L_casin_%=\n"                                                         \
    NQ_H(v, 8) LDL NQ_AO NQ_PREP WL "s_branch L_cacos_%=\n"                                                         \
    NQ_H(v, 9) LDL NQ_AO NQ_PREP WL "s_branch L_catan_%=\n"                                                         \
    NQ_H(v, 10) LDL NQ_AO NQ_PREP WL                                 /* EXP: e = exp(av); isv ? e : e * a */         \
    "v_mov_b32 v43, v35\n v_mov_b32_dpp v35, v35" NQ_Q3 NQ_EXP                                               \
    "v_mul_f32 v38, v37, v43\n v_cndmask_b32 v37, v38, v37, s[98:99]\n" NQ_END                               \
    NQ_H(v, 11) LDL NQ_AO NQ_PREP WL                                 /* ABS: av < 0 ? -a : a */                      \
    "v_mov_b32 v39, 0\n v_xor_b32 v38, 0x80000000, v35\n v_mov_b32_dpp v46, v35" NQ_Q3 "v_cmp_lt_f32 vcc, v46, v39\n"               \
    "s_nop 1\n v_cndmask_b32 v37, v35, v38, vcc\n" NQ_END                                                    \
    NQ_H(v, 12) LDL NQ_AO NQ_PREP WL                                 /* LOG: isv ? log(av) : a / av */               \
    "v_mov_b32 v43, v35\n v_mov_b32_dpp v35, v35" NQ_Q3 "s_nop 0\n v_mov_b32 v45, v35\n" NQ_LOG              \
    "v_mov_b32 v44, v37\n v_mov_b32 v35, v43\n v_mov_b32 v36, v45\n" NQ_DIV                                  \
    "v_cndmask_b32 v37, v37, v44, s[98:99]\n" NQ_END                                                         \
    NQ_H(v, 13) NQ_IMM LDL NQ_AO NQ_PREP WL                          /* a + imm: only the value */                   \
    "s_nop 0\n v_add_f32 v38, s87, v35\n v_cndmask_b32 v37, v35, v38, s[98:99]\n" NQ_END                     \
    NQ_H(v, 14) LDL LDR NQ_AO NQ_PREP WLR "v_add_f32 v37, v35, v36\n" NQ_END                                         \
    NQ_H(v, 15) NQ_IMM LDL NQ_AO NQ_PREP WL "s_nop 0\n v_mul_f32 v37, s87, v35\n" NQ_END                             \
    NQ_H(v, 16) LDL LDR NQ_AO NQ_PREP WLR                            /* MUL: isv ? a*b : a*bv + b*av */              \
    "v_mul_f32_dpp v38, v36, v35" NQ_Q3 "v_mul_f32_dpp v39, v35, v36" NQ_Q3                                  \
    "v_mul_f32 v40, v35, v36\n v_add_f32 v38, v38, v39\n v_cndmask_b32 v37, v38, v40, s[98:99]\n" NQ_END     \
    NQ_H(v, 17) NQ_IMM LDL NQ_AO NQ_PREP WL                          /* MIN_IMM: b = (0, 0, 0, imm); av < imm ? a : b */ \
    "v_mov_b32 v39, s87\n v_cndmask_b32 v36, 0, v39, s[98:99]\n v_mov_b32_dpp v46, v35" NQ_Q3 "v_cmp_lt_f32 vcc, v46, v39\n"        \
    "s_swappc_b64 s[46:47], s[66:67]\n"                                                                       \
    "s_nop 1\n v_cndmask_b32 v37, v36, v35, vcc\n" NQ_END                                                    \
    NQ_H(v, 18) LDL LDR NQ_AO NQ_PREP WLR                            /* MIN: av < bv ? a : b */                      \
    "v_mov_b32_dpp v39, v36" NQ_Q3 "s_nop 1\n v_mov_b32_dpp v46, v35" NQ_Q3 "v_cmp_lt_f32 vcc, v46, v39\n"                          \
    "s_swappc_b64 s[46:47], s[66:67]\n"                                                                       \
    "s_nop 1\n v_cndmask_b32 v37, v36, v35, vcc\n" NQ_END                                                    \
    NQ_H(v, 19) NQ_IMM LDL NQ_AO NQ_PREP WL                          /* MAX_IMM: av >= imm ? a : b */                \
    "v_mov_b32 v39, s87\n v_cndmask_b32 v36, 0, v39, s[98:99]\n v_mov_b32_dpp v46, v35" NQ_Q3 "v_cmp_ge_f32 vcc, v46, v39\n"        \
    "s_swappc_b64 s[46:47], s[66:67]\n"                                                                       \
    "s_nop 1\n v_cndmask_b32 v37, v36, v35, vcc\n" NQ_END                                                    \
    NQ_H(v, 20) LDL LDR NQ_AO NQ_PREP WLR                                                                            \
    "v_mov_b32_dpp v39, v36" NQ_Q3 "s_nop 1\n v_mov_b32_dpp v46, v35" NQ_Q3 "v_cmp_ge_f32 vcc, v46, v39\n"                          \
    "s_swappc_b64 s[46:47], s[66:67]\n"                                                                       \
    "s_nop 1\n v_cndmask_b32 v37, v36, v35, vcc\n" NQ_END                                                    \
    NQ_H(v, 21) NQ_IMM LDL NQ_AO NQ_PREP WL                          /* a - imm: only the value */                   \
    "s_nop 0\n v_subrev_f32 v38, s87, v35\n v_cndmask_b32 v37, v35, v38, s[98:99]\n" NQ_END                  \
    NQ_H(v, 22) NQ_IMM LDR NQ_AO NQ_PREP WR                          /* imm - b: isv ? imm - b : -b */               \
    "s_nop 0\n v_sub_f32 v38, s87, v36\n v_xor_b32 v39, 0x80000000, v36\n v_cndmask_b32 v37, v39, v38, s[98:99]\n" NQ_END \
    NQ_H(v, 23) LDL LDR NQ_AO NQ_PREP WLR "v_sub_f32 v37, v35, v36\n" NQ_END                                         \
    NQ_H(v, 24) NQ_IMM LDL NQ_AO NQ_PREP WL "s_nop 0\n v_mov_b32 v36, s87\n" NQ_DIV NQ_END      /* a / imm, all components */ \
    NQ_H(v, 25) NQ_IMM LDR NQ_AO NQ_PREP WR                          /* imm / b: isv ? imm / b : (-imm * b) / (bv * bv) */ \
    "v_mov_b32_dpp v41, v36" NQ_Q3 "v_mul_f32_e64 v39, -s87, v36\n v_mov_b32 v40, s87\n v_mul_f32 v38, v41, v41\n" \
    "v_cndmask_b32 v35, v39, v40, s[98:99]\n v_cndmask_b32 v36, v38, v36, s[98:99]\n" NQ_DIV NQ_END          \
    NQ_H(v, 26) LDL LDR NQ_AO NQ_PREP WLR                            /* a / b: isv ? a / b : (bv*a - av*b) / (bv*bv) */ \
    "v_mul_f32_dpp v38, v36, v35" NQ_Q3 "v_mul_f32_dpp v39, v35, v36" NQ_Q3 "v_mov_b32_dpp v41, v36" NQ_Q3      \
    "v_sub_f32 v38, v38, v39\n s_nop 0\n v_mul_f32 v40, v41, v41\n"                                           \
    "v_cndmask_b32 v35, v38, v35, s[98:99]\n v_cndmask_b32 v36, v40, v36, s[98:99]\n"                         \
    NQ_DIV NQ_END                                                                                            \
    NQ_H(v, 27) NQ_IMM NQ_AO NQ_PREP "s_nop 0\n v_mov_b32 v39, s87\n v_cndmask_b32 v37, 0, v39, s[98:99]\n" NQ_END   \
    NQ_H(v, 28) LDL NQ_AO NQ_PREP WL "v_mov_b32 v37, v35\n" NQ_END                                                   \
    NQ_H(v, 29) LDR NQ_AO NQ_PREP WR "v_mov_b32 v37, v36\n" NQ_END                                                   \
    NQ_H(v, 30) NQ_H30                                                                                       \
    NQ_H(v, 31) "s_add_u32 s89, s89, 63\n s_branch L_load_%=\n"

/* The walk itself, as text: expanded in interp_normals_asm (slots in LDS) and in interp_normals_asm_vgpr (slots in
 * VGPRs), each time with that variant's NQ_AL / NQ_AR / NQ_AO / NQ_ST / NQ_H30 / NQ_VS_ENTER / NQ_VS_LEAVE. */
#define NQ_ASM_TEXT \
    NQ_VS_ENTER \
    "s_mov_b32 s89, %[base]\n" \
    "s_mov_b32 s88, %[sj]\n" \
    "s_mov_b32 s64, %[dec]\n" \
    "s_mov_b32 s65, %[mmc]\n" \
    "s_mov_b32 s90, 0x260\n" \
    "s_mov_b32 s96, 0xff00\n" \
    "s_mov_b32 s98, 0x88888888\n" \
    "s_mov_b32 s99, 0x88888888\n" \
    "v_mov_b32 v37, %[prev]\n" \
    "s_getpc_b64 s[82:83]\n" \
    "L_pc_%=:\n" \
    "s_add_u32 s72, s82, L_div_%=-L_pc_%=\n s_addc_u32 s73, s83, 0\n" \
    "s_add_u32 s74, s82, L_sqrt_%=-L_pc_%=\n s_addc_u32 s75, s83, 0\n" \
    "s_add_u32 s76, s82, L_exp_%=-L_pc_%=\n s_addc_u32 s77, s83, 0\n" \
    "s_add_u32 s78, s82, L_log_%=-L_pc_%=\n s_addc_u32 s79, s83, 0\n" \
    "s_add_u32 s68, s82, L_sincos_%=-L_pc_%=\n s_addc_u32 s69, s83, 0\n" \
    "s_add_u32 s66, s82, L_dec_%=-L_pc_%=\n s_addc_u32 s67, s83, 0\n" \
    "s_add_u32 s82, s82, L_n0_0_%=-L_pc_%=\n" \
    "s_addc_u32 s83, s83, 0\n" \
    "s_cmp_eq_u32 %[mode], 0\n" \
    "s_cbranch_scc1 L_load_%=\n" \
    NQ_DISPATCH \
    "L_load_%=:\n" \
    "s_mov_b32 s84, s89\n" \
    "s_mov_b32 s85, 0\n" \
    "s_lshl_b64 s[84:85], s[84:85], 3\n" \
    "s_add_u32 s84, s84, %[tlo]\n" \
    "s_addc_u32 s85, s85, %[thi]\n" \
    "global_load_dword %[blo], %[lane8], s[84:85]\n" \
    "global_load_dword %[bhi], %[lane8], s[84:85] offset:4\n" \
    "s_mov_b32 s88, -1\n" \
    "v_mov_b32 v41, 0\n" \
    "v_mov_b32 v43, 32\n" \
    "v_mov_b32 v44, 64\n" \
    "s_waitcnt vmcnt(0)\n" \
    "v_bfe_u32 v40, %[blo], 8, 8\n" \
    "v_and_b32 v38, 0xff, %[blo]\n" \
    "v_min_u32 v38, 30, v38\n" \
    "v_mov_b32_dpp v41, v40 wave_shr:1 row_mask:0xf bank_mask:0xf\n" \
    "v_bfe_u32 v42, %[blo], 16, 8\n" \
    "v_lshrrev_b32 v39, 24, %[blo]\n" \
    "v_cmp_eq_u32 s[92:93], v39, v41\n" \
    "v_cmp_eq_u32 vcc, v42, v41\n" \
    "v_cmp_ne_u32 s[94:95], 0, v41\n" \
    "v_cndmask_b32 v42, 0, v44, s[92:93]\n" \
    "v_cndmask_b32 v42, v42, v43, vcc\n" \
    "v_cndmask_b32 v42, 0, v42, s[94:95]\n" \
    "v_cmp_eq_u32 vcc, 0x1f8, %[lane8]\n" \
    "v_mov_b32 v39, 31\n" \
    "v_add_u32 v38, v38, v42\n" \
    "v_cndmask_b32 v38, v38, v39, vcc\n" \
    "v_lshl_or_b32 v38, v38, 8, v40\n" \
    "v_and_b32 %[blo], 0xffff0000, %[blo]\n" \
    "v_or_b32 %[blo], %[blo], v38\n" \
    "s_nop 0\n" \
    NQ_DISPATCH \
    NQ_TABLE(0, NQ_AL, NQ_AR, NQ_W, NQ_W, NQ_W) \
    NQ_TABLE(1, NQ_FL, NQ_AR, "", NQ_W, NQ_W) \
    NQ_TABLE(2, NQ_AL, NQ_FR, NQ_W, "", NQ_W) \
    /* shared routines: argument v35 (and v36), result v37, return to s[70:71] */ \
    ".p2align 8\n" \
    "L_div_%=:\n" MPR_ASM_DIV_BODY "s_setpc_b64 s[70:71]\n" \
    "L_sqrt_%=:\n" MPR_ASM_SQRT_BODY "s_setpc_b64 s[70:71]\n" MPR_ASM_SQRT_TAIL \
    "L_exp_%=:\n" MPR_ASM_EXP_BODY "s_setpc_b64 s[70:71]\n" MPR_ASM_EXP_TAIL \
    "L_log_%=:\n" MPR_ASM_LOG_BODY "s_setpc_b64 s[70:71]\n" MPR_ASM_LOG_TAIL \
    "L_sincos_%=:\n" MPR_ASM_SINCOS_BODY "s_setpc_b64 s[70:71]\n" \
    /* a min / max clause: count it; when decisions are present, the lanes whose tile decided it take the chosen \
    * operand whatever the comparison said (vcc set: lhs).  Bits beyond the 192 kept per lane: undecided. */ \
    "L_dec_%=:\n" \
    "s_add_u32 s65, s65, 1\n" \
    "s_cmp_eq_u32 s64, 0\n" \
    "s_cbranch_scc1 L_decret_%=\n" \
    "s_sub_u32 s40, s65, 1\n" \
    "s_lshr_b32 s41, s40, 6\n" \
    "s_cmp_ge_u32 s41, 3\n" \
    "s_cbranch_scc1 L_decret_%=\n" \
    "s_and_b32 s40, s40, 63\n" \
    "s_lshl_b32 s41, s41, 7\n" \
    "v_add_u32 v42, s41, %[decb]\n" \
    "ds_read_b64 v[38:39], v42\n" \
    "ds_read_b64 v[40:41], v42 offset:384\n" \
    "s_waitcnt lgkmcnt(0)\n" \
    "v_lshrrev_b64 v[38:39], s40, v[38:39]\n" \
    "v_lshrrev_b64 v[40:41], s40, v[40:41]\n" \
    "v_and_b32 v38, 1, v38\n" \
    "v_and_b32 v40, 1, v40\n" \
    "v_cmp_ne_u32 s[42:43], 0, v38\n" \
    "v_cmp_ne_u32 s[44:45], 0, v40\n" \
    "s_or_b64 vcc, vcc, s[42:43]\n" \
    "s_andn2_b64 vcc, vcc, s[44:45]\n" \
    "L_decret_%=:\n" \
    "s_setpc_b64 s[46:47]\n" \
    "L_casin_%=:\n" NQ_CALLC("mpr_nq_asin") \
    "L_cacos_%=:\n" NQ_CALLC("mpr_nq_acos") \
    "L_catan_%=:\n" NQ_CALLC("mpr_nq_atan") \
    "L_exit_%=:\n" \
    "s_waitcnt lgkmcnt(0)\n" \
    NQ_VS_LEAVE \
    "s_mov_b32 %[dlo], s86\n" \
    "s_mov_b32 %[dhi], s87\n" \
    "s_mov_b32 %[base], s89\n" \
    "s_mov_b32 %[sj], s88\n" \
    "s_mov_b32 %[mmc], s65\n"

/* Walks the tape at tro[first] over the slot file at LDS offset 0; returns the result slot. */
DEV uint32_t interp_normals_asm(const uint64_t* __restrict__ tro, uint32_t first, unsigned char* smem, int lane, bool isv,
                                uint32_t decisions, uint32_t decb)
{
    unsigned char* const myslot = smem + lane * 4;
    uint32_t blo = 0, bhi = 0;
    uint32_t base = first, sj = 0, dlo = 0, dhi = 0;
    const uint32_t lb = (uint32_t)(uintptr_t)smem + (uint32_t)lane * 4u;
    const uint32_t lane8 = (uint32_t)lane * 8u;
    const uint32_t selO = to_vgpr(0x0c0c0400u), selL = to_vgpr(0x0c0c0600u), selR = to_vgpr(0x0c0c0700u);
    const uint32_t tlo = (uint32_t)(uintptr_t)tro, thi = (uint32_t)((uintptr_t)tro >> 32);
    float prev = 0.0f;
    uint32_t mode = 0;
    uint32_t mmc = 0;                  /* min / max clauses met so far */
    decisions = rdfirst(decisions);

    for (;;) {
        base = rdfirst(base);
        sj = rdfirst(sj);
        mode = rdfirst(mode);
        mmc = rdfirst(mmc);
        asm volatile(
            NQ_ASM_TEXT
            : [blo] "+&v"(blo), [bhi] "+&v"(bhi), [base] "+&s"(base), [sj] "+&s"(sj), [dlo] "=&s"(dlo), [dhi] "=&s"(dhi), [mmc] "+&s"(mmc)
            : [lb] "v"(lb), [selL] "v"(selL), [selR] "v"(selR), [selO] "v"(selO), [lane8] "v"(lane8),
              [tlo] "s"(tlo), [thi] "s"(thi), [mode] "s"(mode), [prev] "v"(prev), [dec] "s"(decisions), [decb] "v"(decb)
            : "memory", "vcc", "scc",
              "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84",
              "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s98", "s99",
              "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48",
              /* what the called routines may use on top (NQ_CALLC) */
              "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15",
              "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31",
              "s0", "s1", "s2", "s3", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17", "s18", "s19",
              "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31");
        const uint32_t op = (dlo >> 8) & 31;
        if (op == 0) break;
        /* sin, cos, asin, acos, atan (and anything that is not an opcode) */
        const float A = *reinterpret_cast<const float*>(myslot + ((dlo >> 8) & 0xFF00));
        prev = deriv_trig_q(op, A, quad_bcast_a(A, 3), isv);
        *reinterpret_cast<float*>(myslot + ((dlo & 0xFF) << 8)) = prev;
        mode = 1;
    }
    return dlo & 0xFF;
}

/* The same walk with the slot file in VECTOR REGISTERS (slot s = v[50 + s], addressed through the scalar GPR index:
 * see tile_interp_asm.hpp), for tapes with many slots: 256 bytes of LDS per slot and wavefront leave a tape with 93 slots
 * 6 wavefronts per CU; 93 registers on top of the walk's own are 3 per SIMD, 12 per CU.  The axes' values arrive and the
 * result leaves inside the one asm statement, and a word that is not an opcode gets its NaN inside the block: nothing
 * re-enters. */
#undef NQ_AL
#undef NQ_AR
#undef NQ_AO
#undef NQ_ST
#undef NQ_H30
#undef NQ_VS_ENTER
#undef NQ_VS_LEAVE
#define NQ_AL "s_bfe_u32 s60, s86, 0x80010\n s_set_gpr_idx_on s60, gpr_idx(SRC0)\n v_mov_b32 v35, v50\n s_set_gpr_idx_off\n"
#define NQ_AR "s_lshr_b32 s60, s86, 24\n s_set_gpr_idx_on s60, gpr_idx(SRC0)\n v_mov_b32 v36, v50\n s_set_gpr_idx_off\n"
#define NQ_AO "s_and_b32 s61, s86, 0xff\n"                        /* the out slot, kept for NQ_ST (NQ_PREP replaces s86) */
#define NQ_ST "s_set_gpr_idx_on s61, gpr_idx(DST)\n v_mov_b32 v50, v37\n s_set_gpr_idx_off\n"
#define NQ_H30 NQ_AO NQ_PREP "v_mov_b32 v37, 0x7fc00000\n" NQ_END
#define NQ_VS_ENTER                                                                                                   \
    "s_set_gpr_idx_on %[ax], gpr_idx(DST)\n v_mov_b32 v50, %[xin]\n s_set_gpr_idx_off\n"                              \
    "s_set_gpr_idx_on %[ay], gpr_idx(DST)\n v_mov_b32 v50, %[yin]\n s_set_gpr_idx_off\n"                              \
    "s_set_gpr_idx_on %[az], gpr_idx(DST)\n v_mov_b32 v50, %[zin]\n s_set_gpr_idx_off\n"
/* at the end clause: byte 0 of the rewritten word = the result's slot */
#define NQ_VS_LEAVE "s_and_b32 s60, s86, 0xff\n s_set_gpr_idx_on s60, gpr_idx(SRC0)\n v_mov_b32 %[res], v50\n s_set_gpr_idx_off\n"
#define NQ_V10(a) "v" #a "0", "v" #a "1", "v" #a "2", "v" #a "3", "v" #a "4", "v" #a "5", "v" #a "6", "v" #a "7", "v" #a "8", "v" #a "9"
constexpr int NQ_VS_MAX_SLOTS = 93;

/* ax / ay / az: the axes' slots (head clause); xin / yin / zin: this lane's component of their Derivs; returns the result's */
DEV float interp_normals_asm_vgpr(const uint64_t* __restrict__ tro, uint32_t first, int lane, uint32_t decisions, uint32_t decb,
                                  uint32_t ax, uint32_t ay, uint32_t az, float xin, float yin, float zin)
{
    uint32_t blo = 0, bhi = 0;
    uint32_t base = rdfirst(first), sj = 0, dlo = 0, dhi = 0, mmc = 0;
    const uint32_t lane8 = (uint32_t)lane * 8u;
    const uint32_t tlo = (uint32_t)(uintptr_t)tro, thi = (uint32_t)((uintptr_t)tro >> 32);
    const uint32_t mode = 0;
    const uint32_t zero = 0;
    float res = 0.0f;
    decisions = rdfirst(decisions);
    ax = rdfirst(ax);
    ay = rdfirst(ay);
    az = rdfirst(az);
#define NQ_VS_OPERANDS \
        : [blo] "+&v"(blo), [bhi] "+&v"(bhi), [base] "+&s"(base), [sj] "+&s"(sj), [dlo] "=&s"(dlo), [dhi] "=&s"(dhi), [mmc] "+&s"(mmc),\
          [res] "=&v"(res)\
        : [lane8] "v"(lane8), [tlo] "s"(tlo), [thi] "s"(thi), [mode] "s"(mode), [prev] "s"(zero), [dec] "s"(decisions), [decb] "v"(decb),\
          [ax] "s"(ax), [ay] "s"(ay), [az] "s"(az), [xin] "v"(xin), [yin] "v"(yin), [zin] "v"(zin)
#define NQ_VS_CLOBBERS \
        : "memory", "vcc", "scc",\
          "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s60", "s61", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84",\
          "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s98", "s99",\
          "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48",\
          "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15",\
          "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31",\
          "s0", "s1", "s2", "s3", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17", "s18", "s19",\
          "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31"
    asm volatile(NQ_ASM_TEXT NQ_VS_OPERANDS NQ_VS_CLOBBERS, NQ_V10(5), NQ_V10(6), NQ_V10(7), NQ_V10(8), NQ_V10(9), NQ_V10(10), NQ_V10(11),
                 NQ_V10(12), NQ_V10(13), "v140", "v141", "v142");
#undef NQ_VS_OPERANDS
#undef NQ_VS_CLOBBERS
    return res;
}

template <int VS>          /* 0, or the slots the register file is built for */
__global__ void __launch_bounds__(64, VS ? 3 : 0)
k_eval_normals_asm(NormalArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    unsigned char* const myslot = smem + lane * 4;
    const int S = a.size;
    const int fside = S / 4;                                  /* footprints per side */
    int fxi, fyi;
    if (a.col_list) {
        /* a rank of a multi-GPU frame visits only the columns it owns: 256 footprints per column */
        const int col = a.col_list[blockIdx.x >> 8], f = blockIdx.x & 255, cols = S / 64;
        fxi = (col % cols) * 16 + (f & 15);
        fyi = (col / cols) * 16 + (f >> 4);
    } else {
        fxi = blockIdx.x % fside;
        fyi = blockIdx.x / fside;
    }
    const int pix = lane >> 2, comp = lane & 3;
    const bool isv = comp == 3;
    const int px = fxi * 4 + (pix & 3), py = fyi * 4 + (pix >> 2);
    const int pxy = px + py * S;
    int pz = a.image[pxy];
    const bool filled = pz != 0;
    uint64_t todo = ballot(filled);
    if (todo == 0) return;
    if (pz < S - 1) pz += 1;                                   /* :1003-1005 */

    const float size_recip = 1.0f / (float)(unsigned)S;
    const float fx = ((px + 0.5f) * size_recip - 0.5f) * 2.0f;
    const float fy = ((py + 0.5f) * size_recip - 0.5f) * 2.0f;
    const float fz = ((pz + 0.5f) * size_recip - 0.5f) * 2.0f;
    const float fw = a.mat[3] * fx + a.mat[7] * fy + a.mat[11] * fz + a.mat[15];
    const float vx = (a.mat[0] * fx + a.mat[4] * fy + a.mat[8] * fz + a.mat[12]) / fw;
    const float vy = (a.mat[1] * fx + a.mat[5] * fy + a.mat[9] * fz + a.mat[13]) / fw;
    const float vz = (a.mat[2] * fx + a.mat[6] * fy + a.mat[10] * fz + a.mat[14]) / fw;

    /* deepest tile's tape (:1034-1066) */
    int my_tape = 0, my_micro = -1;
    if (filled) {
        const int t64 = S / 64;
        const int tile = px / 64 + (py / 64) * t64 + (pz / 64) * t64 * t64;
        const mpr_tile_node tn = a.tiles[tile];
        if (tn.next == -1) {
            my_tape = tn.tape;
        } else {
            const int subtile = tn.next * 64 + (px % 64) / 16 + ((py % 64) / 16) * 4 + ((pz % 64) / 16) * 16;
            const mpr_tile_node sn = a.subtiles[subtile];
            if (sn.next == -1) {
                my_tape = sn.tape;
            } else {
                const int micro = sn.next * 64 + (px % 16) / 4 + ((py % 16) / 4) * 4 + ((pz % 16) / 4) * 16;
                my_tape = a.microtiles[micro].tape;
                my_micro = micro;
            }
        }
    }
    /* the last tile stage pushed no tapes: this smallest tile's own tape is its group's (my_tape) with these decisions */
    /* with the groups' records at hand every smallest tile is walked on its group's tape (in a frame whose last stage pushed
     * tapes the node names the tile's own: same values either way, but tiles of a group then share one walk) */
    if (a.groups && my_micro >= 0) my_tape = a.groups[my_micro >> 6].tape;
    unsigned long long dl0 = 0, dl1 = 0, dl2 = 0, dr0 = 0, dr1 = 0, dr2 = 0;
    if (a.groups) {
        /* one (group, tile) at a time — a footprint meets a handful: lane i fetches the group's i-th pair of masks, and
         * the tile's bit of every mask is a ballot away (the float pass gets its decisions the same way) */
        uint64_t pending = ballot(my_micro >= 0);
        while (pending) {
            const int leader = __ffsll((long long)pending) - 1;
            const int micro = __builtin_amdgcn_readlane(my_micro, leader);
            pending &= ~ballot(my_micro == micro);
            const int g = micro >> 6, child = micro & 63;
            const GroupInfo gi = a.groups[g];
            if (!((gi.pushed >> child) & 1ull)) continue;
            const ulonglong2* const m = a.choice_masks + (size_t)g * a.choice_cap;
            const int n = gi.nchoices < 192 ? gi.nchoices : 192;
            ulonglong2 w0 = make_ulonglong2(0ull, 0ull), w1 = w0, w2 = w0;
            if (lane < n) w0 = m[lane];
            if (n > 64) {
                if (lane + 64 < n) w1 = m[lane + 64];
                if (lane + 128 < n) w2 = m[lane + 128];
            }
            const uint64_t l0 = ballot((w0.x >> child) & 1ull), r0 = ballot((w0.y >> child) & 1ull);
            uint64_t l1 = 0, r1 = 0, l2 = 0, r2 = 0;
            if (n > 64) {
                l1 = ballot((w1.x >> child) & 1ull);
                r1 = ballot((w1.y >> child) & 1ull);
                l2 = ballot((w2.x >> child) & 1ull);
                r2 = ballot((w2.y >> child) & 1ull);
            }
            if (my_micro == micro) {
                dl0 = l0; dr0 = r0; dl1 = l1; dr1 = r1; dl2 = l2; dr2 = r2;
            }
        }
    }
    unsigned char* const dec = smem + (VS ? 0 : (size_t)a.nslots * 256);          /* [2][3][16 pixels] 64-bit words behind the slot file */
    if (a.groups) {
        unsigned long long* const pl = reinterpret_cast<unsigned long long*>(dec) + pix;
        unsigned long long* const pr = reinterpret_cast<unsigned long long*>(dec + 384) + pix;
        pl[0] = dl0; pl[16] = dl1; pl[32] = dl2;
        pr[0] = dr0; pr[16] = dr1; pr[32] = dr2;
    }
    const uint32_t decb = (uint32_t)(uintptr_t)dec + (uint32_t)pix * 8u;

    const uint64_t* __restrict__ const tro = a.tape_ro;
    const uint64_t head0 = tro[0];
    const uint32_t sx = (head0 >> 8) & 0xFF, sy = (head0 >> 16) & 0xFF, sz = (head0 >> 24) & 0xFF;
    float result = 0.0f;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int tape = __builtin_amdgcn_readlane(my_tape, leader);
        const bool mine = filled && my_tape == tape;
        const uint64_t grp = ballot(mine);
        todo &= ~grp;

        /* :1021-1031 — value first, then the unit partials (unused axes alias slot 0) */
        float rr;
        if (VS) {
            rr = interp_normals_asm_vgpr(tro, (uint32_t)(tape + 1), lane, a.groups ? 1u : 0u, decb, sx, sy, sz,
                                         isv ? vx : (comp == 0 ? 1.0f : 0.0f), isv ? vy : (comp == 1 ? 1.0f : 0.0f),
                                         isv ? vz : (comp == 2 ? 1.0f : 0.0f));
        } else {
            *reinterpret_cast<float*>(myslot + sx * 256) = isv ? vx : 0.0f;
            *reinterpret_cast<float*>(myslot + sy * 256) = isv ? vy : 0.0f;
            *reinterpret_cast<float*>(myslot + sz * 256) = isv ? vz : 0.0f;
            if (comp == 0) *reinterpret_cast<float*>(myslot + sx * 256) = 1.0f;
            if (comp == 1) *reinterpret_cast<float*>(myslot + sy * 256) = 1.0f;
            if (comp == 2) *reinterpret_cast<float*>(myslot + sz * 256) = 1.0f;
            const uint32_t rslot = interp_normals_asm(tro, (uint32_t)(tape + 1), smem, lane, isv, a.groups ? 1u : 0u, decb);
            rr = *reinterpret_cast<const float*>(myslot + rslot * 256);
        }
        if (mine) result = rr;
    }

    /* :1123-1131 */
    const float gx = quad_bcast_a(result, 0), gy = quad_bcast_a(result, 1), gz = quad_bcast_a(result, 2);
    const float norm = __builtin_sqrtf(gx * gx + gy * gy + gz * gz);
    const uint32_t u = f2u8((result / norm) * 127 + 128);
    const uint32_t ux = mpr_f2u(quad_bcast_a(mpr_u2f(u), 0)), uy = mpr_f2u(quad_bcast_a(mpr_u2f(u), 1)),
                   uz = mpr_f2u(quad_bcast_a(mpr_u2f(u), 2));
    if (filled && comp == 0) a.output[pxy] = (0xFFu << 24) | (uz << 16) | (uy << 8) | ux;
}

/* ---- the pass on the ROOT tape's generated code (tile_gen.hpp: TileGen::deriv) ----
 * A pixel's own tape — what the reference walks here — is the root tape with the decisions of its 16^3 tile (which shortened
 * the tape it pushed) and of its 4^3 tile (recorded by the last stage, in that shortened tape's numbering) applied.  With both
 * sets of decisions at hand as bits over the ROOT tape's min / max clauses, every pixel can run the one piece of code that
 * exists once per tape: no interpreter, and one walk per footprint whatever tiles its 16 pixels belong to. */
#define NG_ADDR(lo, hi, label) "s_add_u32 s" #lo ", s40, " label "_%=-L_pc_%=\n s_addc_u32 s" #hi ", s41, 0\n"
#define NG_CALLC(sym)                                                                            \
    "s_getpc_b64 s[40:41]\n"                                                                     \
    "s_add_u32 s40, s40, " sym "@rel32@lo+4\n"                                                   \
    "s_addc_u32 s41, s41, " sym "@rel32@hi+12\n"                                                 \
    "s_swappc_b64 s[30:31], s[40:41]\n"                                                          \
    "v_mov_b32 v37, v0\n s_setpc_b64 s[70:71]\n"
DEV float normals_gen_walk(const uint32_t* code, uint32_t ax, uint32_t ay, uint32_t az, float xin, float yin, float zin,
                           uint32_t dl0, uint32_t dl1, uint32_t dr0, uint32_t dr1, uint64_t all_l = 0, uint64_t all_r = 0)
{
    const uint32_t clo = rdfirst((uint32_t)(uintptr_t)code), chi = rdfirst((uint32_t)((uintptr_t)code >> 32));
    /* min / max clauses every lane of the wavefront has decided for the lhs / rhs: the guarded walk jumps over what they leave dead */
    const uint32_t al0 = rdfirst((uint32_t)all_l), al1 = rdfirst((uint32_t)(all_l >> 32)), ar0 = rdfirst((uint32_t)all_r), ar1 = rdfirst((uint32_t)(all_r >> 32));
    ax = rdfirst(ax);
    ay = rdfirst(ay);
    az = rdfirst(az);
    float res;
    asm volatile(
        "s_mov_b32 s90, 0x260\n"
        "s_mov_b32 s98, 0x88888888\n"
        "s_mov_b32 s99, 0x88888888\n"
        "s_set_gpr_idx_on %[ax], gpr_idx(DST)\n v_mov_b32 v50, %[xin]\n s_set_gpr_idx_off\n"
        "s_set_gpr_idx_on %[ay], gpr_idx(DST)\n v_mov_b32 v50, %[yin]\n s_set_gpr_idx_off\n"
        "s_set_gpr_idx_on %[az], gpr_idx(DST)\n v_mov_b32 v50, %[zin]\n s_set_gpr_idx_off\n"
        "v_mov_b32 v74, %[dl0]\n v_mov_b32 v75, %[dl1]\n v_mov_b32 v76, %[dr0]\n v_mov_b32 v77, %[dr1]\n"
        "s_mov_b32 s64, %[al0]\n s_mov_b32 s65, %[al1]\n s_mov_b32 s66, %[ar0]\n s_mov_b32 s67, %[ar1]\n"
        "s_getpc_b64 s[40:41]\n"
        "L_pc_%=:\n"
        NG_ADDR(72, 73, "L_div") NG_ADDR(74, 75, "L_sqrt") NG_ADDR(76, 77, "L_exp") NG_ADDR(78, 79, "L_log") NG_ADDR(68, 69, "L_sincos")
        NG_ADDR(80, 81, "L_gasin") NG_ADDR(82, 83, "L_gacos") NG_ADDR(84, 85, "L_gatan")
        "s_mov_b32 s34, %[clo]\n"
        "s_mov_b32 s35, %[chi]\n"
        "s_swappc_b64 s[38:39], s[34:35]\n"
        "v_mov_b32 %[res], v37\n"
        "s_branch L_end_%=\n"
        /* shared routines: argument v35 (and v36), result v37, return to s[70:71] */
        ".p2align 8\n"
        "L_div_%=:\n" MPR_ASM_DIV_BODY "s_setpc_b64 s[70:71]\n"
        "L_sqrt_%=:\n" MPR_ASM_SQRT_BODY "s_setpc_b64 s[70:71]\n" MPR_ASM_SQRT_TAIL
        "L_exp_%=:\n" MPR_ASM_EXP_BODY "s_setpc_b64 s[70:71]\n" MPR_ASM_EXP_TAIL
        "L_log_%=:\n" MPR_ASM_LOG_BODY "s_setpc_b64 s[70:71]\n" MPR_ASM_LOG_TAIL
        "L_sincos_%=:\n" MPR_ASM_SINCOS_BODY "s_setpc_b64 s[70:71]\n"
        "L_gasin_%=:\n" NG_CALLC("mpr_nq_asin")
        "L_gacos_%=:\n" NG_CALLC("mpr_nq_acos")
        "L_gatan_%=:\n" NG_CALLC("mpr_nq_atan")
        "L_end_%=:\n"
        : [res] "=&v"(res)
        : [ax] "s"(ax), [ay] "s"(ay), [az] "s"(az), [xin] "v"(xin), [yin] "v"(yin), [zin] "v"(zin), [dl0] "v"(dl0), [dl1] "v"(dl1),
          [dr0] "v"(dr0), [dr1] "v"(dr1), [clo] "s"(clo), [chi] "s"(chi), [al0] "s"(al0), [al1] "s"(al1), [ar0] "s"(ar0), [ar1] "s"(ar1)
        : "memory", "vcc", "scc",
          "s34", "s35", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71",
          "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89",
          "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s98", "s99",
          "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48",
          "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15",
          "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31",
          "s0", "s1", "s2", "s3", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17", "s18", "s19",
          "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31",
          NQ_V10(5), NQ_V10(6), "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77");
    return res;
}
#undef NG_ADDR
#undef NG_CALLC

__global__ void __launch_bounds__(64)
k_eval_normals_gen(NormalArgs a)
{
    const int lane = threadIdx.x;
    const int S = a.size;
    const int fside = S / 4;                                  /* footprints per side */
    int fxi, fyi;
    if (a.col_list) {
        const int col = a.col_list[blockIdx.x >> 8], f = blockIdx.x & 255, cols = S / 64;
        fxi = (col % cols) * 16 + (f & 15);
        fyi = (col / cols) * 16 + (f >> 4);
    } else {
        fxi = blockIdx.x % fside;
        fyi = blockIdx.x / fside;
    }
    const int pix = lane >> 2, comp = lane & 3;
    const bool isv = comp == 3;
    const int px = fxi * 4 + (pix & 3), py = fyi * 4 + (pix >> 2);
    const int pxy = px + py * S;
    int pz = a.image[pxy];
    const bool filled = pz != 0;
    if (ballot(filled) == 0) return;
    if (pz < S - 1) pz += 1;                                   /* :1003-1005 */

    const float size_recip = 1.0f / (float)(unsigned)S;
    const float fx = ((px + 0.5f) * size_recip - 0.5f) * 2.0f;
    const float fy = ((py + 0.5f) * size_recip - 0.5f) * 2.0f;
    const float fz = ((pz + 0.5f) * size_recip - 0.5f) * 2.0f;
    const float fw = a.mat[3] * fx + a.mat[7] * fy + a.mat[11] * fz + a.mat[15];
    const float vx = (a.mat[0] * fx + a.mat[4] * fy + a.mat[8] * fz + a.mat[12]) / fw;
    const float vy = (a.mat[1] * fx + a.mat[5] * fy + a.mat[9] * fz + a.mat[13]) / fw;
    const float vz = (a.mat[2] * fx + a.mat[6] * fy + a.mat[10] * fz + a.mat[14]) / fw;

    /* the pixel's 16^3 tile (index in its stage's list, and the tape that tile pushed: 0 = none) and 4^3 tile (:1034-1066) */
    int my_sub = -1, sub_tape = 0, my_micro = -1, my_top = -1;
    unsigned long long parent_l = 0, parent_r = 0;             /* what the pixel's 64^3 tile decided in a frame that skipped that stage */
    if (filled) {
        const int t64 = S / 64;
        const int tile = px / 64 + (py / 64) * t64 + (pz / 64) * t64 * t64;
        if (a.skip0_parents) {
            const unsigned long long* const pi = a.skip0_parents + (size_t)tile * SKIP0_INFO_U64;
            if (pi[2] == (unsigned long long)SKIP0_AMBIGUOUS) {
                parent_l = pi[0];
                parent_r = pi[1];
            }
        }
        const mpr_tile_node tn = a.tiles[tile];
        if (tn.next == -1) {
            if (tn.tape != 0 && a.gen_decisions0) my_top = tile;           /* a 64^3 tile that pushed a tape and was decided afterwards */
        } else {
            my_sub = tn.next * 64 + (px % 64) / 16 + ((py % 64) / 16) * 4 + ((pz % 64) / 16) * 16;
            const mpr_tile_node sn = a.subtiles[my_sub];
            sub_tape = sn.tape;
            if (sn.next != -1) my_micro = sn.next * 64 + (px % 16) / 4 + ((py % 16) / 4) * 4 + ((pz % 16) / 4) * 16;
        }
    }
    const unsigned long long all = a.gen_nchoices >= 64 ? ~0ull : ((1ull << a.gen_nchoices) - 1ull);
    unsigned long long dl = 0, dr = 0;
    /* pixels with a 4^3 tile: the 16^3 tile's decisions, then the 4^3 tile's, renumbered from the clauses its tape keeps to the
     * root tape's; one (16^3 tile, 4^3 tile) at a time — a footprint meets a handful */
    uint64_t pending = a.gen_decisions2 ? 0 : ballot(my_micro >= 0);
    if (a.gen_decisions2 && my_micro >= 0) {
        /* the last stage pushed: the pixel's 4^3 tile has everything in its own record */
        const unsigned long long* const rec = a.gen_decisions2 + (size_t)my_micro * GEN_RECORD_U64;
        dl = rec[0];
        dr = rec[1];
    }
    while (pending) {
        const int leader = __ffsll((long long)pending) - 1;
        const int micro = __builtin_amdgcn_readlane(my_micro, leader);
        const int sub = __builtin_amdgcn_readlane(my_sub, leader);
        const int stape = __builtin_amdgcn_readlane(sub_tape, leader);
        pending &= ~ballot(my_micro == micro);
        const int g = micro >> 6, child = micro & 63;
        unsigned long long L = 0, R = 0, K = all;
        if (stape != 0) {
            const unsigned long long* const rec = a.gen_decisions + (size_t)sub * GEN_RECORD_U64;
            L = rec[0];
            R = rec[1];
            K = rec[2];
        }
        const GroupInfo gi = a.groups[g];
        uint64_t l0 = 0, r0 = 0;
        if ((gi.pushed >> child) & 1ull) {
            const ulonglong2* const m = a.choice_masks + (size_t)g * a.choice_cap;
            const int n = gi.nchoices < 64 ? gi.nchoices : 64;
            ulonglong2 w0 = make_ulonglong2(0ull, 0ull);
            if (lane < n) w0 = m[lane];
            l0 = ballot((w0.x >> child) & 1ull);
            r0 = ballot((w0.y >> child) & 1ull);
        }
        /* lane k: root clause k is the j-th the 16^3 tile's tape keeps */
        const bool kept = (K >> lane) & 1ull;
        const int j = __popcll(K & ((1ull << lane) - 1ull));
        const uint64_t DL = L | ballot(kept && ((l0 >> j) & 1ull));
        const uint64_t DR = R | ballot(kept && ((r0 >> j) & 1ull));
        if (my_micro == micro) {
            dl = DL;
            dr = DR;
        }
    }
    /* pixels whose 16^3 tile was not subdivided but pushed a tape before it was decided: that tape's decisions */
    pending = ballot(my_micro < 0 && my_sub >= 0 && sub_tape != 0);
    while (pending) {
        const int leader = __ffsll((long long)pending) - 1;
        const int sub = __builtin_amdgcn_readlane(my_sub, leader);
        const bool mine = my_micro < 0 && my_sub == sub;
        pending &= ~ballot(mine);
        const unsigned long long* const rec = a.gen_decisions + (size_t)sub * GEN_RECORD_U64;
        if (mine) {
            dl = rec[0];
            dr = rec[1];
        }
    }

    if (my_top >= 0) {
        const unsigned long long* const rec = a.gen_decisions0 + (size_t)my_top * GEN_RECORD_U64;
        dl = rec[0];
        dr = rec[1];
    }
    dl |= parent_l & ~dr;
    dr |= parent_r & ~dl;

    const uint64_t head0 = a.tape_ro[0];
    const uint32_t sx = (head0 >> 8) & 0xFF, sy = (head0 >> 16) & 0xFF, sz = (head0 >> 24) & 0xFF;
    /* :1021-1031 — value first, then the unit partials */
    /* what EVERY filled pixel of the footprint has decided (pixels without a surface decide nothing and see nothing): the AND over
     * the distinct (dl, dr) pairs among them — a footprint meets a handful of tiles */
    uint64_t all_l = 0, all_r = 0;
    if (a.gen_code_guarded) {
        all_l = all_r = ~0ull;
        uint64_t todo = ballot(filled);
        while (todo) {
            const uint32_t leader = (uint32_t)(__ffsll((long long)todo) - 1);
            const uint64_t vl = rdlane64(dl, leader), vr = rdlane64(dr, leader);
            all_l &= vl;
            all_r &= vr;
            todo &= ~ballot(filled && dl == vl && dr == vr);
        }
        if (!filled) {           /* their lanes walk along (and store nothing): let them take the same branches of every min / max */
            dl = all_l;
            dr = all_r;
        }
    }
    const float result = normals_gen_walk(a.gen_code_guarded ? a.gen_code_guarded : a.gen_code, sx, sy, sz, isv ? vx : (comp == 0 ? 1.0f : 0.0f),
                                          isv ? vy : (comp == 1 ? 1.0f : 0.0f), isv ? vz : (comp == 2 ? 1.0f : 0.0f), (uint32_t)dl,
                                          (uint32_t)(dl >> 32), (uint32_t)dr, (uint32_t)(dr >> 32), all_l, all_r);

    /* :1123-1131 */
    const float gx = quad_bcast_a(result, 0), gy = quad_bcast_a(result, 1), gz = quad_bcast_a(result, 2);
    const float norm = __builtin_sqrtf(gx * gx + gy * gy + gz * gz);
    const uint32_t u = f2u8((result / norm) * 127 + 128);
    const uint32_t ux = mpr_f2u(quad_bcast_a(mpr_u2f(u), 0)), uy = mpr_f2u(quad_bcast_a(mpr_u2f(u), 1)),
                   uz = mpr_f2u(quad_bcast_a(mpr_u2f(u), 2));
    if (filled && comp == 0) a.output[pxy] = (0xFFu << 24) | (uz << 16) | (uy << 8) | ux;
}

void launch_eval_normals_asm(hipStream_t s, const NormalArgs& a)
{
    const int fside = a.size / 4;
    const int groups = a.col_list ? a.ncols * 256 : fside * fside;
    if (groups <= 0) return;
    /* slots in registers when the LDS slot file would hold a CU under the 12 wavefronts those registers allow (tried for
     * small slot files too: bear, 23 slots, 0.378 -> 0.420 ms — one wavefront per SIMD fewer, and the index switching) */
    if (a.gen_code && a.gen_decisions && (a.groups || a.gen_decisions2)) {
        hipLaunchKernelGGL(k_eval_normals_gen, dim3(groups), dim3(64), 0, s, a);
        return;
    }
    const size_t lds = (size_t)a.nslots * 256 + (a.groups ? 768 : 0);
    if (a.vgpr_slots && a.nslots <= NQ_VS_MAX_SLOTS && lds > (size_t)160 * 1024 / 12)
        hipLaunchKernelGGL(k_eval_normals_asm<NQ_VS_MAX_SLOTS>, dim3(groups), dim3(64), (size_t)(a.groups ? 768 : 16), s, a);
    else
        hipLaunchKernelGGL(k_eval_normals_asm<0>, dim3(groups), dim3(64), lds, s, a);
}

}  // namespace mprk
