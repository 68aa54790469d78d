/*
 * kernels_voxel_asm.hip — the float voxel / pixel pass (reference src/context.cu:707-964,
 * eval_voxels_f + calculate_voxels / calculate_pixels) with the tape interpreter written
 * directly in gfx950 assembly.
 *
 * Why: the compiled interpreter (kernels_float.hip, k_eval_voxels) spends ~16 scalar
 * instructions per clause on loop control and on the opcode decision tree, and a CU has ONE
 * scalar ALU for its four SIMDs (0.95 instr/clk/CU, scripts/ubench/issue_rates.hip): the pass
 * is bound by scalar issue, not by VALU, LDS or HBM.  Here a clause costs 2 scalar instructions:
 *
 *   - clauses are fetched 63 at a time (one coalesced load, lane j holds clause j); per block the
 *     VALU computes, for all 63 clauses at once, the ADDRESS of each clause's handler
 *     (table base + op * 128) into a VGPR pair; lane 63 always holds the address of the
 *     "fetch the next block" handler, so running off the end of a block needs no test;
 *   - per clause: s_add (clause counter), 4 x v_readlane (handler address, clause lo / hi),
 *     s_setpc_b64.  The handlers are threaded code: each ends with that same sequence;
 *   - operand addresses in LDS come from one v_perm_b32 each: byte 0 = lane * 4, byte 1 = slot
 *     (slot s of lane l lives at s * 256 + l * 4; one wave per workgroup, LDS base 0);
 *   - immediates are used straight from the SGPR the clause was read into.
 *
 * Arithmetic is the same instruction selection the compiler makes for the C++ interpreter
 * (v_add/v_mul/v_min/v_max/v_sub_f32, the IEEE division and square-root expansions), so the
 * results are bit-identical to k_eval_voxels and to the oracle; tests/test_gpu_render.py compares
 * both kernels frame by frame.  The inverse trigonometric opcodes (asin, acos, atan) leave the assembly block, are evaluated by the shared C++ routines of mpr_fmath.h, and
 * re-enter it.
 *
 * Software-visible hazards of gfx940/gfx950 that the assembler does not fix up in inline asm
 * (LLVM GCNHazardRecognizer): VALU-written SGPR/VCC -> VALU read needs 2 wait states,
 * VALU-written VGPR -> v_readlane of it 1, transcendental result -> VALU use 1, VALU-written VCC
 * -> v_div_fmas 4.  The s_nop's below are those.
 */
#include "asm_float_bodies.hpp"
#include "kernel_common.hpp"

namespace mprk {

/* out of line, one function per opcode: the switch below runs on the scalar unit (op is uniform) */
__device__ __noinline__ float na_sin(float v) { return mpr_sinf(v); }
__device__ __noinline__ float na_cos(float v) { return mpr_cosf(v); }
/* asin / acos / atan are also called from inside the assembly loop (MPR_CALL) under these names: leaf
 * functions of the AMDGPU calling convention, argument and result in v0, return address s[30:31] */
__device__ __attribute__((noinline, used)) float na_asin(float v) __asm__("mpr_fa_asin");
__device__ __attribute__((noinline, used)) float na_acos(float v) __asm__("mpr_fa_acos");
__device__ __attribute__((noinline, used)) float na_atan(float v) __asm__("mpr_fa_atan");
__device__ float na_asin(float v) { return mpr_asinf(v); }
__device__ float na_acos(float v) { return mpr_acosf(v); }
__device__ float na_atan(float v) { return mpr_atanf(v); }
__device__ __noinline__ float na_exp(float v) { return mpr_expf(v); }
__device__ __noinline__ float na_log(float v) { return mpr_logf(v); }
DEV float rare_unary_a(uint32_t op, float v)
{
    switch (op) {
        case MPR_OP_SIN_LHS: return na_sin(v);
        case MPR_OP_COS_LHS: return na_cos(v);
        case MPR_OP_ASIN_LHS: return na_asin(v);
        case MPR_OP_ACOS_LHS: return na_acos(v);
        case MPR_OP_ATAN_LHS: return na_atan(v);
        case MPR_OP_EXP_LHS: return na_exp(v);
        default: return na_log(v);
    }
}

/* Fixed registers of the interpreter (declared as clobbers):
 *   s[80:81] handler address      s[82:83] handler table base     s[84:85] block address
 *   s86 / s87 clause lo / hi      s88 clause counter in block     s89 block base (clause index)
 *   s90 0x260 (class mask)        s91, s[92:95] scratch (compare masks, v_div_scale sdst)
 *   v32 aA  v33 aB  v34 aO  v35 A  v36 B  v37 result of the clause (and of the previous one)
 *   v38..v44 temporaries                                                                       */
#define MPR_DISPATCH                                   \
    "s_add_u32 s88, s88, 1\n"                          \
    "v_readlane_b32 s86, %[blo], s88\n"                \
    "s_and_b32 s80, s86, s96\n"                        \
    "s_add_u32 s80, s80, s82\n"                        \
    "s_addc_u32 s81, s83, 0\n"                         \
    "s_setpc_b64 s[80:81]\n"
/* the immediate (or jump distance) is fetched only by the handlers that use it */
#define MPR_IMM "v_readlane_b32 s87, %[bhi], s88\n"
#define MPR_AL "v_perm_b32 v32, s86, %[lb], %[selL]\n ds_read_b32 v35, v32\n"
#define MPR_AR "v_perm_b32 v33, s86, %[lb], %[selR]\n ds_read_b32 v36, v33\n"
#define MPR_AO "v_perm_b32 v34, s86, %[lb], %[selO]\n"
#define MPR_W "s_waitcnt lgkmcnt(0)\n"
#define MPR_ST "ds_write_b32 v34, v37\n"
#define MPR_H(v, n) ".p2align 8\nL_h" #v "_" #n "_%=:\n"
#define MPR_EXIT "s_branch L_exit_%=\n"
#define MPR_H30(LDL, WL, MVA) MPR_EXIT      /* a word that is not an opcode: evaluated (as a logarithm, like k_eval_voxels) outside the block */
#define MPR_VS_ENTER
#define MPR_VS_LEAVE
#define MPR_END MPR_ST MPR_DISPATCH
/* v37 = sym(v35) by a compiled routine; v34 (address of the out slot) survives in v44.  The routines
 * keep v40..v47 and every SGPR from s34 up (calling convention), where the interpreter's state is. */
#define MPR_CALL(sym)                                                                            \
    "v_mov_b32 v44, v34\n v_mov_b32 v0, v35\n"                                                   \
    "s_getpc_b64 s[40:41]\n"                                                                     \
    "s_add_u32 s40, s40, " sym "@rel32@lo+4\n"                                                   \
    "s_addc_u32 s41, s41, " sym "@rel32@hi+12\n"                                                 \
    "s_swappc_b64 s[30:31], s[40:41]\n"                                                          \
    "v_mov_b32 v37, v0\n v_mov_b32 v34, v44\n" MPR_ST MPR_DISPATCH

/* One handler table, 32 entries of 128 bytes, indexed by opcode.  There are three of them:
 *   table 0: operands come from the slot file in LDS;
 *   table 1: the lhs slot is the previous clause's out slot -> take it from v37, no LDS read;
 *   table 2: the rhs slot is.
 * (70-80% of the clauses of the benchmark models consume the previous result.)  Every result is
 * still written to its slot, so a table-0 handler is always correct; which table a clause uses
 * is decided per 63-clause block on the VALU (see L_load) and costs nothing per clause.
 *   LDL / LDR  load lhs / rhs (or nothing), A / B the register the operand is then in,
 *   WL / WR / WLR  s_waitcnt if lhs / rhs / either was loaded,
 *   MVA / MVB  copy a forwarded operand into v35 / v36 for the shared long sequences,
 *   NL / NR  the wait state still owed between the v_readlane of the immediate and its first
 *   VALU use when no load of lhs / rhs sits in between,
 *   AO / END  tables 3-5 are tables 0-2 for clauses whose result dies in the next clause (which takes
 *   it from v37 and overwrites the same slot — half of the clauses of the benchmark models): the
 *   one-instruction handlers neither compute the slot's address (a s_nop keeps the wait states) nor
 *   store.                                                                                       */
#define MPR_TABLE(v, LDL, A, LDR, B, WL, WR, WLR, MVA, MVB, NL, NR, AO, END)                                               \
    MPR_H(v, 0) MPR_EXIT                                              /* end of tape */                   \
    MPR_H(v, 1)                                                       /* JUMP: base += j + imm + 1 */     \
    MPR_IMM "s_add_u32 s89, s89, s88\n s_add_u32 s89, s89, s87\n s_add_u32 s89, s89, 1\n s_branch L_load_%=\n"   \
    MPR_H(v, 2) LDL AO WL "v_mul_f32 v37, " A ", " A "\n" END                                     \
    MPR_H(v, 3) LDL MPR_AO WL MVA "s_branch L_sqrt_%=\n"                                                  \
    MPR_H(v, 4) LDL AO WL "v_xor_b32 v37, 0x80000000, " A "\n" END                                \
    MPR_H(v, 5) LDL MPR_AO WL MVA "s_branch L_sin_%=\n"                                                  \
    MPR_H(v, 6) LDL MPR_AO WL MVA "s_branch L_cos_%=\n"                                                  \
    MPR_H(v, 7) LDL MPR_AO WL MVA "s_branch L_casin_%=\n"                                                \
    MPR_H(v, 8) LDL MPR_AO WL MVA "s_branch L_cacos_%=\n"                                                \
    MPR_H(v, 9) LDL MPR_AO WL MVA "s_branch L_catan_%=\n"                                                \
    MPR_H(v, 10) LDL MPR_AO WL MVA "s_branch L_exp_%=\n"                                                  \
    MPR_H(v, 11) LDL AO WL "v_and_b32 v37, 0x7fffffff, " A "\n" END                               \
    MPR_H(v, 12) LDL MPR_AO WL MVA "s_branch L_log_%=\n"                                                  \
    MPR_H(v, 13) MPR_IMM LDL AO WL NL "v_add_f32 v37, s87, " A "\n" END                                      \
    MPR_H(v, 14) LDL LDR AO WLR "v_add_f32 v37, " A ", " B "\n" END                               \
    MPR_H(v, 15) MPR_IMM LDL AO WL NL "v_mul_f32 v37, s87, " A "\n" END                                      \
    MPR_H(v, 16) LDL LDR AO WLR "v_mul_f32 v37, " A ", " B "\n" END                               \
    /* min / max: operands canonicalised first (a signalling NaN loses against a number, like fminf) */  \
    MPR_H(v, 17) MPR_IMM LDL AO NL "v_max_f32 v36, s87, s87\n" WL "v_max_f32 v35, " A ", " A "\n v_min_f32 v37, v35, v36\n" END \
    MPR_H(v, 18) LDL LDR AO WLR "v_max_f32 v35, " A ", " A "\n v_max_f32 v36, " B ", " B "\n v_min_f32 v37, v35, v36\n" END \
    MPR_H(v, 19) MPR_IMM LDL AO NL "v_max_f32 v36, s87, s87\n" WL "v_max_f32 v35, " A ", " A "\n v_max_f32 v37, v35, v36\n" END \
    MPR_H(v, 20) LDL LDR AO WLR "v_max_f32 v35, " A ", " A "\n v_max_f32 v36, " B ", " B "\n v_max_f32 v37, v35, v36\n" END \
    MPR_H(v, 21) MPR_IMM LDL AO WL NL "v_subrev_f32 v37, s87, " A "\n" END          /* lhs - imm */         \
    MPR_H(v, 22) MPR_IMM LDR AO WR NR "v_sub_f32 v37, s87, " B "\n" END             /* imm - rhs */         \
    MPR_H(v, 23) LDL LDR AO WLR "v_sub_f32 v37, " A ", " B "\n" END                               \
    MPR_H(v, 24) MPR_IMM LDL MPR_AO WL MVA "v_mov_b32 v36, s87\n s_branch L_div_%=\n"    /* lhs / imm */         \
    MPR_H(v, 25) MPR_IMM LDR MPR_AO WR MVB "v_mov_b32 v35, s87\n s_branch L_div_%=\n"    /* imm / rhs */         \
    MPR_H(v, 26) LDL LDR MPR_AO WLR MVA MVB "s_branch L_div_%=\n"                                         \
    MPR_H(v, 27) MPR_IMM AO "s_nop 0\n v_mov_b32 v37, s87\n" END                 /* COPY_IMM */          \
    MPR_H(v, 28) LDL AO WL "v_mov_b32 v37, " A "\n" END                  /* COPY_LHS */          \
    MPR_H(v, 29) LDR AO WR "v_mov_b32 v37, " B "\n" END                  /* COPY_RHS */          \
    MPR_H(v, 30) MPR_H30(LDL, WL, MVA)                                            /* not an opcode */     \
    MPR_H(v, 31) "s_add_u32 s89, s89, 63\n s_branch L_load_%=\n"                  /* lane 63: next block */

/* Walks the tape whose first clause is tro[first] over the slot file at LDS offset 0 (slot s of
 * lane l at s * 256 + l * 4) and returns the result slot named by the end clause. */
/* The walk itself, as text: expanded in interp_asm (slots in LDS) and in interp_asm_vgpr (slots in VGPRs), each time
 * with that variant's MPR_AL / MPR_AR / MPR_AO / MPR_ST / MPR_H30 / MPR_VS_ENTER / MPR_VS_LEAVE. */
#define MPR_ASM_TEXT \
    MPR_VS_ENTER \
    "s_mov_b32 s89, %[base]\n" \
    "s_mov_b32 s88, %[sj]\n" \
    "s_mov_b32 s90, 0x260\n" \
    "s_mov_b32 s96, 0xff00\n" \
    "s_mov_b32 s91, 0\n"                            /* no heightmap value in flight */ \
    "s_mov_b32 %[ab], 0\n" \
    "v_mov_b32 v37, %[prev]\n" \
    "s_getpc_b64 s[82:83]\n" \
    "L_pc_%=:\n" \
    "s_add_u32 s82, s82, L_h0_0_%=-L_pc_%=\n" \
    "s_addc_u32 s83, s83, 0\n" \
    "s_cmp_eq_u32 %[mode], 0\n" \
    "s_cbranch_scc1 L_load_%=\n" \
    "s_cmp_eq_u32 %[mode], 2\n" \
    "s_cbranch_scc1 L_loaded_%=\n" \
    MPR_DISPATCH \
    /* ---- fetch 63 clauses at s89, build the handler addresses ---- */ \
    "L_load_%=:\n" \
    "s_mov_b32 s84, s89\n" \
    "s_mov_b32 s85, 0\n" \
    "s_lshl_b64 s[84:85], s[84:85], 3\n" \
    "s_add_u32 s84, s84, %[tlo]\n" \
    "s_addc_u32 s85, s85, %[thi]\n" \
    "global_load_dword %[blo], %[lane8], s[84:85]\n" \
    "global_load_dword %[bhi], %[lane8], s[84:85] offset:4\n" \
    "s_mov_b32 s91, %[rck]\n" \
    "s_cmp_eq_u32 s91, 0\n" \
    "s_cbranch_scc1 L_loaded_%=\n" \
    "s_mov_b32 s84, %[ilo]\n"                       /* the lane's heightmap entry, past the L1 */ \
    "s_mov_b32 s85, %[ihi]\n" \
    "global_load_dword v45, %[ioff], s[84:85] sc1\n" \
    "L_loaded_%=:\n" \
    "s_mov_b32 s88, -1\n" \
    "v_mov_b32 v41, 0\n" \
    "v_mov_b32 v43, 32\n" \
    "v_mov_b32 v44, 64\n" \
    "s_waitcnt vmcnt(0)\n" \
    "s_cmp_eq_u32 s91, 0\n" \
    "s_cbranch_scc1 L_nocheck_%=\n" \
    "v_cmp_ge_i32 vcc, v45, %[thr]\n"               /* hidden by now? */ \
    "s_mov_b32 s91, 0\n" \
    "s_cmp_eq_u64 vcc, exec\n" \
    "s_cbranch_scc1 L_abort_%=\n" \
    "L_nocheck_%=:\n" \
    "v_bfe_u32 v40, %[blo], 8, 8\n"                 /* out slot */ \
    "v_and_b32 v38, 0xff, %[blo]\n" \
    "v_min_u32 v38, 30, v38\n"                       /* opcode; unknown ones -> handler 30 */ \
    "v_mov_b32_dpp v41, v40 wave_shr:1 row_mask:0xf bank_mask:0xf\n"   /* out slot of the previous clause (lane 0: none) */ \
    "v_bfe_u32 v42, %[blo], 16, 8\n"                /* lhs slot */ \
    "v_lshrrev_b32 v39, 24, %[blo]\n"               /* rhs slot */ \
    "v_cmp_eq_u32 s[92:93], v39, v41\n" \
    "v_cmp_eq_u32 vcc, v42, v41\n" \
    "v_cmp_ne_u32 s[94:95], 0, v41\n" \
    "v_cndmask_b32 v42, 0, v44, s[92:93]\n"          /* rhs forwarded: table 2 */ \
    "v_cndmask_b32 v42, v42, v43, vcc\n"             /* lhs forwarded: table 1 */ \
    "v_cndmask_b32 v42, 0, v42, s[94:95]\n" \
    /* this clause takes the previous result from v37 (exactly one operand), writes the same slot \
    * and is an ordinary opcode: the previous clause need not store (tables 3-5 = +96) */ \
    "s_xor_b64 s[40:41], s[92:93], vcc\n" \
    "s_and_b64 s[40:41], s[40:41], s[94:95]\n" \
    "v_cmp_eq_u32 s[42:43], v40, v41\n" \
    "v_add_u32 v45, -2, v38\n" \
    "v_cmp_gt_u32 s[44:45], 28, v45\n" \
    "v_cmp_ne_u32 vcc, 0x1f8, %[lane8]\n"           /* lane 63's clause runs as lane 0 of the next block: from LDS */ \
    "s_and_b64 s[40:41], s[40:41], s[42:43]\n" \
    "s_and_b64 s[44:45], s[44:45], vcc\n" \
    "s_and_b64 s[40:41], s[40:41], s[44:45]\n" \
    "v_mov_b32 v46, 0\n" \
    "v_mov_b32 v47, 0x60\n" \
    "v_cndmask_b32 v45, 0, v47, s[40:41]\n" \
    "v_add_u32 v38, v38, v42\n" \
    "s_nop 0\n" \
    "v_mov_b32_dpp v46, v45 wave_shl:1 row_mask:0xf bank_mask:0xf\n"   /* lane j <- lane j + 1 (lane 63: 0) */ \
    "v_cmp_eq_u32 vcc, 0x1f8, %[lane8]\n"           /* lane 63 -> handler 31 of table 0 */ \
    "v_mov_b32 v39, 31\n" \
    "v_add_u32 v38, v38, v46\n" \
    "v_cndmask_b32 v38, v38, v39, vcc\n"             /* handler index = table * 32 + opcode */ \
    /* clause word as the handlers see it: byte 0 out slot, byte 1 handler index, bytes 2, 3 lhs, rhs */ \
    "v_lshl_or_b32 v38, v38, 8, v40\n" \
    "v_and_b32 %[blo], 0xffff0000, %[blo]\n" \
    "v_or_b32 %[blo], %[blo], v38\n" \
    "s_nop 0\n" \
    MPR_DISPATCH \
    /* ---- handlers: three tables of 32 x 128 bytes ---- */ \
    MPR_TABLE(0, MPR_AL, "v35", MPR_AR, "v36", MPR_W, MPR_W, MPR_W, "", "", "", "", MPR_AO, MPR_END) \
    MPR_TABLE(1, "", "v37", MPR_AR, "v36", "", MPR_W, MPR_W, "v_mov_b32 v35, v37\n", "", "s_nop 0\n", "", MPR_AO, MPR_END) \
    MPR_TABLE(2, MPR_AL, "v35", "", "v37", MPR_W, "", MPR_W, "", "v_mov_b32 v36, v37\n", "", "s_nop 0\n", MPR_AO, MPR_END) \
    MPR_TABLE(3, MPR_AL, "v35", MPR_AR, "v36", MPR_W, MPR_W, MPR_W, "", "", "", "", "s_nop 0\n", MPR_DISPATCH) \
    MPR_TABLE(4, "", "v37", MPR_AR, "v36", "", MPR_W, MPR_W, "v_mov_b32 v35, v37\n", "", "s_nop 0\n", "", "s_nop 0\n", MPR_DISPATCH) \
    MPR_TABLE(5, MPR_AL, "v35", "", "v37", MPR_W, "", MPR_W, "", "v_mov_b32 v36, v37\n", "", "s_nop 0\n", "s_nop 0\n", MPR_DISPATCH) \
    /* ---- v37 = v35 / v36, correctly rounded ---- */ \
    ".p2align 7\n" \
    "L_div_%=:\n" \
    MPR_ASM_DIV_BODY \
    MPR_ST MPR_DISPATCH \
    /* ---- v37 = sqrt(v35), correctly rounded ---- */ \
    "L_sqrt_%=:\n" \
    MPR_ASM_SQRT_BODY \
    MPR_ST MPR_DISPATCH \
    MPR_ASM_SQRT_TAIL \
    /* ---- v37 = mpr_expf(v35) (include/mpr_fmath.h), same operations in the same order ---- */ \
    "L_exp_%=:\n" \
    MPR_ASM_EXP_BODY \
    MPR_ST MPR_DISPATCH \
    MPR_ASM_EXP_TAIL \
    /* ---- v37 = mpr_logf(v35) ---- */ \
    "L_log_%=:\n" \
    MPR_ASM_LOG_BODY \
    MPR_ST MPR_DISPATCH \
    MPR_ASM_LOG_TAIL \
    /* ---- v37 = mpr_sinf(v35) / mpr_cosf(v35) ---- */ \
    "L_sin_%=:\n" \
    MPR_ASM_SINCOS_BODY \
    MPR_ST MPR_DISPATCH \
    "L_cos_%=:\n" \
    MPR_ASM_SINCOS_BODY \
    "v_mov_b32 v37, v36\n" \
    MPR_ST MPR_DISPATCH \
    /* ---- v37 = mpr_asinf / mpr_acosf / mpr_atanf(v35): compiled routines, called ---- */ \
    "L_casin_%=:\n" MPR_CALL("mpr_fa_asin") \
    "L_cacos_%=:\n" MPR_CALL("mpr_fa_acos") \
    "L_catan_%=:\n" MPR_CALL("mpr_fa_atan") \
    /* ---- every lane is hidden: give up ---- */ \
    "L_abort_%=:\n" \
    "s_mov_b32 %[ab], 1\n" \
    "s_mov_b32 s86, 0\n" \
    /* ---- leave: end of tape, or an opcode evaluated in C++ ---- */ \
    "L_exit_%=:\n" \
    "s_waitcnt lgkmcnt(0)\n" \
    MPR_VS_LEAVE \
    "s_mov_b32 %[dlo], s86\n" \
    "s_mov_b32 %[dhi], s87\n" \
    "s_mov_b32 %[base], s89\n" \
    "s_mov_b32 %[sj], s88\n"

/* first_block: the 64 words at tro[first + lane], when the caller has already fetched them (under
 * other latencies of its prologue), else null */
/* image / img_off / hidden_at (3-D frames): whenever a further block of the tape is fetched, the lane's
 * heightmap entry (byte offset img_off) is read again with it; once every lane finds it at or above
 * hidden_at — the skip test of src/context.cu:852-864, which another tile of the column has made true
 * in the meantime — the walk stops and INTERP_ABORTED is returned: nothing this tile could still
 * write would change the image. */
constexpr uint32_t INTERP_ABORTED = 0xFFFFFFFFu;
DEV uint32_t interp_asm(const uint64_t* __restrict__ tro, uint32_t first, unsigned char* smem, int lane,
                        const uint64_t* first_block = nullptr, const int* image = nullptr, uint32_t img_off = 0,
                        int hidden_at = 0)
{
    unsigned char* const myslot = smem + lane * 4;
    /* state of the assembly interpreter that has to survive a trip through C++ */
    uint32_t blo = 0, bhi = 0;                       /* the clause block: lane j = clause j */
    uint32_t base = first, sj = 0, dlo = 0, dhi = 0;
    /* the v_perm_b32 address trick needs the dynamic LDS segment at offset 0 (no static LDS here) */
    const uint32_t lb = (uint32_t)(uintptr_t)smem + (uint32_t)lane * 4u;
    const uint32_t lane8 = (uint32_t)lane * 8u;
    /* v_perm_b32 selectors: byte 0 <- lb byte 0, byte 1 <- byte 0 (out) / 2 (lhs) / 3 (rhs) of the
     * clause word as rewritten per block (see L_load) */
    const uint32_t selO = to_vgpr(0x0c0c0400u), selL = to_vgpr(0x0c0c0600u), selR = to_vgpr(0x0c0c0700u);
    const uint32_t tlo = (uint32_t)(uintptr_t)tro, thi = (uint32_t)((uintptr_t)tro >> 32);
    float prev = 0.0f;
    const uint32_t ilo = rdfirst((uint32_t)(uintptr_t)image), ihi = rdfirst((uint32_t)((uintptr_t)image >> 32));
    const uint32_t recheck = rdfirst(image ? 1u : 0u);
    uint32_t aborted = 0;
    uint32_t mode = 0;                               /* 0: fetch the block at `base`; 1: continue after `sj`; 2: block is there */
    if (first_block) {
        blo = (uint32_t)*first_block;
        bhi = (uint32_t)(*first_block >> 32);
        mode = 2;
    }

    for (;;) {
        base = rdfirst(base);
        sj = rdfirst(sj);
        mode = rdfirst(mode);
        asm volatile(
            MPR_ASM_TEXT
            : [blo] "+&v"(blo), [bhi] "+&v"(bhi),          /* early clobber: an input of equal value must not share them */
              [base] "+&s"(base), [sj] "+&s"(sj), [dlo] "=&s"(dlo), [dhi] "=&s"(dhi), [ab] "=&s"(aborted)
            : [lb] "v"(lb), [selL] "v"(selL), [selR] "v"(selR), [selO] "v"(selO), [lane8] "v"(lane8),
              [tlo] "s"(tlo), [thi] "s"(thi), [mode] "s"(mode), [prev] "v"(prev),
              [ilo] "s"(ilo), [ihi] "s"(ihi), [rck] "s"(recheck), [ioff] "v"(img_off), [thr] "v"(hidden_at)
            : "memory", "vcc", "scc",
              "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96",
              "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47",
              "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47",
              /* what the called routines may use on top (MPR_CALL) */
              "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15",
              "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31",
              "s0", "s1", "s2", "s3", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17", "s18", "s19",
              "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31");
        /* dlo is the rewritten clause word: byte 0 out slot, byte 1 handler index (opcode in its low 5 bits) */
        if (aborted) return INTERP_ABORTED;
        const uint32_t op = (dlo >> 8) & 31;
        if (op == 0) break;
        /* sin .. log (and anything that is not an opcode, like k_eval_voxels) */
        const float A = *reinterpret_cast<const float*>(myslot + ((dlo >> 8) & 0xFF00));
        prev = rare_unary_a(op, A);      /* re-enters as "the previous clause's result" (operand forwarding) */
        *reinterpret_cast<float*>(myslot + ((dlo & 0xFF) << 8)) = prev;
        mode = 1;
    }
    return dlo & 0xFF;
}

/* The same walk with the slot file in VECTOR REGISTERS (slot s = v[48 + s], addressed through the scalar GPR index: see
 * tile_interp_asm.hpp), for tapes with many slots: 256 bytes of LDS per slot and wavefront leave a tape with 84 slots
 * 7 wavefronts per CU; 93 registers on top of the walk's own are 3 per SIMD, 12 per CU.  The axes' values arrive and the
 * result leaves inside the one asm statement; a word that is not an opcode is the logarithm it is outside (MPR_H30). */
#undef MPR_AL
#undef MPR_AR
#undef MPR_AO
#undef MPR_ST
#undef MPR_H30
#undef MPR_VS_ENTER
#undef MPR_VS_LEAVE
#define MPR_AL "s_bfe_u32 s60, s86, 0x80010\n s_set_gpr_idx_on s60, gpr_idx(SRC0)\n v_mov_b32 v35, v48\n s_set_gpr_idx_off\n"
#define MPR_AR "s_lshr_b32 s60, s86, 24\n s_set_gpr_idx_on s60, gpr_idx(SRC0)\n v_mov_b32 v36, v48\n s_set_gpr_idx_off\n"
#define MPR_AO "s_and_b32 s61, s86, 0xff\n"                       /* the out slot, kept for MPR_ST */
#define MPR_ST "s_set_gpr_idx_on s61, gpr_idx(DST)\n v_mov_b32 v48, v37\n s_set_gpr_idx_off\n"
#define MPR_H30(LDL, WL, MVA) LDL MPR_AO WL MVA "s_branch L_log_%=\n"
#define MPR_VS_ENTER                                                                                                   \
    "s_set_gpr_idx_on %[ax], gpr_idx(DST)\n v_mov_b32 v48, %[xin]\n s_set_gpr_idx_off\n"                              \
    "s_set_gpr_idx_on %[ay], gpr_idx(DST)\n v_mov_b32 v48, %[yin]\n s_set_gpr_idx_off\n"                              \
    "s_set_gpr_idx_on %[az], gpr_idx(DST)\n v_mov_b32 v48, %[zin]\n s_set_gpr_idx_off\n"
#define MPR_VS_LEAVE "s_and_b32 s60, s86, 0xff\n s_set_gpr_idx_on s60, gpr_idx(SRC0)\n v_mov_b32 %[res], v48\n s_set_gpr_idx_off\n"
#define MPR_V10(a) "v" #a "0", "v" #a "1", "v" #a "2", "v" #a "3", "v" #a "4", "v" #a "5", "v" #a "6", "v" #a "7", "v" #a "8", "v" #a "9"
constexpr int MPR_VS_MAX_SLOTS = 93;

/* ax / ay / az: the axes' slots (head clause); xin / yin / zin: their values; *res: the result; returns INTERP_ABORTED or 0 */
DEV uint32_t interp_asm_vgpr(const uint64_t* __restrict__ tro, uint32_t first, int lane, const uint64_t* first_block,
                             const int* image, uint32_t img_off, int hidden_at, uint32_t ax, uint32_t ay, uint32_t az,
                             float xin, float yin, float zin, float* res)
{
    uint32_t blo = (uint32_t)*first_block, bhi = (uint32_t)(*first_block >> 32);
    uint32_t base = rdfirst(first), sj = 0, dlo = 0, dhi = 0;
    const uint32_t lane8 = (uint32_t)lane * 8u;
    const uint32_t tlo = (uint32_t)(uintptr_t)tro, thi = (uint32_t)((uintptr_t)tro >> 32);
    const uint32_t ilo = rdfirst((uint32_t)(uintptr_t)image), ihi = rdfirst((uint32_t)((uintptr_t)image >> 32));
    const uint32_t recheck = rdfirst(image ? 1u : 0u);
    uint32_t aborted = 0;
    const uint32_t mode = 2, zero = 0;
    float r = 0.0f;
    ax = rdfirst(ax);
    ay = rdfirst(ay);
    az = rdfirst(az);
    asm volatile(
        MPR_ASM_TEXT
        : [blo] "+&v"(blo), [bhi] "+&v"(bhi), [base] "+&s"(base), [sj] "+&s"(sj), [dlo] "=&s"(dlo), [dhi] "=&s"(dhi), [ab] "=&s"(aborted),
          [res] "=&v"(r)
        : [lane8] "v"(lane8), [tlo] "s"(tlo), [thi] "s"(thi), [mode] "s"(mode), [prev] "s"(zero),
          [ilo] "s"(ilo), [ihi] "s"(ihi), [rck] "s"(recheck), [ioff] "v"(img_off), [thr] "v"(hidden_at),
          [ax] "s"(ax), [ay] "s"(ay), [az] "s"(az), [xin] "v"(xin), [yin] "v"(yin), [zin] "v"(zin)
        : "memory", "vcc", "scc",
          "s60", "s61", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96",
          "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47",
          "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47",
          "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15",
          "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31",
          "s0", "s1", "s2", "s3", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17", "s18", "s19",
          "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31",
          /* the slot file */
          "v48", "v49", MPR_V10(5), MPR_V10(6), MPR_V10(7), MPR_V10(8), MPR_V10(9), MPR_V10(10), MPR_V10(11), MPR_V10(12), MPR_V10(13), "v140");
    *res = r;
    return aborted ? INTERP_ABORTED : 0u;
}

template <int DIM, bool VS>
__global__ void __launch_bounds__(64, VS ? 3 : 0)
k_eval_voxels_asm(VoxelArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    unsigned char* const myslot = smem + lane * 4;            /* slot s: myslot + s * 256 */

    /* workgroups go to the eight XCDs round robin, each with an L2 of its own: inside every window of
     * 64 tiles the map hands eight CONSECUTIVE tiles to one XCD, so that siblings (neighbours in the
     * list, which mostly walk one tape) fetch it through one L2 instead of eight; the front-to-back
     * order of the list is kept at that granularity */
    const int b = blockIdx.x;
    const int tile_index = (b & ~63) | ((b & 7) << 3) | ((b >> 3) & 7);
    if (tile_index >= a.count) return;
    const uint64_t* __restrict__ const tro = a.tape_ro;
    const int position = __builtin_amdgcn_readfirstlane(a.tiles[tile_index].position);
    const int tape = __builtin_amdgcn_readfirstlane(a.tiles[tile_index].tape);
    /* the tape's first 64 words and the root head are requested now, so that they travel together with
     * the heightmap read of the skip test instead of after it (three dependent round trips to L2 per
     * tile otherwise; tiles with short tapes are bound by exactly that) */
    const uint64_t first_block = tro[tape + 1 + lane];
    const uint64_t head0 = tro[0];

    constexpr int SUB = (DIM == 3) ? 4 : 8;
    const int S = a.tps * SUB;
    const int4_ pos = unpack(position, a.tps);
    const int4_ sub = unpack(lane, SUB);
    const int px = pos.x * SUB + sub.x;
    const int py = pos.y * SUB + sub.y;
    const int pz = (DIM == 3) ? pos.z * 4 + sub.z : 0;

    bool skip = false;
    if (DIM == 3) {
        /* reference :852-864: the thread owning (pz_low, pz_low + 2) leaves when image >= pz_low + 2 */
        const int pz_low = pos.z * 4 + (sub.z & 1);
        /* read past this CU's vector L1: the heights other tiles of the column have written so far */
        skip = __hip_atomic_load(&a.image[px + py * S], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= pz_low + 2;
        if (ballot(!skip) == 0) return;
    }
    const float size_recip = 1.0f / (float)(unsigned)(a.tps * SUB);
    const float fx = ((px + 0.5f) * size_recip - 0.5f) * 2.0f;
    const float fy = ((py + 0.5f) * size_recip - 0.5f) * 2.0f;
    float vx, vy, vz;
    if (DIM == 3) {
        const float fz = ((pz + 0.5f) * size_recip - 0.5f) * 2.0f;
        const float fw = a.mat[3] * fx + a.mat[7] * fy + a.mat[11] * fz + a.mat[15];
        vx = (a.mat[0] * fx + a.mat[4] * fy + a.mat[8] * fz + a.mat[12]) / fw;
        vy = (a.mat[1] * fx + a.mat[5] * fy + a.mat[9] * fz + a.mat[13]) / fw;
        vz = (a.mat[2] * fx + a.mat[6] * fy + a.mat[10] * fz + a.mat[14]) / fw;
    } else {
        const float fw = a.mat[2] * fx + a.mat[5] * fy + a.mat[8];
        vx = (a.mat[0] * fx + a.mat[3] * fy + a.mat[6]) / fw;
        vy = (a.mat[1] * fx + a.mat[4] * fy + a.mat[7]) / fw;
        vz = a.z;
    }
    const int pz_low2 = (DIM == 3) ? pos.z * 4 + (sub.z & 1) + 2 : 0;
    float res;
    if (VS) {
        if (interp_asm_vgpr(tro, (uint32_t)(tape + 1), lane, &first_block, DIM == 3 ? a.image : nullptr, (uint32_t)(px + py * S) * 4u, pz_low2,
                            (uint32_t)(head0 >> 8) & 0xFFu, (uint32_t)(head0 >> 16) & 0xFFu, (uint32_t)(head0 >> 24) & 0xFFu, vx, vy, vz,
                            &res) == INTERP_ABORTED)
            return;
    } else {
        *reinterpret_cast<float*>(myslot + ((head0 >> 8) & 0xFF) * 256) = vx;
        *reinterpret_cast<float*>(myslot + ((head0 >> 16) & 0xFF) * 256) = vy;
        *reinterpret_cast<float*>(myslot + ((head0 >> 24) & 0xFF) * 256) = vz;
        const uint32_t rslot = interp_asm(tro, (uint32_t)(tape + 1), smem, lane, &first_block, DIM == 3 ? a.image : nullptr,
                                          (uint32_t)(px + py * S) * 4u, pz_low2);
        if (rslot == INTERP_ABORTED) return;
        res = *reinterpret_cast<const float*>(myslot + rslot * 256);
    }
    if (!skip && res < 0.0f) {
        if (DIM == 3) {
            int* p = &a.image[px + py * S];
            if (*p < pz) atomicMax(p, pz);
        } else {
            a.image[px + py * S] = 1;
        }
    }
}

#ifdef MPR_TEST_HOOKS
/* one clause through the assembly interpreter: tape3 = {head (x,y,z in slots 1,2,3), the clause
 * (lhs = slot 1, rhs = slot 2, out = slot 4), end (result slot 4)} */
__global__ void __launch_bounds__(64)
k_test_float_asm(const uint64_t* tape3, int n, const float* a, const float* b, float* out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int i = blockIdx.x * 64 + lane;
    float* const myslot = reinterpret_cast<float*>(smem + lane * 4);
    myslot[1 * 64] = i < n ? a[i] : 0.0f;
    myslot[2 * 64] = (i < n && b) ? b[i] : 0.0f;
    myslot[3 * 64] = 0.0f;
    const uint32_t rslot = interp_asm(tape3, 1u, smem, lane);
    const float r = myslot[rslot * 64];
    if (i < n) out[i] = r;
}
void launch_test_float_asm(hipStream_t s, const uint64_t* tape3, int n, const float* a, const float* b, float* out)
{
    hipLaunchKernelGGL(k_test_float_asm, dim3((n + 63) / 64), dim3(64), 8 * 256, s, tape3, n, a, b, out);
}

/* The square-root routine of the assembly interpreters and of the generated code (MPR_ASM_SQRT_BODY) on EVERY float: bit
 * patterns first .. first + count - 1, 64 consecutive ones per wavefront (the routine picks its fast path per wavefront),
 * against the compiler's correctly rounded sqrtf.  out[0] += mismatches; out[1] = one offending bit pattern. */
__global__ void __launch_bounds__(256)
k_test_sqrt_all(unsigned long long first, unsigned long long count, unsigned long long* out)
{
    const unsigned long long per = 1ull << 16;                 /* values per workgroup */
    unsigned long long bad = 0;
    uint32_t bad_bits = 0;
    for (unsigned long long base = (unsigned long long)blockIdx.x * per; base < count; base += (unsigned long long)gridDim.x * per) {
        for (unsigned long long k = threadIdx.x; k < per && base + k < count; k += blockDim.x) {
            const uint32_t bits = (uint32_t)(first + base + k);
            const float x = mpr_u2f(bits);
            float r;
            asm volatile("v_mov_b32 v35, %1\n s_mov_b32 s90, 0x260\n" MPR_ASM_SQRT_BODY "v_mov_b32 %0, v37\n s_branch L_sqrtover_%=\n" MPR_ASM_SQRT_TAIL "L_sqrtover_%=:\n"
                         : "=v"(r) : "v"(x)
                         : "vcc", "scc", "s90", "s91", "s92", "s93", "s94", "s95", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42");
            const float want = __builtin_sqrtf(x);
            const uint32_t rb = mpr_f2u(r), wb = mpr_f2u(want);
            const bool both_nan = (rb & 0x7fffffffu) > 0x7f800000u && (wb & 0x7fffffffu) > 0x7f800000u;
            if (rb != wb && !both_nan) { ++bad; bad_bits = bits; }
        }
    }
    if (bad) {
        atomicAdd(out, bad);
        out[1] = bad_bits;
    }
}
void launch_test_sqrt_all(hipStream_t s, unsigned long long first, unsigned long long count, unsigned long long* out)
{
    hipLaunchKernelGGL(k_test_sqrt_all, dim3(4096), dim3(256), 0, s, first, count, out);
}

#endif  /* MPR_TEST_HOOKS */
size_t voxel_asm_lds_bytes(int nslots) { return (size_t)nslots * 256; }
void launch_eval_voxels_asm(hipStream_t s, int dim, const VoxelArgs& a)
{
    if (a.count <= 0) return;
    const dim3 g((a.count + 63) & ~63), b(64);           /* whole windows of 64: the tile map permutes inside them */
    const size_t lds = voxel_asm_lds_bytes(a.nslots);
    /* slots in registers when the LDS slot file would hold a CU under the 12 wavefronts those registers allow */
    const bool vs = a.vgpr_slots && a.nslots <= MPR_VS_MAX_SLOTS && lds > (size_t)160 * 1024 / 12;
    if (dim == 3) {
        if (vs) hipLaunchKernelGGL((k_eval_voxels_asm<3, true>), g, b, 16, s, a);
        else hipLaunchKernelGGL((k_eval_voxels_asm<3, false>), g, b, lds, s, a);
    } else {
        if (vs) hipLaunchKernelGGL((k_eval_voxels_asm<2, true>), g, b, 16, s, a);
        else hipLaunchKernelGGL((k_eval_voxels_asm<2, false>), g, b, lds, s, a);
    }
}

}  // namespace mprk
