/*
 * kernels_voxel_jit.hip — the float voxel / pixel pass (reference src/context.cu:707-964,
 * calculate_voxels / calculate_pixels + eval_voxels_f) with every tape turned into gfx950 machine
 * code on the device and run as straight-line code: one clause = one VALU instruction.
 *
 * Why.  An interpreter pays for every clause, however cheap, a dispatch (read the clause word, form
 * the handler's address, jump: 5 scalar + 1 vector instruction in the threaded-code interpreter of
 * kernels_voxel_asm.hip) and the operands' addresses, because slots are runtime values; measured,
 * profiles/r02a_sq_bear1024_baseline.txt: 7.9 VALU + 5.6 SALU wave-instructions per clause visit, of
 * which the model's own arithmetic is 4.3, and the one scalar unit of a CU (0.96 instr/clk/CU,
 * profiles/r02a_issue_rates.txt) is busy 63 % of the time with dispatch alone.  The reference's answer to
 * "why not compile the expression" is compile time (seconds per edit); a clause-to-instruction
 * TEMPLATE translation costs microseconds and needs no compiler:
 *
 *   - slot s of the tape IS vector register v[48 + s] for the duration of a tile (tapes with up to
 *     192 slots; the kernel is built for 24 / 40 / 96 / 192 slots so that small slot files keep a high
 *     occupancy; larger slot files go through the LDS interpreter of kernels_voxel_asm.hip).  `ADD_LHS_RHS out, lhs, rhs` becomes the 4-byte instruction `v_add_f32 v[48+out],
 *     v[48+lhs], v[48+rhs]`; immediates become literals; min / max keep the canonicalising v_max pair;
 *     division, square root, exp, log, sin, cos, asin, acos, atan become `v_mov` + `s_swappc_b64` to the
 *     routines the interpreters use (asm_float_bodies.hpp; asin / acos / atan compiled) — the same
 *     instructions in the same order, hence the same bits as k_eval_voxels_asm, k_eval_voxels and the oracle;
 *   - a wavefront translates a tape 64 clauses at a time: lane j decodes clause j into up to 5 dwords
 *     through a 32-entry template table in LDS, a ballot prefix sum places them, five stores write them
 *     to the wavefront's own region of an EXECUTABLE buffer (context.hip: HSA executable pool), JUMP /
 *     chunk links disappear;
 *   - then `s_waitcnt vmcnt(0)`, `s_icache_inv` (the region is rewritten tile after tile: stale lines
 *     of the previous tape would otherwise execute — scripts/ubench/jit_probe.hip shows exactly that
 *     without the invalidate, and zero stale rounds with it), `s_swappc_b64` into the region.  Code is
 *     only ever executed by the wavefront that wrote it, after its own invalidate, so no other
 *     instruction cache can hold a stale copy that matters;
 *   - wavefronts are persistent (grid = what the chip holds) and walk the tile list in runs of eight
 *     neighbours; a tile whose tape is the one just translated (siblings share tapes: 54 % of bear's
 *     tiles) reuses the code and skips the invalidate.
 *
 * The region of a wavefront holds the longest code any tape of the frame can have: shortening only
 * replaces min / max by copies and drops clauses, so the root tape's code length is the bound (host:
 * jit_code_dwords).
 */
#include "asm_float_bodies.hpp"
#include "kernel_common.hpp"

namespace mprk {

constexpr int JIT_SLOT_BASE = 48;        /* slot s = v[48 + s]; v32..v34 inputs, v35..v47 routine registers */
constexpr int JIT_RUN = 8;               /* consecutive tiles a wavefront takes at a time */

/* asin / acos / atan: compiled leaf routines (argument and result in v0, return s[30:31]) under fixed
 * names; they use v0..v7, s0..s15 and vcc at most (tests/test_callee_registers.py), which is all the
 * generated code's asm statement declares clobbered below v32 / s30: the kernel's own values live there */
__device__ __attribute__((noinline, used)) float jit_asin(float v) __asm__("mpr_fj_asin");
__device__ __attribute__((noinline, used)) float jit_acos(float v) __asm__("mpr_fj_acos");
__device__ __attribute__((noinline, used)) float jit_atan(float v) __asm__("mpr_fj_atan");
__device__ float jit_asin(float v) { return mpr_asinf(v); }
__device__ float jit_acos(float v) { return mpr_acosf(v); }
__device__ float jit_atan(float v) { return mpr_atanf(v); }

/* ---- the template table -------------------------------------------------------------------------- */
/* per opcode: number of dwords, and per dword a base word + a program byte:
 *   bits 1:0 register put into vdst  (bits 24:17)   0 none, 1 out, 2 lhs, 3 rhs
 *   bits 3:2 register put into vsrc1 (bits 16:9)
 *   bits 5:4 register put into src0  (bits 8:0; the base word carries the VGPR bit 0x100)
 *   bits 7:6 0 instruction, 1 the clause's immediate, 2 the immediate canonicalised (v_max_f32 x, x) */
struct JitTemplate {
    uint32_t base[32][5];
    uint32_t prog[32];       /* program bytes of dwords 0..3 */
    uint8_t prog4[32];       /* program byte of dword 4 */
    uint8_t n[32];
};
namespace jt {
constexpr uint32_t O = 1, A = 2, R = 3;
constexpr uint32_t P(uint32_t d, uint32_t s1, uint32_t s0, uint32_t kind = 0) { return d | (s1 << 2) | (s0 << 4) | (kind << 6); }
constexpr uint32_t VOP2(uint32_t op, uint32_t vdst, uint32_t vsrc1, uint32_t src0) { return (op << 25) | (vdst << 17) | (vsrc1 << 9) | src0; }
constexpr uint32_t VREG = 0x100;                                   /* src0 names a VGPR */
constexpr uint32_t LITERAL = 255;
constexpr uint32_t MOV(uint32_t vdst, uint32_t src0) { return 0x7E000200u | (vdst << 17) | src0; }
constexpr uint32_t CALL(uint32_t sgpr) { return 0xBE9E1E00u | sgpr; }      /* s_swappc_b64 s[30:31], s[sgpr:sgpr+1] */
constexpr uint32_t V_ADD = 1, V_SUB = 2, V_SUBREV = 3, V_MUL = 5, V_MIN = 10, V_MAX = 11, V_AND = 19, V_XOR = 21;
/* SGPRs the generated code refers to (set up by jit_run) */
constexpr uint32_t S_DIV = 52, S_SQRT = 54, S_EXP = 56, S_LOG = 58, S_SIN = 60, S_COS = 62, S_ASIN = 64, S_ACOS = 66, S_ATAN = 68;
constexpr uint32_t S_SIGN = 70, S_ABS = 71;
struct Row { uint32_t n; uint32_t w[5]; uint32_t p[5]; };
constexpr Row none() { return Row{0, {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}}; }
constexpr Row one(uint32_t w, uint32_t p) { return Row{1, {w, 0, 0, 0, 0}, {p, 0, 0, 0, 0}}; }
constexpr Row with_imm(uint32_t w, uint32_t p) { return Row{2, {w, 0, 0, 0, 0}, {p, P(0, 0, 0, 1), 0, 0, 0}}; }
constexpr Row unary_call(uint32_t s) { return Row{3, {MOV(35, VREG), CALL(s), MOV(0, VREG + 37), 0, 0}, {P(0, 0, A), 0, P(O, 0, 0), 0, 0}}; }
constexpr Row leaf_call(uint32_t s) { return Row{3, {MOV(0, VREG), CALL(s), MOV(0, VREG + 0), 0, 0}, {P(0, 0, A), 0, P(O, 0, 0), 0, 0}}; }
constexpr Row minmax_rr(uint32_t op)
{
    return Row{3, {VOP2(V_MAX, 35, 0, VREG), VOP2(V_MAX, 36, 0, VREG), VOP2(op, 0, 36, VREG + 35), 0, 0},
               {P(0, A, A), P(0, R, R), P(O, 0, 0), 0, 0}};
}
constexpr Row minmax_ri(uint32_t op)
{
    return Row{4, {VOP2(V_MAX, 35, 0, VREG), MOV(36, LITERAL), 0, VOP2(op, 0, 36, VREG + 35), 0},
               {P(0, A, A), 0, P(0, 0, 0, 2), P(O, 0, 0), 0}};
}
constexpr Row row_of(uint32_t op)
{
    switch (op) {
        case MPR_OP_SQUARE_LHS: return one(VOP2(V_MUL, 0, 0, VREG), P(O, A, A));
        case MPR_OP_SQRT_LHS: return unary_call(S_SQRT);
        case MPR_OP_NEG_LHS: return one(VOP2(V_XOR, 0, 0, S_SIGN), P(O, A, 0));
        case MPR_OP_SIN_LHS: return unary_call(S_SIN);
        case MPR_OP_COS_LHS: return unary_call(S_COS);
        case MPR_OP_ASIN_LHS: return leaf_call(S_ASIN);
        case MPR_OP_ACOS_LHS: return leaf_call(S_ACOS);
        case MPR_OP_ATAN_LHS: return leaf_call(S_ATAN);
        case MPR_OP_EXP_LHS: return unary_call(S_EXP);
        case MPR_OP_ABS_LHS: return one(VOP2(V_AND, 0, 0, S_ABS), P(O, A, 0));
        case MPR_OP_LOG_LHS: return unary_call(S_LOG);
        case MPR_OP_ADD_LHS_IMM: return with_imm(VOP2(V_ADD, 0, 0, LITERAL), P(O, A, 0));
        case MPR_OP_ADD_LHS_RHS: return one(VOP2(V_ADD, 0, 0, VREG), P(O, R, A));
        case MPR_OP_MUL_LHS_IMM: return with_imm(VOP2(V_MUL, 0, 0, LITERAL), P(O, A, 0));
        case MPR_OP_MUL_LHS_RHS: return one(VOP2(V_MUL, 0, 0, VREG), P(O, R, A));
        case MPR_OP_MIN_LHS_IMM: return minmax_ri(V_MIN);
        case MPR_OP_MIN_LHS_RHS: return minmax_rr(V_MIN);
        case MPR_OP_MAX_LHS_IMM: return minmax_ri(V_MAX);
        case MPR_OP_MAX_LHS_RHS: return minmax_rr(V_MAX);
        case MPR_OP_SUB_LHS_IMM: return with_imm(VOP2(V_SUBREV, 0, 0, LITERAL), P(O, A, 0));      /* lhs - imm */
        case MPR_OP_SUB_IMM_RHS: return with_imm(VOP2(V_SUB, 0, 0, LITERAL), P(O, R, 0));         /* imm - rhs */
        case MPR_OP_SUB_LHS_RHS: return one(VOP2(V_SUB, 0, 0, VREG), P(O, R, A));
        case MPR_OP_DIV_LHS_IMM:
            return Row{5, {MOV(35, VREG), MOV(36, LITERAL), 0, CALL(S_DIV), MOV(0, VREG + 37)}, {P(0, 0, A), 0, P(0, 0, 0, 1), 0, P(O, 0, 0)}};
        case MPR_OP_DIV_IMM_RHS:
            return Row{5, {MOV(35, LITERAL), 0, MOV(36, VREG), CALL(S_DIV), MOV(0, VREG + 37)}, {0, P(0, 0, 0, 1), P(0, 0, R), 0, P(O, 0, 0)}};
        case MPR_OP_DIV_LHS_RHS:
            return Row{4, {MOV(35, VREG), MOV(36, VREG), CALL(S_DIV), MOV(0, VREG + 37), 0}, {P(0, 0, A), P(0, 0, R), 0, P(O, 0, 0), 0}};
        case MPR_OP_COPY_IMM: return with_imm(MOV(0, LITERAL), P(O, 0, 0));
        case MPR_OP_COPY_LHS: return one(MOV(0, VREG), P(O, 0, A));
        case MPR_OP_COPY_RHS: return one(MOV(0, VREG), P(O, 0, R));
        default: return none();                                   /* end, JUMP, not an opcode: no code */
    }
}
constexpr JitTemplate make_table()
{
    JitTemplate t{};
    for (uint32_t op = 0; op < 32; ++op) {
        const Row r = row_of(op);
        t.n[op] = (uint8_t)r.n;
        for (int k = 0; k < 5; ++k) t.base[op][k] = r.w[k];
        t.prog[op] = r.p[0] | (r.p[1] << 8) | (r.p[2] << 16) | (r.p[3] << 24);
        t.prog4[op] = (uint8_t)r.p[4];
    }
    return t;
}
}  // namespace jt
static const JitTemplate h_jit_table = jt::make_table();
__constant__ JitTemplate d_jit_table = jt::make_table();

/* dwords of the code of a tape (host; the bound for every tape shortened from it) */
size_t jit_code_dwords(const uint64_t* clauses, int n)
{
    size_t d = 3 + 2;                         /* prologue: three axis moves; epilogue: result move, return */
    for (int i = 0; i < n; ++i) d += h_jit_table.n[mpr_cl_op(clauses[i]) & 31];
    return d;
}

/* LDS copy of the table, one 64-byte row per opcode, in the form the translator's instructions want:
 *   [0..4]   base word k
 *   [5..9]   v_perm_b32 selector k: the registers of the clause (bytes 1..3 of `regs` = 48 + out / lhs / rhs,
 *            byte 0 = 0) gathered as {src0, vsrc1, vdst, 0}, or all four bytes of the literal
 *   [10..14] mask k: 0xFFFFFF00 for an instruction (word = T + (T & mask) + base moves vsrc1 to bit 9 and vdst
 *            to bit 17), 0 for a literal (word = the literal)
 *   [15]     number of dwords | 0x100 if the literal is the canonicalised immediate */
constexpr int JIT_ROW = 16;
DEV void jit_load_table(uint32_t* lds, int lane)
{
    for (int i = lane; i < 32 * JIT_ROW; i += 64) {
        const int op = i / JIT_ROW, k = i % JIT_ROW;
        const uint64_t progs = (uint64_t)d_jit_table.prog[op] | ((uint64_t)d_jit_table.prog4[op] << 32);
        uint32_t v = 0;
        if (k < 15) {
            const int w = k % 5;
            const uint32_t p = (uint32_t)(progs >> (8 * w)) & 0xFFu;
            const bool literal = (p >> 6) != 0;
            if (k < 5) v = literal ? 0u : d_jit_table.base[op][w];
            else if (k < 10) v = literal ? 0x07060504u : (((p >> 4) & 3u) | (((p >> 2) & 3u) << 8) | ((p & 3u) << 16) | 0x0C000000u);
            else v = literal ? 0u : 0xFFFFFF00u;
        } else {
            bool canon = false;
            for (int w = 0; w < 5; ++w) canon |= (((uint32_t)(progs >> (8 * w)) >> 6) & 3u) == 2u;
            v = (uint32_t)d_jit_table.n[op] | (canon ? 0x100u : 0u);
        }
        lds[i] = v;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
}

/* Translate the tape whose first clause is tro[first] into `code`; returns the number of dwords.
 * head0 = the root tape's head clause (axis slots), the same for every tape of the frame.
 * trash_off: byte offset (from `code`) of this lane's slot in a part of the region that is never executed.
 *
 * Written in assembly because two things decide its cost.  (1) Latency: a tape is a linked list of 64-word
 * chunks, so the next block's address is known only once the current block has arrived; the next block is
 * requested the moment the current one is in registers, before any of the work on it, and no memory
 * operation of that work is conditional — every lane stores five dwords per block, the ones its clause does
 * not have go to the dump — so the wait for the next block is the exact `s_waitcnt vmcnt(5)`.  (2) Instruction
 * count: the registers of a clause are placed into a template word with one v_perm_b32 (gather the bytes),
 * one v_and_b32 and one v_add3_u32 (move vsrc1 / vdst to their odd bit positions and add the base word);
 * ~60 vector instructions translate 63 clauses.
 *   s42 block base (clause index)   s43 dwords emitted   s46 first terminator lane (64: none)
 *   s52 / s53 terminator lo / hi   s54 its opcode (0 end, 1 jump)   s55 next block base
 *   v[34:35] block (lane j = clause j)   v[36:37] next block   v40..v44 the five words   v59 dwords of the lane
 * (all of them registers the generated code owns while it runs and nobody needs while it is being written) */
DEV uint32_t jit_translate(const uint64_t* __restrict__ tro, uint32_t first, uint64_t head0, uint32_t* __restrict__ code,
                           uint32_t trash_off, uint32_t* __restrict__ tbl, int lane, uint32_t dbg = 0)
{
    /* LDS behind the table: 336 dwords in which a block's dwords are put side by side (lane after lane) before
     * they leave in 16-byte pieces, and 8 dwords nobody reads */
    uint32_t* const stage = tbl + 32 * JIT_ROW;
    if (lane < 3) {
        const uint32_t slot = (uint32_t)(head0 >> (8 * (lane + 1))) & 0xFFu;
        stage[lane] = jt::MOV(JIT_SLOT_BASE + slot, jt::VREG + 32 + lane);       /* v[48 + axis slot] = v32 / v33 / v34 */
    }
    const uint32_t tlo = rdfirst((uint32_t)(uintptr_t)tro), thi = rdfirst((uint32_t)((uintptr_t)tro >> 32));
    const uint64_t cbase = rfl64((uint64_t)(uintptr_t)code);
    const uint32_t ltab = rdfirst((uint32_t)(uintptr_t)tbl);
    const uint32_t lstage = rdfirst((uint32_t)(uintptr_t)stage), ldump = lstage + 336u * 4u;
    const uint32_t lane8 = (uint32_t)lane * 8u, lane16 = (uint32_t)lane * 16u;
    const uint32_t l3 = lstage + ((uint32_t)lane & 3u) * 4u;
    first = rdfirst(first);
    uint32_t used;
    asm volatile(
        "s_mov_b32 s42, %[first]\n"
        "s_mov_b32 s43, 0\n"                               /* dwords that have left for memory */
        "s_mov_b32 s44, 3\n"                               /* dwords waiting at the start of the staging buffer */
        "s_mov_b32 s50, s42\n s_mov_b32 s51, 0\n s_lshl_b64 s[50:51], s[50:51], 3\n"
        "s_add_u32 s50, s50, %[tlo]\n s_addc_u32 s51, s51, %[thi]\n"
        "global_load_dwordx2 v[34:35], %[lane8], s[50:51]\n"
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
        "s_branch L_body_%=\n"
        "L_next_%=:\n"
        "s_waitcnt vmcnt(2)\n"                             /* the prefetched block; the two stores behind it stay in flight */
        "v_mov_b32 v34, v36\n v_mov_b32 v35, v37\n"
        "L_body_%=:\n"
        "v_and_b32 v38, 0xff, v34\n"                       /* opcode */
        "v_cmp_gt_u32 vcc, 2, v38\n"                       /* end of tape or JUMP */
        "s_ff1_i32_b64 s46, vcc\n"
        "s_cmp_lg_u32 s46, -1\n"
        "s_cbranch_scc1 L_term_%=\n"
        "s_mov_b32 s46, 64\n s_mov_b32 s54, 1\n s_add_u32 s55, s42, 64\n"
        "s_branch L_pf_%=\n"
        "L_term_%=:\n"
        "v_readlane_b32 s52, v34, s46\n v_readlane_b32 s53, v35, s46\n"
        "s_and_b32 s54, s52, 0xff\n"
        "s_add_u32 s55, s42, s46\n s_add_u32 s55, s55, s53\n s_add_u32 s55, s55, 1\n"      /* JUMP: relative, then pre-increment */
        "L_pf_%=:\n"
        "s_cmp_eq_u32 s54, 0\n"
        "s_cbranch_scc1 L_emit_%=\n"
        "s_mov_b32 s50, s55\n s_mov_b32 s51, 0\n s_lshl_b64 s[50:51], s[50:51], 3\n"
        "s_add_u32 s50, s50, %[tlo]\n s_addc_u32 s51, s51, %[thi]\n"
        "global_load_dwordx2 v[36:37], %[lane8], s[50:51]\n"
        "L_emit_%=:\n"
        "v_min_u32 v38, 31, v38\n"                         /* anything that is not an opcode: the empty row 31 */
        "v_lshlrev_b32 v39, 6, v38\n"
        "v_add_u32 v39, %[ltab], v39\n"
        "ds_read_b128 v[40:43], v39\n"                     /* base 0..3 */
        "ds_read_b128 v[44:47], v39 offset:16\n"           /* base 4, selector 0..2 */
        "ds_read_b128 v[48:51], v39 offset:32\n"           /* selector 3, 4, mask 0, 1 */
        "ds_read_b128 v[52:55], v39 offset:48\n"           /* mask 2..4, meta */
        "v_and_b32 v56, 0xffffff00, v34\n"
        "v_add_u32 v56, 0x30303000, v56\n"                 /* registers: 48 + out / lhs / rhs in bytes 1..3 */
        "v_max_f32 v57, v35, v35\n"                        /* the immediate as the min / max handlers see it */
        "s_waitcnt lgkmcnt(0)\n"
        "v_and_b32 v58, 0x100, v55\n"
        "v_cmp_ne_u32 vcc, 0, v58\n"
        "v_and_b32 v59, 7, v55\n"
        "s_nop 0\n"
        "v_cndmask_b32 v57, v35, v57, vcc\n"               /* the literal of this clause */
        "v_cmp_gt_u32 vcc, s46, %[lane]\n"                 /* clauses in front of the terminator */
        "s_nop 1\n"
        "v_cndmask_b32 v59, 0, v59, vcc\n"                 /* dwords of this lane */
        "v_perm_b32 v45, v57, v56, v45\n v_and_b32 v50, v50, v45\n v_add3_u32 v40, v45, v50, v40\n"
        "v_perm_b32 v46, v57, v56, v46\n v_and_b32 v51, v51, v46\n v_add3_u32 v41, v46, v51, v41\n"
        "v_perm_b32 v47, v57, v56, v47\n v_and_b32 v52, v52, v47\n v_add3_u32 v42, v47, v52, v42\n"
        "v_perm_b32 v48, v57, v56, v48\n v_and_b32 v53, v53, v48\n v_add3_u32 v43, v48, v53, v43\n"
        "v_perm_b32 v49, v57, v56, v49\n v_and_b32 v54, v54, v49\n v_add3_u32 v44, v49, v54, v44\n"
        /* exclusive prefix sum of the dword counts (0..5) over the lanes, bit by bit */
        "v_and_b32 v45, 1, v59\n v_cmp_ne_u32 s[56:57], 0, v45\n"
        "v_and_b32 v45, 2, v59\n v_cmp_ne_u32 s[58:59], 0, v45\n"
        "v_and_b32 v45, 4, v59\n v_cmp_ne_u32 s[60:61], 0, v45\n"
        "v_cmp_eq_u32 vcc, 0, v59\n"                       /* lanes without a dword */
        "v_mbcnt_lo_u32_b32 v46, s56, 0\n v_mbcnt_hi_u32_b32 v46, s57, v46\n"
        "v_mbcnt_lo_u32_b32 v47, s58, 0\n v_mbcnt_hi_u32_b32 v47, s59, v47\n"
        "v_lshl_add_u32 v46, v47, 1, v46\n"
        "v_mbcnt_lo_u32_b32 v47, s60, 0\n v_mbcnt_hi_u32_b32 v47, s61, v47\n"
        "v_lshl_add_u32 v46, v47, 2, v46\n"
        "v_add_u32 v46, s44, v46\n"
        "v_lshl_add_u32 v46, v46, 2, %[stage]\n"           /* where this lane's first dword goes in the staging buffer */
        "v_mov_b32 v47, %[dump]\n"
        "v_cndmask_b32 v46, v46, v47, vcc\n"
        "s_bcnt1_i32_b64 s56, s[56:57]\n s_bcnt1_i32_b64 s58, s[58:59]\n s_bcnt1_i32_b64 s60, s[60:61]\n"
        "s_lshl_b32 s58, s58, 1\n s_lshl_b32 s60, s60, 2\n"
        "s_add_u32 s44, s44, s56\n s_add_u32 s44, s44, s58\n s_add_u32 s44, s44, s60\n"      /* dwords now in the buffer */
        /* All five, last one first: a dword a lane does not have lands on a later lane's place, and that lane's
         * own dword for the place — always one with a smaller number — is written after it (LDS operations of a
         * wavefront happen in order).  Lanes without any write to the dump, so that within one instruction no two
         * lanes share an address. */
        "ds_write_b32 v46, v44 offset:16\n"
        "ds_write_b32 v46, v43 offset:12\n"
        "ds_write_b32 v46, v42 offset:8\n"
        "ds_write_b32 v46, v41 offset:4\n"
        "ds_write_b32 v46, v40\n"
        /* what is complete leaves in 16-byte pieces: lane i carries dwords 4 i .. 4 i + 3 (and 256 + 4 i ...) */
        "s_and_b32 s45, s44, -4\n"                         /* dwords that leave now */
        "v_add_u32 v47, %[stage], %[lane16]\n"
        "s_lshl_b32 s47, s43, 2\n"
        "v_add_u32 v58, s47, %[lane16]\n"                  /* byte offset in the code region */
        "v_lshrrev_b32 v45, 2, %[lane16]\n"                /* 4 i */
        "v_cmp_gt_u32 vcc, s45, v45\n"
        "v_add_u32 v45, 0x100, v45\n"
        "v_cmp_gt_u32 s[56:57], s45, v45\n"
        "s_waitcnt lgkmcnt(0)\n"
        "ds_read_b128 v[48:51], v47\n"
        "ds_read_b128 v[52:55], v47 offset:1024\n"
        "v_cndmask_b32 v60, %[trash], v58, vcc\n"
        "v_add_u32 v58, 0x400, v58\n"
        "v_cndmask_b32 v61, %[trash], v58, s[56:57]\n"
        /* the 0..3 dwords that stay move to the front */
        "s_lshl_b32 s47, s45, 2\n"
        "v_add_u32 v45, s47, %[l3]\n"
        "ds_read_b32 v45, v45\n"
        "s_waitcnt lgkmcnt(2)\n"
        "global_store_dwordx4 v60, v[48:51], %[code]\n"
        "s_waitcnt lgkmcnt(1)\n"
        "global_store_dwordx4 v61, v[52:55], %[code]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "ds_write_b32 %[l3], v45\n"
        "s_add_u32 s43, s43, s45\n"
        "s_sub_u32 s44, s44, s45\n"
        "s_cmp_eq_u32 s54, 0\n"
        "s_cbranch_scc1 L_done_%=\n"
        "s_mov_b32 s42, s55\n"
        "s_branch L_next_%=\n"
        "L_done_%=:\n"
        /* end clause (byte 1 = result slot): v37 = v[48 + slot], return; with the 0..3 dwords still waiting */
        "s_bfe_u32 s56, s52, 0x80008\n"
        "s_add_u32 s56, s56, 0x7e4a0330\n"                 /* v_mov_b32 v37, v[48 + slot] */
        "s_lshl_b32 s57, s44, 2\n"
        "s_add_u32 s57, s57, %[stage]\n"
        "v_mov_b32 v40, s56\n"
        "v_mov_b32 v41, 0xbe801d48\n"                      /* s_setpc_b64 s[72:73] */
        "v_mov_b32 v46, s57\n"
        "s_waitcnt lgkmcnt(0)\n"
        "ds_write_b32 v46, v40\n"
        "ds_write_b32 v46, v41 offset:4\n"
        "s_add_u32 s44, s44, 2\n"                          /* 2..5 dwords: one per lane */
        "v_lshrrev_b32 v45, 2, %[lane16]\n"                /* 4 i */
        "v_add_u32 v47, %[stage], v45\n"
        "s_lshl_b32 s47, s43, 2\n"
        "v_add_u32 v58, s47, v45\n"
        "s_lshl_b32 s45, s44, 2\n"
        "v_cmp_gt_u32 vcc, s45, v45\n"
        "s_waitcnt lgkmcnt(0)\n"
        "ds_read_b32 v48, v47\n"
        "v_cndmask_b32 v60, %[trash], v58, vcc\n"
        "s_waitcnt lgkmcnt(0)\n"
        "global_store_dword v60, v48, %[code]\n"
        "s_add_u32 %[used], s43, s44\n"
        : [used] "=&s"(used)
        : [first] "s"(first), [tlo] "s"(tlo), [thi] "s"(thi), [code] "s"(cbase), [ltab] "s"(ltab), [stage] "s"(lstage), [dump] "s"(ldump),
          [lane] "v"((uint32_t)lane), [lane8] "v"(lane8), [lane16] "v"(lane16), [l3] "v"(l3), [trash] "v"(trash_off), [dbg] "s"(dbg)
        : "memory", "vcc", "scc", "s42", "s43", "s44", "s45", "s46", "s47", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59",
          "s60", "s61", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50",
          "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61");
    return used;
}

/* Ask for the tape of a tile to be brought near (into the L2) long before it is translated: a tape is a linked
 * list of 64-word chunks and following it costs one trip to memory per chunk — about 2 us each from HBM, six in a
 * row for bear — unless the chunks are already in the cache.  Where they are can be said without reading them:
 * the tile stages write a tape from its end, chunk after chunk at rising addresses (kernels.hip: out_index +=
 * MPR_SUBTAPE_CHUNK), so reading forward from the head means falling addresses, chunk starts are congruent to the
 * root tape's length modulo 64, and the root tape itself is contiguous.  Three LDS-DMA loads of 1 KB (no
 * destination registers; the bytes land in a part of LDS nobody reads) cover six chunks; a wrong guess costs
 * nothing but those loads. */
DEV void jit_prefetch_tape(const uint64_t* __restrict__ tro, int tape, int tape_len, uint32_t lds_dump, int lane)
{
    int w0;                                            /* first word of the highest 1 KB piece */
    int step;
    if (tape < tape_len) {
        w0 = tape;
        step = 128;
    } else {
        const int cs = tape - ((tape - tape_len) & 63);
        w0 = cs - 64;
        step = -128;
    }
    const uint32_t lane16 = (uint32_t)lane * 16u;
    const uint32_t m0v = rdfirst(lds_dump);
#pragma unroll
    for (int p = 0; p < 3; ++p) {                      /* always three: the code that follows counts them (s_waitcnt vmcnt(3)) */
        int w = w0 + p * step;
        if (w < 0) w = 0;
        const uint64_t addr = rfl64((uint64_t)(uintptr_t)(tro + w));
        asm volatile("s_mov_b32 m0, %2\n s_nop 0\n global_load_lds_dwordx4 %0, %1\n" : : "v"(lane16), "s"(addr), "s"(m0v) : "memory", "m0");
    }
}

/* Run the code at `code` on (vx, vy, vz); `fresh`: the region was just rewritten.  NS: slots the
 * kernel provides registers for.  The register lists are what the generated code, the routines and the
 * compiled leaf routines may touch. */
#define JIT_V10(a) "v" #a "0", "v" #a "1", "v" #a "2", "v" #a "3", "v" #a "4", "v" #a "5", "v" #a "6", "v" #a "7", "v" #a "8", "v" #a "9"
#define JIT_CLOBBER_BASE                                                                                                   \
    "memory", "vcc", "scc", "s0", "s1", "s2", "s3", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13", "s14", "s15", "s30", "s31", "s40", "s41", "s42", \
        "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60",      \
        "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s90", "s91", "s92",      \
        "s93", "s94", "s95", "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39",       \
        JIT_V10(4), JIT_V10(5), JIT_V10(6), "v70", "v71"
#define JIT_CLOBBER_40 "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", JIT_V10(8)
#define JIT_CLOBBER_96 JIT_CLOBBER_40, JIT_V10(9), JIT_V10(10), JIT_V10(11), JIT_V10(12), JIT_V10(13), "v140", "v141", "v142", "v143"
#define JIT_CLOBBER_192                                                                                                    \
    JIT_CLOBBER_96, "v144", "v145", "v146", "v147", "v148", "v149", JIT_V10(15), JIT_V10(16), JIT_V10(17), JIT_V10(18), JIT_V10(19), JIT_V10(20), \
        JIT_V10(21), JIT_V10(22), JIT_V10(23)

#define JIT_ROUTINE_ADDR(reg_lo, reg_hi, label)          \
    "s_add_u32 s" #reg_lo ", s50, " label "_%=-L_pc_%=\n" \
    "s_addc_u32 s" #reg_hi ", s51, 0\n"
#define JIT_LEAF_ADDR(reg_lo, reg_hi, sym)                             \
    "s_getpc_b64 s[" #reg_lo ":" #reg_hi "]\n"                           \
    "s_add_u32 s" #reg_lo ", s" #reg_lo ", " sym "@rel32@lo+4\n"        \
    "s_addc_u32 s" #reg_hi ", s" #reg_hi ", " sym "@rel32@hi+12\n"
#define JIT_ASM_TEXT                                                                                   \
    "s_bitcmp1_b32 %[fresh], 1\n"                     /* bit 1: three prefetch loads were issued after the code's stores */ \
    "s_cbranch_scc1 L_w3_%=\n"                                                                         \
    "s_waitcnt vmcnt(0)\n"                                                                             \
    "L_w3_%=:\n"                                                                                       \
    "s_waitcnt vmcnt(3) lgkmcnt(0)\n"                                                                  \
    "s_bitcmp1_b32 %[fresh], 0\n"                     /* bit 0: the region was just rewritten */       \
    "s_cbranch_scc0 L_cached_%=\n"                                                                     \
    "s_icache_inv\n"                                                                                   \
    "s_nop 7\n s_nop 7\n"                                                                              \
    "L_cached_%=:\n"                                                                                   \
    "s_getpc_b64 s[50:51]\n"                                                                           \
    "L_pc_%=:\n"                                                                                       \
    JIT_ROUTINE_ADDR(52, 53, "L_div") JIT_ROUTINE_ADDR(54, 55, "L_sqrt") JIT_ROUTINE_ADDR(56, 57, "L_exp")  \
    JIT_ROUTINE_ADDR(58, 59, "L_log") JIT_ROUTINE_ADDR(60, 61, "L_sin") JIT_ROUTINE_ADDR(62, 63, "L_cos")   \
    JIT_LEAF_ADDR(64, 65, "mpr_fj_asin") JIT_LEAF_ADDR(66, 67, "mpr_fj_acos") JIT_LEAF_ADDR(68, 69, "mpr_fj_atan") \
    "s_mov_b32 s70, 0x80000000\n"                                                                      \
    "s_mov_b32 s71, 0x7fffffff\n"                                                                      \
    "s_mov_b32 s90, 0x260\n"                            /* class mask of the square root */           \
    "v_mov_b32 v32, %[vx]\n v_mov_b32 v33, %[vy]\n v_mov_b32 v34, %[vz]\n"                             \
    "s_mov_b32 s74, %[clo]\n s_mov_b32 s75, %[chi]\n"                                                  \
    "s_swappc_b64 s[72:73], s[74:75]\n"                                                                \
    "v_mov_b32 %[res], v37\n"                                                                          \
    "s_branch L_end_%=\n"                                                                              \
    "L_div_%=:\n" MPR_ASM_DIV_BODY "s_setpc_b64 s[30:31]\n"                                            \
    "L_sqrt_%=:\n" MPR_ASM_SQRT_BODY "s_setpc_b64 s[30:31]\n"                                          \
    "L_exp_%=:\n" MPR_ASM_EXP_BODY "s_setpc_b64 s[30:31]\n"                                            \
    "L_log_%=:\n" MPR_ASM_LOG_BODY "s_setpc_b64 s[30:31]\n"                                            \
    "L_sin_%=:\n" MPR_ASM_SINCOS_BODY "s_setpc_b64 s[30:31]\n"                                         \
    "L_cos_%=:\n" MPR_ASM_SINCOS_BODY "v_mov_b32 v37, v36\n s_setpc_b64 s[30:31]\n"                    \
    "L_end_%=:\n"

template <int NS>
DEV float jit_run(const uint32_t* code, uint32_t fresh, float vx, float vy, float vz)
{
    const uint32_t clo = rdfirst((uint32_t)(uintptr_t)code), chi = rdfirst((uint32_t)((uintptr_t)code >> 32));
    fresh = rdfirst(fresh);
    float res;
    if constexpr (NS <= 24) {
        asm volatile(JIT_ASM_TEXT : [res] "=&v"(res) : [vx] "v"(vx), [vy] "v"(vy), [vz] "v"(vz), [clo] "s"(clo), [chi] "s"(chi), [fresh] "s"(fresh)
                     : JIT_CLOBBER_BASE);
    } else if constexpr (NS <= 40) {
        asm volatile(JIT_ASM_TEXT : [res] "=&v"(res) : [vx] "v"(vx), [vy] "v"(vy), [vz] "v"(vz), [clo] "s"(clo), [chi] "s"(chi), [fresh] "s"(fresh)
                     : JIT_CLOBBER_BASE, JIT_CLOBBER_40);
    } else if constexpr (NS <= 96) {
        asm volatile(JIT_ASM_TEXT : [res] "=&v"(res) : [vx] "v"(vx), [vy] "v"(vy), [vz] "v"(vz), [clo] "s"(clo), [chi] "s"(chi), [fresh] "s"(fresh)
                     : JIT_CLOBBER_BASE, JIT_CLOBBER_96);
    } else {
        asm volatile(JIT_ASM_TEXT : [res] "=&v"(res) : [vx] "v"(vx), [vy] "v"(vy), [vz] "v"(vz), [clo] "s"(clo), [chi] "s"(chi), [fresh] "s"(fresh)
                     : JIT_CLOBBER_BASE, JIT_CLOBBER_192);
    }
    return res;
}

struct JitVoxelArgs {
    VoxelArgs v;
    uint32_t* code;            /* executable; one region per wavefront of the grid */
    uint32_t region_dwords;    /* per wavefront: `slots` pieces of slot_dwords for code, then 320 dwords the translator dumps into */
    uint32_t slot_dwords;
    uint32_t slots;
    int debug;                 /* development (MPR_JIT_DEBUG): 1 = translate only, 2 = translate each wavefront's first tape only, 8 = no prefetch */
    int tape_len;              /* words of the root tape at pool[0] */
    unsigned long long* dbg;   /* development (MPR_JIT_DEBUG & 16): cycles in translation / generated code / all, counts */
};

template <int DIM, int NS>
__global__ void __launch_bounds__(64)
k_eval_voxels_jit(JitVoxelArgs j)
{
    __shared__ __attribute__((aligned(64))) uint32_t tbl[32 * JIT_ROW + 336 + 8 + 256];
    const VoxelArgs& a = j.v;
    const int lane = threadIdx.x;
    jit_load_table(tbl, lane);
    /* The wavefront's region is a ring of code slots.  A tape is translated into the slot after the one used
     * last, which nothing has been fetched from since the ring was last entered at slot 0 — so no instruction
     * cache can hold lines of it, and the invalidate (which also throws out the routines, this loop and the
     * translator for every wavefront of the CU pair) is needed only once per trip round the ring.  Slots are
     * separated by 256 unused bytes: the sequential prefetch that runs past the end of one slot's code must not
     * reach into the next slot before that is written. */
    /* development (MPR_JIT_DEBUG & 64): every wavefront runs ONE region's code (whatever tape got there first):
     * wrong results, but the speed of generated code that stays in the instruction caches */
    uint32_t* const region = j.code + (size_t)((j.debug & 64) ? 0 : blockIdx.x) * j.region_dwords;
    uint32_t* code = region;
    uint32_t slot = 0;
    bool ring_dirty = true;                   /* slot 0 may still be in an instruction cache from the last trip / frame */
    const uint64_t* __restrict__ const tro = a.tape_ro;
    const uint64_t head0 = tro[0];
    constexpr int SUB = (DIM == 3) ? 4 : 8;
    const int S = a.tps * SUB;
    const int4_ sub = unpack(lane, SUB);
    const float size_recip = 1.0f / (float)(unsigned)S;
    int cached_tape = -1;
    unsigned long long c_tr = 0, c_run = 0, n_tr = 0, n_run = 0;
    const unsigned long long t_begin = j.dbg ? __builtin_readcyclecounter() : 0ull;

    for (int run = blockIdx.x; run * JIT_RUN < a.count; run += gridDim.x) {
        for (int k = 0; k < JIT_RUN; ++k) {
            const int tile_index = run * JIT_RUN + k;
            if (tile_index >= a.count) break;
            const int position = __builtin_amdgcn_readfirstlane(a.tiles[tile_index].position);
            const int tape = __builtin_amdgcn_readfirstlane(a.tiles[tile_index].tape);
            const int4_ pos = unpack(position, a.tps);
            const int px = pos.x * SUB + sub.x;
            const int py = pos.y * SUB + sub.y;
            const int pz = (DIM == 3) ? pos.z * 4 + sub.z : 0;
            bool skip = false;
            if (DIM == 3) {
                /* reference :852-864: the thread owning (pz_low, pz_low + 2) leaves when image >= pz_low + 2;
                 * read past this CU's vector L1: the heights other tiles of the column have written so far */
                const int pz_low = pos.z * 4 + (sub.z & 1);
                skip = __hip_atomic_load(&a.image[px + py * S], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= pz_low + 2;
                if (ballot(!skip) == 0) continue;
            }
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
            uint32_t fresh = 0;
            const unsigned long long t0 = j.dbg ? __builtin_readcyclecounter() : 0ull;
            if (j.debug & 64) {
                volatile uint32_t* const flag = region + j.region_dwords - 1;
                if (cached_tape == -1) {
                    if (blockIdx.x == 0 && *flag != 0x600Du) {
                        (void)jit_translate(tro, (uint32_t)(tape + 1), head0, code, (uint32_t)((j.region_dwords - 320) * 4u) + (uint32_t)lane * 16u, tbl,
                                            lane, 0);
                        __builtin_amdgcn_s_waitcnt(0);
                        __threadfence();
                        *flag = 0x600Du;
                    }
                    while (*flag != 0x600Du) __builtin_amdgcn_s_sleep(10);
                    fresh = 1;
                }
                cached_tape = tape;
            } else if (tape != cached_tape && !((j.debug & 2) && cached_tape != -1)) {
                ++n_tr;
                if (cached_tape != -1) {
                    slot = slot + 1 == j.slots ? 0 : slot + 1;
                    if (slot == 0) ring_dirty = true;
                }
                code = region + (size_t)slot * j.slot_dwords;
                (void)jit_translate(tro, (uint32_t)(tape + 1), head0, code,
                                    (uint32_t)((j.region_dwords - 320 - slot * j.slot_dwords) * 4u) + (uint32_t)lane * 16u, tbl, lane,
                                    rdfirst((uint32_t)j.debug));
                cached_tape = tape;
                if (ring_dirty) fresh = 1;
                ring_dirty = false;
            }
            if (j.debug & 1) continue;
            if (!(j.debug & 8)) {
                /* the next tile's tape travels while this one runs */
                int nt = tile_index + 1;
                if (k + 1 == JIT_RUN) nt = (run + (int)gridDim.x) * JIT_RUN;
                int ntape = tape;
                if (nt < a.count) ntape = __builtin_amdgcn_readfirstlane(a.tiles[nt].tape);
                jit_prefetch_tape(tro, ntape, j.tape_len, (uint32_t)(uintptr_t)(tbl + 32 * JIT_ROW + 336 + 8), lane);
                fresh |= 2u;                              /* three loads are in flight behind the code's stores */
            }
            const unsigned long long t1 = j.dbg ? __builtin_readcyclecounter() : 0ull;
            const float res = jit_run<NS>(code, fresh, vx, vy, vz);
            if (j.dbg) {
                const unsigned long long t2 = __builtin_readcyclecounter();
                c_tr += t1 - t0;
                c_run += t2 - t1;
                ++n_run;
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
    }
    if (j.dbg && lane == 0) {
        atomicAdd(&j.dbg[0], c_tr);
        atomicAdd(&j.dbg[1], c_run);
        atomicAdd(&j.dbg[2], (unsigned long long)__builtin_readcyclecounter() - t_begin);
        atomicAdd(&j.dbg[3], n_tr);
        atomicAdd(&j.dbg[4], n_run);
        atomicAdd(&j.dbg[5], 1ull);
    }
}

/* one clause through the translator and the generated code: tape3 as for k_test_float_asm */
__global__ void __launch_bounds__(64)
k_test_float_jit(const uint64_t* tape3, uint32_t* code, uint32_t region_dwords, int n, const float* a, const float* b, float* out)
{
    __shared__ __attribute__((aligned(64))) uint32_t tbl[32 * JIT_ROW + 336 + 8];
    const int lane = threadIdx.x;
    jit_load_table(tbl, lane);
    const int i = blockIdx.x * 64 + lane;
    uint32_t* const my = code + (size_t)blockIdx.x * region_dwords;
    (void)jit_translate(tape3, 1u, tape3[0], my, (region_dwords - 320) * 4u + (uint32_t)lane * 16u, tbl, lane);
    const float r = jit_run<24>(my, 1u, i < n ? a[i] : 0.0f, (i < n && b) ? b[i] : 0.0f, 0.0f);
    if (i < n) out[i] = r;
}
void launch_test_float_jit(hipStream_t s, const uint64_t* tape3, uint32_t* code, uint32_t region_dwords, int n, const float* a,
                           const float* b, float* out)
{
    hipLaunchKernelGGL(k_test_float_jit, dim3((n + 63) / 64), dim3(64), 0, s, tape3, code, region_dwords, n, a, b, out);
}

int jit_slot_class(int nslots)
{
    return nslots <= 24 ? 24 : nslots <= 40 ? 40 : nslots <= 96 ? 96 : nslots <= 192 ? 192 : 0;
}
template <int DIM, int NS>
static int jit_grid_of(int cus)
{
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_eval_voxels_jit<DIM, NS>, 64, 0) != hipSuccess || per_cu <= 0) per_cu = 8;
    return per_cu * cus;
}
/* wavefronts the device holds of the kernel for this slot class: the grid, and the number of code regions */
int jit_grid(int dim, int nslots, int cus)
{
    const int ns = jit_slot_class(nslots);
    if (dim == 3) return ns == 24 ? jit_grid_of<3, 24>(cus) : ns == 40 ? jit_grid_of<3, 40>(cus) : ns == 96 ? jit_grid_of<3, 96>(cus) : jit_grid_of<3, 192>(cus);
    return ns == 24 ? jit_grid_of<2, 24>(cus) : ns == 40 ? jit_grid_of<2, 40>(cus) : ns == 96 ? jit_grid_of<2, 96>(cus) : jit_grid_of<2, 192>(cus);
}
void launch_eval_voxels_jit(hipStream_t s, int dim, const VoxelArgs& a, uint32_t* code, uint32_t region_dwords, int slot_dwords, int slots, int grid, int debug, int tape_len,
                            unsigned long long* dbg)
{
    if (a.count <= 0) return;
    JitVoxelArgs j;
    j.v = a;
    j.code = code;
    j.region_dwords = region_dwords;
    j.slot_dwords = (uint32_t)slot_dwords;
    j.slots = (uint32_t)slots;
    j.debug = debug;
    j.tape_len = tape_len;
    j.dbg = dbg;
    const int runs = (a.count + JIT_RUN - 1) / JIT_RUN;
    const dim3 g(std::min(grid, runs)), b(64);
    const int ns = jit_slot_class(a.nslots);
#define JIT_LAUNCH(D, N) hipLaunchKernelGGL((k_eval_voxels_jit<D, N>), g, b, 0, s, j)
    if (dim == 3) {
        if (ns == 24) JIT_LAUNCH(3, 24); else if (ns == 40) JIT_LAUNCH(3, 40); else if (ns == 96) JIT_LAUNCH(3, 96); else JIT_LAUNCH(3, 192);
    } else {
        if (ns == 24) JIT_LAUNCH(2, 24); else if (ns == 40) JIT_LAUNCH(2, 40); else if (ns == 96) JIT_LAUNCH(2, 96); else JIT_LAUNCH(2, 192);
    }
#undef JIT_LAUNCH
}

}  // namespace mprk
