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
 * names; they use v0..v31 and s0..s31 at most (tests/test_callee_registers.py) */
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

/* LDS copy of the table: words [op * 8 + k] = base k (k < 5), [op * 8 + 5] = prog, [op * 8 + 6] = prog4 | n << 8 */
DEV void jit_load_table(uint32_t* lds, int lane)
{
    for (int i = lane; i < 32 * 8; i += 64) {
        const int op = i >> 3, k = i & 7;
        uint32_t v = 0;
        if (k < 5) v = d_jit_table.base[op][k];
        else if (k == 5) v = d_jit_table.prog[op];
        else if (k == 6) v = (uint32_t)d_jit_table.prog4[op] | ((uint32_t)d_jit_table.n[op] << 8);
        lds[i] = v;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
}

/* one template dword with the clause's registers and immediate filled in */
DEV uint32_t jit_word(uint32_t base, uint32_t p, uint32_t regs, uint32_t imm, uint32_t cimm)
{
    const uint32_t d = (regs >> ((p & 3u) * 8u)) & 0xFFu;
    const uint32_t s1 = (regs >> (((p >> 2) & 3u) * 8u)) & 0xFFu;
    const uint32_t s0 = (regs >> (((p >> 4) & 3u) * 8u)) & 0xFFu;
    uint32_t w = base | (d << 17) | (s1 << 9) | s0;
    const uint32_t kind = (p >> 6) & 3u;
    w = kind == 1 ? imm : w;
    w = kind == 2 ? cimm : w;
    return w;
}

/* Translate the tape whose first clause is tro[first] into `code`; returns the number of dwords.
 * head0 = the root tape's head clause (axis slots), the same for every tape of the frame. */
DEV uint32_t jit_translate(const uint64_t* __restrict__ tro, uint32_t first, uint64_t head0, uint32_t* __restrict__ code,
                           const uint32_t* __restrict__ tbl, int lane)
{
    uint32_t used = 0;
    if (lane < 3) {
        const uint32_t slot = (uint32_t)(head0 >> (8 * (lane + 1))) & 0xFFu;
        code[lane] = jt::MOV(JIT_SLOT_BASE + slot, jt::VREG + 32 + lane);        /* v[48 + axis slot] = v32 / v33 / v34 */
    }
    used = 3;
    uint32_t base = first;
    for (;;) {
        const uint64_t c = tro[base + lane];
        const uint32_t lo = (uint32_t)c, imm = (uint32_t)(c >> 32);
        const uint32_t op = lo & 0xFFu;
        const uint64_t term = ballot(op < 2u);                       /* end of tape or JUMP */
        const int first_term = term ? __builtin_ctzll(term) : 64;
        const uint32_t opx = op < 32u ? op : 0u;
        const uint32_t* const row = tbl + opx * 8;
        const uint32_t meta = row[6];
        const uint32_t n = lane < first_term ? (meta >> 8) : 0u;
        /* registers of the clause: byte 1 out, byte 2 lhs, byte 3 rhs (byte 0 = "none" = 0) */
        const uint32_t regs = ((lo & 0xFFFFFF00u) + 0x30303000u) & 0xFFFFFF00u;        /* + 48 each; slots <= 191 */
        float cf;
        {
            const float f = mpr_u2f(imm);
            asm("v_max_f32 %0, %1, %1" : "=v"(cf) : "v"(f));                          /* what the handlers do to an immediate */
        }
        const uint32_t cimm = mpr_f2u(cf);
        /* where this lane's dwords go: exclusive prefix sum of n (0..5) over the lanes */
        const uint64_t b0 = ballot(n & 1u), b1 = ballot(n & 2u), b2 = ballot(n & 4u);
        const uint64_t below = (1ull << lane) - 1ull;
        const uint32_t off = (uint32_t)__popcll(b0 & below) + 2u * (uint32_t)__popcll(b1 & below) + 4u * (uint32_t)__popcll(b2 & below);
        const uint32_t total = (uint32_t)__popcll(b0) + 2u * (uint32_t)__popcll(b1) + 4u * (uint32_t)__popcll(b2);
        uint32_t* const dst = code + used + off;
        const uint32_t prog = row[5];
        if (n > 0) dst[0] = jit_word(row[0], prog & 0xFFu, regs, imm, cimm);
        if (n > 1) dst[1] = jit_word(row[1], (prog >> 8) & 0xFFu, regs, imm, cimm);
        if (n > 2) dst[2] = jit_word(row[2], (prog >> 16) & 0xFFu, regs, imm, cimm);
        if (n > 3) dst[3] = jit_word(row[3], prog >> 24, regs, imm, cimm);
        if (n > 4) dst[4] = jit_word(row[4], meta & 0xFFu, regs, imm, cimm);
        used += total;
        if (first_term < 64) {
            const uint32_t tlo = rdlane(lo, (uint32_t)first_term), thi = rdlane(imm, (uint32_t)first_term);
            if ((tlo & 0xFFu) == 0u) {
                /* end clause: byte 1 names the result slot */
                if (lane == 0) {
                    code[used] = jt::MOV(37, jt::VREG + JIT_SLOT_BASE + ((tlo >> 8) & 0xFFu));
                    code[used + 1] = 0xBE801D00u | 72u;                              /* s_setpc_b64 s[72:73] */
                }
                return used + 2;
            }
            base = base + (uint32_t)first_term + thi + 1u;                           /* JUMP: relative, then pre-increment */
        } else {
            base += 64;
        }
    }
}

/* Run the code at `code` on (vx, vy, vz); `fresh`: the region was just rewritten.  NS: slots the
 * kernel provides registers for.  The register lists are what the generated code, the routines and the
 * compiled leaf routines may touch. */
#define JIT_V10(a) "v" #a "0", "v" #a "1", "v" #a "2", "v" #a "3", "v" #a "4", "v" #a "5", "v" #a "6", "v" #a "7", "v" #a "8", "v" #a "9"
#define JIT_CLOBBER_BASE                                                                                                   \
    "memory", "vcc", "scc", "s0", "s1", "s2", "s3", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13", "s14", "s15", "s16", \
        "s17", "s18", "s19", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s40", "s41", "s42",      \
        "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60",      \
        "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s90", "s91", "s92",      \
        "s93", "s94", "s95", "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", JIT_V10(1), JIT_V10(2), JIT_V10(3), JIT_V10(4),   \
        JIT_V10(5), JIT_V10(6), "v70", "v71"
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
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n"                                                                  \
    "s_cmp_eq_u32 %[fresh], 0\n"                                                                       \
    "s_cbranch_scc1 L_cached_%=\n"                                                                     \
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
    uint32_t region_dwords;
};

template <int DIM, int NS>
__global__ void __launch_bounds__(64)
k_eval_voxels_jit(JitVoxelArgs j)
{
    __shared__ uint32_t tbl[32 * 8];
    const VoxelArgs& a = j.v;
    const int lane = threadIdx.x;
    jit_load_table(tbl, lane);
    uint32_t* const code = j.code + (size_t)blockIdx.x * j.region_dwords;
    const uint64_t* __restrict__ const tro = a.tape_ro;
    const uint64_t head0 = tro[0];
    constexpr int SUB = (DIM == 3) ? 4 : 8;
    const int S = a.tps * SUB;
    const int4_ sub = unpack(lane, SUB);
    const float size_recip = 1.0f / (float)(unsigned)S;
    int cached_tape = -1;

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
            if (tape != cached_tape) {
                (void)jit_translate(tro, (uint32_t)(tape + 1), head0, code, tbl, lane);
                cached_tape = tape;
                fresh = 1;
            }
            const float res = jit_run<NS>(code, fresh, vx, vy, vz);
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
}

/* one clause through the translator and the generated code: tape3 as for k_test_float_asm */
__global__ void __launch_bounds__(64)
k_test_float_jit(const uint64_t* tape3, uint32_t* code, uint32_t region_dwords, int n, const float* a, const float* b, float* out)
{
    __shared__ uint32_t tbl[32 * 8];
    const int lane = threadIdx.x;
    jit_load_table(tbl, lane);
    const int i = blockIdx.x * 64 + lane;
    uint32_t* const my = code + (size_t)blockIdx.x * region_dwords;
    (void)jit_translate(tape3, 1u, tape3[0], my, tbl, lane);
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
void launch_eval_voxels_jit(hipStream_t s, int dim, const VoxelArgs& a, uint32_t* code, uint32_t region_dwords, int grid)
{
    if (a.count <= 0) return;
    JitVoxelArgs j;
    j.v = a;
    j.code = code;
    j.region_dwords = region_dwords;
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
