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
 *   - a wavefront translates a tape 64 clauses at a time: lane j decodes clause j into up to 12 dwords
 *     through a 32-entry template table in LDS (in three batches of four), a ballot prefix sum places them,
 *     they are staged in LDS and leave as three 16-byte stores per lane to the wavefront's (group form: the
 *     workgroup's) own region of an EXECUTABLE buffer (context.hip: HSA executable pool); JUMP / chunk links
 *     disappear; a division by a constant becomes a multiplication by its reciprocal plus an exact correction
 *     (L_divc: the reciprocal is computed once, by the translator);
 *   - then `s_waitcnt vmcnt(0)`, `s_icache_inv` (the region is rewritten tile after tile: stale lines
 *     of the previous tape would otherwise execute — scripts/ubench/jit_probe.hip shows exactly that
 *     without the invalidate, and zero stale rounds with it), `s_swappc_b64` into the region.  Code is
 *     only ever executed by the wavefront that wrote it, after its own invalidate, so no other
 *     instruction cache can hold a stale copy that matters;
 *   - wavefronts are persistent (grid = what the chip holds) and walk the tile list in runs of eight
 *     neighbours; a tile whose tape is the one just translated reuses the code and skips the invalidate.
 *
 * Generated code is fetched through the instruction caches (64 KB per CU pair, about 5 bytes per clock each when
 * it misses: profiles/r02a_jit_probe.txt), and code that is written per tile never hits: measured, the same
 * instructions run 4x faster from a warm instruction cache.  Hence the GROUP form (k_eval_voxels_jit_groups),
 * used when the tapes of the last tile stage record at most 128 min / max decisions and are not much longer than the
 * tapes that stage hands on (context.hip: the stage samples both lengths): the 64 children of a tile
 * walked ONE tape in the last interval stage, and a child's own shortened tape is that tape with the child's
 * min / max decisions applied and dead clauses dropped.  So the group's tape is translated once, by one wavefront
 * of a workgroup, with every min / max followed by two selects steered by the child's decisions (bit i of two
 * scalar registers: `chose lhs`, `chose rhs`; the last tile stage stores the decision masks per group), and the
 * workgroup's wavefronts run that one piece of code — a few KB that stay in the instruction cache — for every
 * surviving child.  Dropped clauses are evaluated for nothing (3 % more clauses for bear); the values of all
 * others are the ones the child's own tape gives, bit for bit.  Workgroups are persistent and take the groups that
 * still have a tile one at a time, in list order (front to back), as they come free.
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

/* ---- the template tables ------------------------------------------------------------------------ */
/* A clause becomes up to 12 dwords.  Each dword of a template is (base, selector, mask):
 *   T = v_perm_b32(literal, regs, selector): regs = {128 + index of this min / max, 48 + out, 48 + lhs, 48 + rhs} (bytes
 *       0..3), literal = the clause's immediate (canonicalised if the row says so); selector bytes 0..3 pick from regs,
 *       4..7 from the literal, 0x0C is zero
 *   dword = T + (T & mask) + base
 * For a VOP1 / VOP2 word T holds {src0, vsrc1, vdst, 0} and mask = 0xFFFFFF00 doubles bytes 1..2 into bit positions 9
 * and 17; for a literal T is the literal and mask = base = 0; for s_bitcmp1_b64 T holds the decision index in byte 1
 * (ssrc1 as an inline integer constant) and mask = 0. */
constexpr int JIT_WORDS = 16;               /* a row holds up to 15 dwords (four batches of four) */
constexpr int JIT_ROW = 4 + 3 * JIT_WORDS;           /* dwords per row: meta, 3 pad, then four batches of {base x4, sel x4, mask x4} */
struct JitRow { uint32_t n, flags; uint32_t base[JIT_WORDS], sel[JIT_WORDS], mask[JIT_WORDS]; };
constexpr int JIT_ROWS = 36;               /* 32 opcodes (30: the translator's constant division), then min / max for decisions 64..127 */
struct JitTable { uint32_t w[JIT_ROWS][JIT_ROW]; };
namespace jt {
constexpr uint32_t NONE = 0x0C, CI = 0, O = 1, A = 2, R = 3;                 /* selector bytes */
constexpr uint32_t VREG = 0x100;                                               /* src0 names a VGPR */
constexpr uint32_t LITERAL = 255;
constexpr uint32_t VOP2(uint32_t op, uint32_t vdst, uint32_t vsrc1, uint32_t src0) { return (op << 25) | (vdst << 17) | (vsrc1 << 9) | src0; }
constexpr uint32_t MOV(uint32_t vdst, uint32_t src0) { return 0x7E000200u | (vdst << 17) | src0; }
constexpr uint32_t CALL(uint32_t sgpr) { return 0xBE9E1E00u | sgpr; }        /* s_swappc_b64 s[30:31], s[sgpr:sgpr+1] */
constexpr uint32_t BITCMP_L = 0xBF0F004Cu, BITCMP_R = 0xBF0F004Eu;            /* s_bitcmp1_b64 s[76:77] / s[78:79], <index>; decisions 64..127 in s[48:49] / s[50:51] */
constexpr uint32_t BITCMP_L1 = 0xBF0F0030u, BITCMP_R1 = 0xBF0F0032u;
constexpr int JIT_MAX_CHOICES = 128;
constexpr uint32_t CSELECT = 0x85DE80C1u;                                      /* s_cselect_b64 s[94:95], -1, 0 */
/* v_cndmask_b32_e64 vdst, src0, vsrc1, s[94:95] (VOP3: two dwords).  NOT the VOP2 form with vcc: a v_cndmask that reads a vcc
 * the SCALAR unit wrote issues at a tenth of the VALU rate (0.175 wave-instr/clk/CU at any occupancy, scripts/ubench/
 * issue_rates2.hip: "v_cndmask"), the VOP3 form with an SGPR pair at half rate (0.82-0.90: "v_cndmask e64 sgpr") */
constexpr uint32_t CND64_LO = 0xD1000000u;                                     /* | vdst */
constexpr uint32_t CND64_HI = (94u << 18) | (1u << 17);                        /* | src0 (9 bits) | vsrc1 << 9; src1 is a VGPR */
constexpr uint32_t V_ADD = 1, V_SUB = 2, V_SUBREV = 3, V_MUL = 5, V_MIN = 10, V_MAX = 11, V_AND = 19, V_XOR = 21, V_FMAMK = 23;
constexpr uint32_t CLASS_V39_V7 = 0x7C200F27u;                                 /* v_cmp_class_f32 vcc, v39, v7 */
constexpr uint32_t BRANCH_VCCZ_2 = 0xBF860002u;                                /* s_cbranch_vccz +2 dwords */
constexpr uint32_t CLASS_NOT_POSITIVE_NORMAL = 0x2FFu;                         /* what v7 holds while generated code runs */
constexpr uint32_t S_DIVC = 70, S_DIV = 52, S_SQRT = 54, S_EXP = 56, S_LOG = 58, S_SIN = 60, S_COS = 62, S_ASIN = 64, S_ACOS = 66, S_ATAN = 68;
constexpr uint32_t FLAG_MINMAX = 1;

struct Builder {
    JitRow r{};
    constexpr void ins(uint32_t base, uint32_t d = NONE, uint32_t s1 = NONE, uint32_t s0 = NONE)      /* an instruction word */
    {
        r.base[r.n] = base;
        r.sel[r.n] = s0 | (s1 << 8) | (d << 16) | (NONE << 24);
        r.mask[r.n] = 0xFFFFFF00u;
        ++r.n;
    }
    constexpr void lit() { r.base[r.n] = 0; r.sel[r.n] = 0x07060504u; r.mask[r.n] = 0; ++r.n; }      /* the literal */
    constexpr void neglit() { r.base[r.n] = 0x80000000u; r.sel[r.n] = 0x07060504u; r.mask[r.n] = 0; ++r.n; }   /* the literal with its sign flipped */
    constexpr void bitcmp(uint32_t base) { r.base[r.n] = base; r.sel[r.n] = NONE | (CI << 8) | (NONE << 16) | (NONE << 24); r.mask[r.n] = 0; ++r.n; }
    constexpr void fixed(uint32_t w) { r.base[r.n] = w; r.sel[r.n] = NONE | (NONE << 8) | (NONE << 16) | (NONE << 24); r.mask[r.n] = 0; ++r.n; }
    /* vdst = s[94:95] ? vsrc1 : v37.  d: selector of vdst (NONE: v37); s1: selector of vsrc1 (NONE: v38) */
    constexpr void cnd64(uint32_t d, uint32_t s1)
    {
        r.base[r.n] = CND64_LO | (d == NONE ? 37u : 0u);
        r.sel[r.n] = d | (NONE << 8) | (NONE << 16) | (NONE << 24);
        r.mask[r.n] = 0;
        ++r.n;
        r.base[r.n] = CND64_HI | (VREG + 37) | (s1 == NONE ? (38u << 9) : 0u);
        r.sel[r.n] = NONE | (s1 << 8) | (NONE << 16) | (NONE << 24);
        r.mask[r.n] = 0xFFFFFF00u;
        ++r.n;
    }
};
constexpr JitRow unary_call(uint32_t s)
{
    Builder b;
    b.ins(MOV(35, VREG), NONE, NONE, A);
    b.fixed(CALL(s));
    b.ins(MOV(0, VREG + 37), O);
    return b.r;
}
constexpr JitRow leaf_call(uint32_t s)
{
    Builder b;
    b.ins(MOV(0, VREG), NONE, NONE, A);
    b.fixed(CALL(s));
    b.ins(MOV(0, VREG + 0), O);
    b.fixed(MOV(7, LITERAL));                           /* the compiled leaves may use v0..v7 */
    b.fixed(CLASS_NOT_POSITIVE_NORMAL);
    return b.r;
}
constexpr JitRow one(uint32_t base, uint32_t d, uint32_t s1, uint32_t s0, bool imm = false)
{
    Builder b;
    b.ins(base, d, s1, s0);
    if (imm) b.lit();
    return b.r;
}
/* min / max.  Generated code runs with MODE.IEEE off (jit_run), where v_min_f32 / v_max_f32 return the other operand for
 * ANY NaN — fminf's rule, which the interpreters get by canonicalising both operands first (with IEEE on, a signalling
 * NaN would win as a quiet one).  Tile form: the operation.  Group form: the result goes to v37 first, then `chose lhs`
 * puts the lhs there instead and `chose rhs` the rhs (raw, as COPY_LHS / COPY_RHS / COPY_IMM of the child's own tape
 * would), and that lands in the out register. */
constexpr JitRow minmax(uint32_t op, bool imm, bool group, uint32_t q = 0)
{
    Builder b;
    b.r.flags = FLAG_MINMAX;
    if (!group) {
        if (imm) {
            b.ins(VOP2(op, 0, 0, LITERAL), O, A, NONE);
            b.lit();
        } else {
            b.ins(VOP2(op, 0, 0, VREG), O, R, A);
        }
        return b.r;
    }
    if (imm) {
        b.fixed(MOV(38, LITERAL));
        b.lit();
        b.ins(VOP2(op, 37, 38, VREG), NONE, NONE, A);
    } else {
        b.ins(VOP2(op, 37, 0, VREG), NONE, R, A);
    }
    b.bitcmp(q ? BITCMP_L1 : BITCMP_L);
    b.fixed(CSELECT);
    b.cnd64(NONE, A);                                                    /* v37 = chose lhs ? lhs : v37 */
    b.bitcmp(q ? BITCMP_R1 : BITCMP_R);
    b.fixed(CSELECT);
    if (imm) b.cnd64(O, NONE);                                           /* out = chose rhs ? immediate (v38) : v37 */
    else b.cnd64(O, R);
    return b.r;
}
constexpr JitRow row_of(uint32_t op, bool group)
{
    switch (op) {
        case MPR_OP_SQUARE_LHS: return one(VOP2(V_MUL, 0, 0, VREG), O, A, A);
        case MPR_OP_SQRT_LHS: return unary_call(S_SQRT);
        case MPR_OP_NEG_LHS: {                          /* a literal, not an SGPR operand: those halve the VALU rate (profiles/r02a_issue_rates.txt) */
            Builder b;
            b.ins(VOP2(V_XOR, 0, 0, LITERAL), O, A, NONE);
            b.fixed(0x80000000u);
            return b.r;
        }
        case MPR_OP_SIN_LHS: return unary_call(S_SIN);
        case MPR_OP_COS_LHS: return unary_call(S_COS);
        case MPR_OP_ASIN_LHS: return leaf_call(S_ASIN);
        case MPR_OP_ACOS_LHS: return leaf_call(S_ACOS);
        case MPR_OP_ATAN_LHS: return leaf_call(S_ATAN);
        case MPR_OP_EXP_LHS: return unary_call(S_EXP);
        case MPR_OP_ABS_LHS: {
            Builder b;
            b.ins(VOP2(V_AND, 0, 0, LITERAL), O, A, NONE);
            b.fixed(0x7FFFFFFFu);
            return b.r;
        }
        case MPR_OP_LOG_LHS: return unary_call(S_LOG);
        case MPR_OP_ADD_LHS_IMM: return one(VOP2(V_ADD, 0, 0, LITERAL), O, A, NONE, true);
        case MPR_OP_ADD_LHS_RHS: return one(VOP2(V_ADD, 0, 0, VREG), O, R, A);
        case MPR_OP_MUL_LHS_IMM: return one(VOP2(V_MUL, 0, 0, LITERAL), O, A, NONE, true);
        case MPR_OP_MUL_LHS_RHS: return one(VOP2(V_MUL, 0, 0, VREG), O, R, A);
        case MPR_OP_MIN_LHS_IMM: return minmax(V_MIN, true, group);
        case MPR_OP_MIN_LHS_RHS: return minmax(V_MIN, false, group);
        case MPR_OP_MAX_LHS_IMM: return minmax(V_MAX, true, group);
        case MPR_OP_MAX_LHS_RHS: return minmax(V_MAX, false, group);
        case MPR_OP_SUB_LHS_IMM: return one(VOP2(V_SUBREV, 0, 0, LITERAL), O, A, NONE, true);      /* lhs - imm */
        case MPR_OP_SUB_IMM_RHS: return one(VOP2(V_SUB, 0, 0, LITERAL), O, R, NONE, true);         /* imm - rhs */
        case MPR_OP_SUB_LHS_RHS: return one(VOP2(V_SUB, 0, 0, VREG), O, R, A);
        case MPR_OP_DIV_LHS_IMM: {
            Builder b;
            b.ins(MOV(35, VREG), NONE, NONE, A);
            b.fixed(MOV(36, LITERAL));
            b.lit();
            b.fixed(CALL(S_DIV));
            b.ins(MOV(0, VREG + 37), O);
            return b.r;
        }
        case MPR_OP_DIV_IMM_RHS: {
            Builder b;
            b.fixed(MOV(35, LITERAL));
            b.lit();
            b.ins(MOV(36, VREG), NONE, NONE, R);
            b.fixed(CALL(S_DIV));
            b.ins(MOV(0, VREG + 37), O);
            return b.r;
        }
        case MPR_OP_DIV_LHS_RHS: {
            Builder b;
            b.ins(MOV(35, VREG), NONE, NONE, A);
            b.ins(MOV(36, VREG), NONE, NONE, R);
            b.fixed(CALL(S_DIV));
            b.ins(MOV(0, VREG + 37), O);
            return b.r;
        }
        case MPR_OP_COPY_IMM: return one(MOV(0, LITERAL), O, NONE, NONE, true);
        case MPR_OP_COPY_LHS: return one(MOV(0, VREG), O, NONE, A);
        case MPR_OP_COPY_RHS: return one(MOV(0, VREG), O, NONE, R);
        case 30: {
            /* not an opcode: the translator's own row for DIV_LHS_IMM by a constant c that is not a power of two
             * (2^-30 <= |c| <= 2^30), with y = RN(1 / c) (the translator computes it with an IEEE division and puts it
             * in dwords 6, 10 and 14).  Inline when x * x is a positive normal number in every lane (2^-63 <= |x| < 2^64:
             * nothing over- or underflows on the way): q = x y, then two rounds of r = x - c q (exact, fused),
             * q += r y end on the correctly rounded quotient.  Any other operand: L_divc divides in general, and comes
             * back to the last instruction with v39 = a zero and v37 = the quotient. */
            Builder b;
            b.ins(VOP2(V_MUL, 39, 0, VREG), NONE, A, A);                 /* v39 = x * x */
            b.fixed(CLASS_V39_V7);                                       /* vcc = lanes where that is anything but a positive normal */
            b.fixed(BRANCH_VCCZ_2);
            b.ins(MOV(35, VREG), NONE, NONE, A);
            b.fixed(CALL(S_DIVC));
            b.ins(VOP2(V_MUL, 37, 0, LITERAL), NONE, A, NONE);           /* v37 = y * x */
            b.fixed(0);
            b.ins(VOP2(V_FMAMK, 39, 0, VREG + 37), NONE, A, NONE);       /* v39 = v37 * (-c) + x */
            b.neglit();
            b.fixed(VOP2(V_FMAMK, 37, 37, VREG + 39));                   /* v37 = v39 * y + v37 */
            b.fixed(0);
            b.ins(VOP2(V_FMAMK, 39, 0, VREG + 37), NONE, A, NONE);
            b.neglit();
            b.ins(VOP2(V_FMAMK, 0, 37, VREG + 39), O);                   /* out = v39 * y + v37 */
            b.fixed(0);
            return b.r;
        }
        default: return JitRow{};                                       /* end, JUMP, not an opcode: no code */
    }
}
constexpr JitTable make_table(bool group)
{
    JitTable t{};
    for (uint32_t op = 0; op < (uint32_t)JIT_ROWS; ++op) {
        /* rows 32..35: MIN_LHS_IMM .. MAX_LHS_RHS whose decision is one of 64..127 */
        const uint32_t mm = MPR_OP_MIN_LHS_IMM + (op - 32) % 4, q = 1 + (op - 32) / 4;
        const JitRow r = op < 32 ? row_of(op, group)
                                 : minmax(mm == MPR_OP_MIN_LHS_IMM || mm == MPR_OP_MIN_LHS_RHS ? V_MIN : V_MAX,
                                          mm == MPR_OP_MIN_LHS_IMM || mm == MPR_OP_MAX_LHS_IMM, group, group ? q : 0);
        t.w[op][0] = r.n | (r.flags << 8);
        for (int k = 0; k < JIT_WORDS; ++k) {
            const int at = 4 + 12 * (k / 4) + (k % 4);
            t.w[op][at] = r.base[k];
            t.w[op][at + 4] = r.sel[k];
            t.w[op][at + 8] = r.mask[k];
        }
    }
    return t;
}
}  // namespace jt
static const JitTable h_jit_table[2] = {jt::make_table(false), jt::make_table(true)};
__constant__ JitTable d_jit_table[2] = {jt::make_table(false), jt::make_table(true)};

/* dwords of the code of a tape (host; the bound for every tape shortened from it) */
size_t jit_code_dwords(const uint64_t* clauses, int n, bool group)
{
    size_t d = 3 + 2;                         /* prologue: three axis moves; epilogue: result move, return */
    for (int i = 0; i < n; ++i) {
        const uint32_t op = mpr_cl_op(clauses[i]) & 31;
        uint32_t w = h_jit_table[group ? 1 : 0].w[op][0] & 15u;
        if (op == MPR_OP_DIV_LHS_IMM) w = std::max(w, h_jit_table[0].w[30][0] & 15u);        /* may become the translator's row 30 */
        d += w;
    }
    return d;
}

/* LDS of the translator (dwords): the table, the staging buffer (a block's dwords side by side before they leave in
 * 16-byte pieces: 3 carried + 62 x 15 + slack), a dump for lanes without a dword, a dump for the tape prefetch */
constexpr int JIT_LDS_STAGE = JIT_ROWS * JIT_ROW, JIT_STAGE_DWORDS = 1056, JIT_LDS_DUMP = JIT_LDS_STAGE + JIT_STAGE_DWORDS;
constexpr int JIT_LDS_PFDUMP = JIT_LDS_DUMP + 16, JIT_LDS_DWORDS = JIT_LDS_PFDUMP + 256;
DEV void jit_load_table(uint32_t* lds, int tid, int nthreads, bool group)
{
    const uint32_t* const src = &d_jit_table[group ? 1 : 0].w[0][0];
    for (int i = tid; i < JIT_ROWS * JIT_ROW; i += nthreads) lds[i] = src[i];
}

/* Translate the tape whose first clause is tro[first] into `code`; returns the number of dwords.
 * head0 = the root tape's head clause (axis slots), the same for every tape of the frame.
 * trash_off: byte offset (from `code`) of this lane's 16 bytes in a part of the region that is never executed.
 *
 * Written in assembly because two things decide its cost.  (1) Latency: a tape is a linked list of 64-word chunks, so
 * the next block's address is known only once the current block has arrived; the next block is requested the moment
 * the current one is in registers, before any of the work on it, and no memory operation of that work is conditional
 * — the dwords of a block are put side by side in LDS and leave in exactly four 16-byte stores per lane (the ones
 * beyond the end go to the dump) — so the wait for the next block is the exact `s_waitcnt vmcnt(4)`.  (2) Instruction
 * count: a template dword gets the clause's registers with one v_perm_b32, one v_and_b32 and one v_add3_u32.
 *   s42 block base (clause index)   s43 dwords that have left   s44 dwords waiting in the staging buffer
 *   s46 first terminator lane (64: none)   s48 min / max clauses so far   s52 / s53 terminator lo / hi
 *   s54 its opcode (0 end, 1 jump)   s55 next block base   v[34:35] block (lane j = clause j)   v[36:37] next block
 *   v38 row   v39 meta   v52 the lane's place in the staging buffer   v56 regs   v57 literal   v59 dwords of the lane
 * (registers the generated code owns while it runs and nobody needs while it is being written) */
#define JIT_BATCH(off, k0)                                                                                             \
    "ds_read_b128 v[40:43], v38 offset:" #off "\n"                                                                      \
    "ds_read_b128 v[44:47], v38 offset:" #off "+16\n"                                                                   \
    "ds_read_b128 v[48:51], v38 offset:" #off "+32\n"                                                                   \
    "s_waitcnt lgkmcnt(0)\n"                                                                                           \
    "v_perm_b32 v44, v57, v56, v44\n v_and_b32 v48, v48, v44\n v_add3_u32 v40, v44, v48, v40\n"                         \
    "v_perm_b32 v45, v57, v56, v45\n v_and_b32 v49, v49, v45\n v_add3_u32 v41, v45, v49, v41\n"                         \
    "v_perm_b32 v46, v57, v56, v46\n v_and_b32 v50, v50, v46\n v_add3_u32 v42, v46, v50, v42\n"                         \
    "v_perm_b32 v47, v57, v56, v47\n v_and_b32 v51, v51, v47\n v_add3_u32 v43, v47, v51, v43\n"                         \
    "ds_write_b32 v52, v43 offset:(" #k0 "+3)*4\n"                                                                      \
    "ds_write_b32 v52, v42 offset:(" #k0 "+2)*4\n"                                                                      \
    "ds_write_b32 v52, v41 offset:(" #k0 "+1)*4\n"                                                                      \
    "ds_write_b32 v52, v40 offset:(" #k0 ")*4\n"
/* the same for dwords 4.., with the batch's third dword (6, 10, 14) of the row-30 lanes (s[70:71]) replaced by the
 * reciprocal in v55 */
#define JIT_BATCH_PATCH(off, k0)                                                                                       \
    "ds_read_b128 v[40:43], v38 offset:" #off "\n"                                                                      \
    "ds_read_b128 v[44:47], v38 offset:" #off "+16\n"                                                                   \
    "ds_read_b128 v[48:51], v38 offset:" #off "+32\n"                                                                   \
    "s_waitcnt lgkmcnt(0)\n"                                                                                           \
    "v_perm_b32 v44, v57, v56, v44\n v_and_b32 v48, v48, v44\n v_add3_u32 v40, v44, v48, v40\n"                         \
    "v_perm_b32 v45, v57, v56, v45\n v_and_b32 v49, v49, v45\n v_add3_u32 v41, v45, v49, v41\n"                         \
    "v_perm_b32 v46, v57, v56, v46\n v_and_b32 v50, v50, v46\n v_add3_u32 v42, v46, v50, v42\n"                         \
    "v_perm_b32 v47, v57, v56, v47\n v_and_b32 v51, v51, v47\n v_add3_u32 v43, v47, v51, v43\n"                         \
    "v_cndmask_b32 v42, v42, v55, s[70:71]\n"                                                                          \
    "ds_write_b32 v52, v43 offset:(" #k0 "+3)*4\n"                                                                      \
    "ds_write_b32 v52, v42 offset:(" #k0 "+2)*4\n"                                                                      \
    "ds_write_b32 v52, v41 offset:(" #k0 "+1)*4\n"                                                                      \
    "ds_write_b32 v52, v40 offset:(" #k0 ")*4\n"
DEV uint32_t jit_translate(const uint64_t* __restrict__ tro, uint32_t first, uint64_t head0, uint32_t* __restrict__ code,
                           uint32_t trash_off, uint32_t* __restrict__ lds, int lane)
{
    uint32_t* const stage = lds + JIT_LDS_STAGE;
    if (lane < 3) {
        const uint32_t slot = (uint32_t)(head0 >> (8 * (lane + 1))) & 0xFFu;
        stage[lane] = jt::MOV(JIT_SLOT_BASE + slot, jt::VREG + 32 + lane);       /* v[48 + axis slot] = v32 / v33 / v34 */
    }
    const uint32_t tlo = rdfirst((uint32_t)(uintptr_t)tro), thi = rdfirst((uint32_t)((uintptr_t)tro >> 32));
    const uint64_t cbase = rfl64((uint64_t)(uintptr_t)code);
    const uint32_t ltab = rdfirst((uint32_t)(uintptr_t)lds);
    const uint32_t lstage = rdfirst((uint32_t)(uintptr_t)stage), ldump = rdfirst((uint32_t)(uintptr_t)(lds + JIT_LDS_DUMP));
    const uint32_t lane8 = (uint32_t)lane * 8u, lane16 = (uint32_t)lane * 16u;
    const uint32_t l3 = lstage + ((uint32_t)lane & 3u) * 4u;
    first = rdfirst(first);
    uint32_t used;
    asm volatile(
        "s_mov_b32 s42, %[first]\n"
        "s_mov_b32 s43, 0\n"
        "s_mov_b32 s44, 3\n"
        "s_mov_b32 s48, 0\n"
        "s_mov_b32 s50, s42\n s_mov_b32 s51, 0\n s_lshl_b64 s[50:51], s[50:51], 3\n"
        "s_add_u32 s50, s50, %[tlo]\n s_addc_u32 s51, s51, %[thi]\n"
        "global_load_dwordx2 v[34:35], %[lane8], s[50:51]\n"
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
        "s_branch L_body_%=\n"
        "L_next_%=:\n"
        "s_waitcnt vmcnt(4)\n"                             /* the prefetched block; the four stores behind it stay in flight */
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
        /* lhs / (+-2^k) is lhs * (+-2^-k), the same real number rounded once either way: DIV_LHS_IMM by a power of
         * two whose reciprocal is a normal number becomes MUL_LHS_IMM (11 instructions less per clause; a third of
         * bear's divisions) */
        "v_bfe_u32 v40, v35, 23, 8\n"                      /* exponent of the immediate */
        "v_add_u32 v39, -1, v40\n"
        "v_cmp_gt_u32 vcc, 253, v39\n"                     /* 1 <= exponent <= 253 */
        "v_and_b32 v39, 0x7fffff, v35\n"                   /* its mantissa */
        "s_mov_b64 s[64:65], vcc\n"
        "v_cmp_eq_u32 vcc, 0, v39\n"
        "s_and_b64 s[64:65], s[64:65], vcc\n"
        "v_cmp_eq_u32 vcc, 24, v38\n"                      /* DIV_LHS_IMM */
        "s_and_b64 vcc, vcc, s[64:65]\n"
        "v_sub_u32 v40, 254, v40\n"
        "v_and_b32 v39, 0x80000000, v35\n"
        "v_lshl_or_b32 v39, v40, 23, v39\n"                /* +-2^-k */
        "v_cndmask_b32 v35, v35, v39, vcc\n"
        "v_cndmask_b32 v38, v38, 15, vcc\n"                /* MUL_LHS_IMM */
        /* DIV_LHS_IMM by any other constant with 2^-30 <= |c| <= 2^30: row 30, with 1 / c (IEEE division: correctly
         * rounded) as a second literal */
        "v_and_b32 v39, 0x7fffffff, v35\n"
        "v_add_u32 v39, 0xcf800000, v39\n"                 /* |c| bits - 0x30800000 */
        "v_cmp_gt_u32 vcc, 0x1e000001, v39\n"              /* <= 0x4e800000 - 0x30800000 */
        "v_cmp_eq_u32 s[64:65], 24, v38\n"
        "s_and_b64 s[70:71], vcc, s[64:65]\n"              /* lanes that get row 30 */
        "v_mov_b32 v54, 1.0\n"
        "v_div_scale_f32 v39, s[64:65], v35, v35, v54\n"
        "v_rcp_f32 v40, v39\n"
        "v_div_scale_f32 v41, vcc, v54, v35, v54\n"
        "v_fma_f32 v42, -v39, v40, 1.0\n"
        "v_fmac_f32 v40, v42, v40\n"
        "v_mul_f32 v42, v41, v40\n"
        "v_fma_f32 v43, -v39, v42, v41\n"
        "v_fmac_f32 v42, v43, v40\n"
        "v_fma_f32 v39, -v39, v42, v41\n"
        "s_nop 1\n"
        "v_div_fmas_f32 v39, v39, v40, v42\n"
        "v_div_fixup_f32 v55, v39, v35, v54\n"             /* 1 / c */
        "v_cndmask_b32 v38, v38, 30, s[70:71]\n"
        "v_min_u32 v38, 31, v38\n"                         /* anything that is not an opcode: the empty row 31 */
        "v_mul_u32_u24 v38, 208, v38\n"                    /* 208-byte rows */
        "v_add_u32 v38, %[ltab], v38\n"
        "ds_read_b32 v39, v38\n"                           /* meta: dwords | flags << 8 */
        "v_and_b32 v56, 0xffffff00, v34\n"
        "v_add_u32 v56, 0x30303000, v56\n"                 /* registers: 48 + out / lhs / rhs in bytes 1..3 */
        "v_mov_b32 v57, v35\n"                             /* the literal of this clause */
        "s_waitcnt lgkmcnt(0)\n"
        "v_and_b32 v59, 15, v39\n"
        "v_and_b32 v58, 0x100, v39\n"
        "v_cmp_gt_u32 vcc, s46, %[lane]\n"                 /* clauses in front of the terminator */
        "v_cmp_ne_u32 s[64:65], 0, v58\n"                  /* min / max clauses */
        "s_and_b64 s[64:65], s[64:65], vcc\n"
        "v_cndmask_b32 v59, 0, v59, vcc\n"                 /* dwords of this lane */
        "v_mbcnt_lo_u32_b32 v53, s64, 0\n"
        "v_mbcnt_hi_u32_b32 v53, s65, v53\n"
        "v_add_u32 v53, s48, v53\n"
        /* decisions 64..127 (group form): the rows whose compares name the second pair of registers */
        "v_lshrrev_b32 v60, 6, v53\n"
        "v_min_u32 v60, 1, v60\n"                          /* (tile form: any number of min / max clauses, and rows 32.. are rows 17..20 again) */
        "v_and_b32 v53, 63, v53\n"
        "v_cmp_ne_u32 vcc, 0, v60\n"
        "v_lshl_add_u32 v60, v60, 2, 11\n"                 /* row = opcode + 11 + 4 q */
        "s_and_b64 vcc, vcc, s[64:65]\n"
        "v_mul_u32_u24 v60, 208, v60\n"
        "v_cndmask_b32 v60, 0, v60, vcc\n"
        "v_add_u32 v38, v38, v60\n"
        "v_add_u32 v53, 128, v53\n"                        /* its index, as the inline constant the scalar compare takes */
        "v_and_b32 v53, 0xff, v53\n"
        "v_or_b32 v56, v56, v53\n"
        "s_bcnt1_i32_b64 s66, s[64:65]\n"
        "s_add_u32 s48, s48, s66\n"
        /* exclusive prefix sum of the dword counts (0..12) over the lanes, bit by bit */
        "v_and_b32 v53, 1, v59\n v_cmp_ne_u32 s[56:57], 0, v53\n"
        "v_and_b32 v53, 2, v59\n v_cmp_ne_u32 s[58:59], 0, v53\n"
        "v_and_b32 v53, 4, v59\n v_cmp_ne_u32 s[60:61], 0, v53\n"
        "v_and_b32 v53, 8, v59\n v_cmp_ne_u32 s[62:63], 0, v53\n"
        "v_cmp_eq_u32 vcc, 0, v59\n"                       /* lanes without a dword */
        "v_mbcnt_lo_u32_b32 v52, s56, 0\n v_mbcnt_hi_u32_b32 v52, s57, v52\n"
        "v_mbcnt_lo_u32_b32 v53, s58, 0\n v_mbcnt_hi_u32_b32 v53, s59, v53\n"
        "v_lshl_add_u32 v52, v53, 1, v52\n"
        "v_mbcnt_lo_u32_b32 v53, s60, 0\n v_mbcnt_hi_u32_b32 v53, s61, v53\n"
        "v_lshl_add_u32 v52, v53, 2, v52\n"
        "v_mbcnt_lo_u32_b32 v53, s62, 0\n v_mbcnt_hi_u32_b32 v53, s63, v53\n"
        "v_lshl_add_u32 v52, v53, 3, v52\n"
        "v_add_u32 v52, s44, v52\n"
        "v_lshl_add_u32 v52, v52, 2, %[stage]\n"           /* where this lane's first dword goes in the staging buffer */
        "v_mov_b32 v53, %[dump]\n"
        "v_cndmask_b32 v52, v52, v53, vcc\n"
        "v_cmp_lt_u32 s[66:67], 8, v59\n"                  /* lanes with dwords 8..11 / 4..7 */
        "v_cmp_lt_u32 s[68:69], 4, v59\n"
        "s_bcnt1_i32_b64 s56, s[56:57]\n s_bcnt1_i32_b64 s58, s[58:59]\n s_bcnt1_i32_b64 s60, s[60:61]\n s_bcnt1_i32_b64 s62, s[62:63]\n"
        "s_lshl_b32 s58, s58, 1\n s_lshl_b32 s60, s60, 2\n s_lshl_b32 s62, s62, 3\n"
        "s_add_u32 s44, s44, s56\n s_add_u32 s44, s44, s58\n s_add_u32 s44, s44, s60\n s_add_u32 s44, s44, s62\n"      /* dwords now in the buffer */
        /* All of a lane's dwords, last one first: a dword a lane does not have lands on a later lane's place, and
         * that lane's own dword for the place — always one with a smaller number — is written after it (LDS
         * operations of a wavefront happen in order).  Lanes without any write to the dump, so that within one
         * instruction no two lanes share an address. */
        "v_cmp_lt_u32 s[56:57], 12, v59\n"                /* lanes with dwords 12..14 (the scan is done with s[56:63]) */
        "s_nop 0\n"
        "s_cmp_eq_u64 s[56:57], 0\n"
        "s_cbranch_scc1 L_b2_%=\n"
        JIT_BATCH_PATCH(16+144, 12)
        "L_b2_%=:\n"
        "s_cmp_eq_u64 s[66:67], 0\n"
        "s_cbranch_scc1 L_b1_%=\n"
        JIT_BATCH_PATCH(16+96, 8)
        "L_b1_%=:\n"
        "s_cmp_eq_u64 s[68:69], 0\n"
        "s_cbranch_scc1 L_b0_%=\n"
        JIT_BATCH_PATCH(16+48, 4)
        "L_b0_%=:\n"
        JIT_BATCH(16, 0)
        /* what is complete leaves in 16-byte pieces: lane i carries dwords 4 i .. 4 i + 3, 256 + 4 i ..., 512 + 4 i ..., 768 + 4 i ... */
        "s_and_b32 s45, s44, -4\n"                         /* dwords that leave now */
        "v_add_u32 v53, %[stage], %[lane16]\n"
        "s_lshl_b32 s47, s43, 2\n"
        "v_add_u32 v54, s47, %[lane16]\n"                  /* byte offset in the code region */
        "v_lshrrev_b32 v58, 2, %[lane16]\n"                /* 4 i */
        "v_cmp_gt_u32 vcc, s45, v58\n"
        "s_waitcnt lgkmcnt(0)\n"
        "ds_read_b128 v[40:43], v53\n"
        "ds_read_b128 v[44:47], v53 offset:1024\n"
        "ds_read_b128 v[48:51], v53 offset:2048\n"
        "ds_read_b128 v[64:67], v53 offset:3072\n"
        "v_cndmask_b32 v60, %[trash], v54, vcc\n"
        "v_add_u32 v58, 0x100, v58\n"
        "v_cmp_gt_u32 vcc, s45, v58\n"
        "v_add_u32 v54, 0x400, v54\n"
        "v_add_u32 v58, 0x100, v58\n"
        "v_cndmask_b32 v61, %[trash], v54, vcc\n"
        "v_cmp_gt_u32 vcc, s45, v58\n"
        "v_add_u32 v54, 0x400, v54\n"
        "v_add_u32 v58, 0x100, v58\n"
        "v_cndmask_b32 v62, %[trash], v54, vcc\n"
        "v_cmp_gt_u32 vcc, s45, v58\n"
        "v_add_u32 v54, 0x400, v54\n"
        "s_lshl_b32 s47, s45, 2\n"
        "v_cndmask_b32 v63, %[trash], v54, vcc\n"
        /* the 0..3 dwords that stay move to the front */
        "v_add_u32 v58, s47, %[l3]\n"
        "ds_read_b32 v58, v58\n"
        "s_waitcnt lgkmcnt(4)\n"
        "global_store_dwordx4 v60, v[40:43], %[code]\n"
        "s_waitcnt lgkmcnt(3)\n"
        "global_store_dwordx4 v61, v[44:47], %[code]\n"
        "s_waitcnt lgkmcnt(2)\n"
        "global_store_dwordx4 v62, v[48:51], %[code]\n"
        "s_waitcnt lgkmcnt(1)\n"
        "global_store_dwordx4 v63, v[64:67], %[code]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "ds_write_b32 %[l3], v58\n"
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
        "v_mov_b32 v53, s57\n"
        "s_waitcnt lgkmcnt(0)\n"
        "ds_write_b32 v53, v40\n"
        "ds_write_b32 v53, v41 offset:4\n"
        "s_add_u32 s44, s44, 2\n"                          /* 2..5 dwords: one per lane */
        "v_lshrrev_b32 v58, 2, %[lane16]\n"                /* 4 i */
        "v_add_u32 v53, %[stage], v58\n"
        "s_lshl_b32 s47, s43, 2\n"
        "v_add_u32 v54, s47, v58\n"
        "s_lshl_b32 s45, s44, 2\n"
        "v_cmp_gt_u32 vcc, s45, v58\n"
        "s_waitcnt lgkmcnt(0)\n"
        "ds_read_b32 v48, v53\n"
        "v_cndmask_b32 v60, %[trash], v54, vcc\n"
        "s_waitcnt lgkmcnt(0)\n"
        "global_store_dword v60, v48, %[code]\n"
        "s_add_u32 %[used], s43, s44\n"
        : [used] "=&s"(used)
        : [first] "s"(first), [tlo] "s"(tlo), [thi] "s"(thi), [code] "s"(cbase), [ltab] "s"(ltab), [stage] "s"(lstage), [dump] "s"(ldump),
          [lane] "v"((uint32_t)lane), [lane8] "v"(lane8), [lane16] "v"(lane16), [l3] "v"(l3), [trash] "v"(trash_off)
        : "memory", "vcc", "scc", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58",
          "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42",
          "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67");
    return used;
}
#undef JIT_BATCH
#undef JIT_BATCH_PATCH

/* Ask for the tape of a tile to be brought near (into the L2) long before it is translated: a tape is a linked
 * list of 64-word chunks and following it costs one trip to memory per chunk unless the chunks are already in the
 * cache.  Where they are can be said without reading them: the tile stages write a tape from its end, chunk after
 * chunk at rising addresses (kernels.hip: out_index += MPR_SUBTAPE_CHUNK), so reading forward from the head means
 * falling addresses, chunk starts are congruent to the root tape's length modulo 64, and the root tape itself is
 * contiguous.  Three LDS-DMA loads of 1 KB (no destination registers; the bytes land in a part of LDS nobody reads)
 * cover six chunks; a wrong guess costs nothing but those loads. */
DEV void jit_prefetch_tape(const uint64_t* __restrict__ tro, int tape, int tape_len, uint32_t lds_dump, int lane)
{
    int w0, step;                                      /* first word of the highest 1 KB piece */
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
        asm volatile("s_mov_b32 m0, %2\n s_nop 0\n global_load_lds_dwordx4 %0, %1\n" : : "v"(lane16), "s"(addr), "s"(m0v) : "memory");
    }
}

/* Run the code at `code` on (vx, vy, vz).  fresh bit 0: the region was just rewritten by this wavefront (invalidate the
 * instruction cache), bit 1: three prefetch loads are in flight behind the code's stores.  decisions: group form, lane i
 * holds the child's `chose lhs` / `chose rhs` bits at the tape's i-th and (64 + i)-th min / max clause.  NS: slots the kernel provides registers
 * for.  The register lists are what the generated code, the routines and the compiled leaf routines may touch. */
#define JIT_V10(a) "v" #a "0", "v" #a "1", "v" #a "2", "v" #a "3", "v" #a "4", "v" #a "5", "v" #a "6", "v" #a "7", "v" #a "8", "v" #a "9"
#define JIT_CLOBBER_BASE                                                                                                   \
    "memory", "vcc", "scc", "s0", "s1", "s2", "s3", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13", "s14", "s15", "s30", "s31", "s40", "s41", "s42", \
        "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60",      \
        "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s90", "s91", "s92",      \
        "s76", "s77", "s78", "s79", "s93", "s94", "s95", "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39",       \
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
    JIT_ROUTINE_ADDR(70, 71, "L_divc")                                                                 \
    "s_mov_b32 s90, 0x260\n"                            /* class mask of the square root */           \
    /* group form: the child's min / max decisions; lane i brings those of the tape's i-th and (64 + i)-th min / max:  \
     * bit 0 chose lhs, bit 1 chose rhs, bits 2 / 3 the same for 64 + i (s[50:51] was the base of the addresses above) */ \
    "v_and_b32 v35, 1, %[dec]\n v_and_b32 v36, 2, %[dec]\n v_and_b32 v37, 4, %[dec]\n v_and_b32 v38, 8, %[dec]\n"     \
    "v_cmp_ne_u32 s[76:77], 0, v35\n v_cmp_ne_u32 s[78:79], 0, v36\n"                                   \
    "v_cmp_ne_u32 s[48:49], 0, v37\n v_cmp_ne_u32 s[50:51], 0, v38\n"                                   \
    "v_mov_b32 v32, %[vx]\n v_mov_b32 v33, %[vy]\n v_mov_b32 v34, %[vz]\n"                             \
    "v_mov_b32 v7, 0x2ff\n"                           /* class mask of the inline constant division */  \
    "v_mov_b32 v46, 0\n v_mov_b32 v47, 1.0\n"          /* L_sin / L_cos: (argument, cosine) seen last: cos(+0) = 1 */ \
    "s_mov_b32 s74, %[clo]\n s_mov_b32 s75, %[chi]\n"                                                  \
    "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 9, 1), 0\n" /* MODE.IEEE off while generated code runs (see jt::minmax) */ \
    "s_swappc_b64 s[72:73], s[74:75]\n"                                                                \
    "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 9, 1), 1\n"                                                 \
    "v_mov_b32 %[res], v37\n"                                                                          \
    "s_branch L_end_%=\n"                                                                              \
    /* The general case of a division by a constant (row 30 of the table: some lane's x * x is not a positive normal   \
     * number).  x is in v35; the return address is that of the row's sixth dword, and the constant — negated — is the   \
     * literal three dwords on.  v37 = x / c by the general sequence; back to the row's last instruction                   \
     * (out = v39 * y + v37) with v39 = a zero whose sign times y's is the quotient's, so that a quotient of -0 stays -0.  \
     * (The code was written a moment ago by this workgroup's translator: the scalar cache may hold the region's           \
     * previous contents.) */                                                                                              \
    "L_divc_%=:\n"                                                                                     \
    "s_dcache_inv\n"                                                                                   \
    "s_load_dword s40, s[30:31], 0xc\n"                                                                \
    "s_waitcnt lgkmcnt(0)\n"                                                                           \
    "s_xor_b32 s40, s40, 0x80000000\n"                                                                 \
    "v_mov_b32 v36, s40\n"                                                                             \
    MPR_ASM_DIV_BODY                                                                                   \
    "v_xor_b32 v39, v37, v36\n"                                                                        \
    "v_and_b32 v39, 0x80000000, v39\n"                                                                 \
    "s_add_u32 s30, s30, 32\n"                                                                         \
    "s_addc_u32 s31, s31, 0\n"                                                                         \
    "s_setpc_b64 s[30:31]\n"                                                                           \
    "L_div_%=:\n" MPR_ASM_DIV_BODY "s_setpc_b64 s[30:31]\n"                                            \
    "L_sqrt_%=:\n" MPR_ASM_SQRT_BODY "s_setpc_b64 s[30:31]\n" MPR_ASM_SQRT_TAIL                        \
    "L_exp_%=:\n" MPR_ASM_EXP_BODY "s_setpc_b64 s[30:31]\n" MPR_ASM_EXP_TAIL                           \
    "L_log_%=:\n" MPR_ASM_LOG_BODY "s_setpc_b64 s[30:31]\n" MPR_ASM_LOG_TAIL                           \
    /* sin and cos: one routine computes both (MPR_ASM_SINCOS_BODY: v37 = sin, v36 = cos), and models ask for both of the same  \
     * argument a few clauses apart (bear: three times).  v46 / v47 — temporaries of this routine alone — keep (argument,      \
     * its cosine) between calls: a cosine of the bits just seen is a compare and a move instead of 61 issue units.  The pair is \
     * always a true one (set with the routine's own results; (0, 1) at the start of a run), so a hit is the routine's value. */ \
    "L_sin_%=:\n" MPR_ASM_SINCOS_BODY "v_mov_b32 v46, v35\n v_mov_b32 v47, v36\n s_setpc_b64 s[30:31]\n"     \
    "L_cos_%=:\n"                                                                                      \
    "v_cmp_ne_u32 vcc, v35, v46\n"                                                                      \
    "s_cbranch_vccnz L_cosmiss_%=\n"                                                                    \
    "v_mov_b32 v37, v47\n"                                                                              \
    "s_setpc_b64 s[30:31]\n"                                                                            \
    "L_cosmiss_%=:\n" MPR_ASM_SINCOS_BODY "v_mov_b32 v37, v36\n v_mov_b32 v46, v35\n v_mov_b32 v47, v36\n s_setpc_b64 s[30:31]\n" \
    "L_end_%=:\n"

template <int NS>
DEV float jit_run(const uint32_t* code, uint32_t fresh, float vx, float vy, float vz, uint32_t decisions = 0)
{
    const uint32_t clo = rdfirst((uint32_t)(uintptr_t)code), chi = rdfirst((uint32_t)((uintptr_t)code >> 32));
    fresh = rdfirst(fresh);
    float res;
#define JIT_OPERANDS : [res] "=&v"(res) : [vx] "v"(vx), [vy] "v"(vy), [vz] "v"(vz), [clo] "s"(clo), [chi] "s"(chi), [fresh] "s"(fresh), [dec] "v"(decisions)
    if constexpr (NS <= 24) {
        asm volatile(JIT_ASM_TEXT JIT_OPERANDS : JIT_CLOBBER_BASE);
    } else if constexpr (NS <= 40) {
        asm volatile(JIT_ASM_TEXT JIT_OPERANDS : JIT_CLOBBER_BASE, JIT_CLOBBER_40);
    } else if constexpr (NS <= 96) {
        asm volatile(JIT_ASM_TEXT JIT_OPERANDS : JIT_CLOBBER_BASE, JIT_CLOBBER_96);
    } else {
        asm volatile(JIT_ASM_TEXT JIT_OPERANDS : JIT_CLOBBER_BASE, JIT_CLOBBER_192);
    }
#undef JIT_OPERANDS
    return res;
}

/* position of a voxel / pixel of a smallest tile, the skip test, the write of the result */
template <int DIM>
struct JitVoxel {
    int px, py, pz;
    bool skip;
    float vx, vy, vz;
    /* returns false when every lane is hidden */
    DEV bool setup(const VoxelArgs& a, int position, int lane)
    {
        constexpr int SUB = (DIM == 3) ? 4 : 8;
        const int S = a.tps * SUB;
        const int4_ pos = unpack(position, a.tps);
        const int4_ sub = unpack(lane, SUB);
        px = pos.x * SUB + sub.x;
        py = pos.y * SUB + sub.y;
        pz = (DIM == 3) ? pos.z * 4 + sub.z : 0;
        skip = false;
        if (DIM == 3) {
            /* reference :852-864: the thread owning (pz_low, pz_low + 2) leaves when image >= pz_low + 2;
             * read past this CU's vector L1: the heights other tiles of the column have written so far */
            const int pz_low = pos.z * 4 + (sub.z & 1);
            skip = __hip_atomic_load(&a.image[px + py * S], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= pz_low + 2;
            if (ballot(!skip) == 0) return false;
        }
        const float size_recip = 1.0f / (float)(unsigned)S;
        const float fx = ((px + 0.5f) * size_recip - 0.5f) * 2.0f;
        const float fy = ((py + 0.5f) * size_recip - 0.5f) * 2.0f;
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
        return true;
    }
    DEV void finish(const VoxelArgs& a, float res) const
    {
        constexpr int SUB = (DIM == 3) ? 4 : 8;
        const int S = a.tps * SUB;
        if (!skip && res < 0.0f) {
            if (DIM == 3) {
                int* p = &a.image[px + py * S];
                if (*p < pz) atomicMax(p, pz);
            } else {
                a.image[px + py * S] = 1;
            }
        }
    }
};

struct JitVoxelArgs {
    VoxelArgs v;               /* tile form: the smallest tiles; group form: the LAST tile stage's list, after its compaction */
    uint32_t* code;            /* executable; one region per wavefront (tile form) / workgroup (group form) */
    uint32_t region_dwords;    /* code (group form: `slots` pieces of slot_dwords), then 320 dwords the translator dumps into */
    uint32_t slot_dwords, slots;
    int tape_len;              /* words of the root tape at pool[0] */
    const GroupInfo* groups;   /* group form: the tape each group of 64 siblings walked, and their min / max decisions */
    const ulonglong2* choice_masks;
    int choice_cap;
    int* group_counter;        /* group form: the next entry of group_list to hand out (zero at the start of the frame) */
    const int* group_list;     /* group form: the groups with a surviving tile, in list order; [number of groups] = how many */
    int always_invalidate;     /* group form: s_icache_inv after every translation, not once per trip round the ring of code slots
                                * (context.hip: slots closer than the validated 4 KB, a device other than gfx950, MPR_VOXEL_JIT=3) */
};

/* ---- tile form: a wavefront per smallest tile, each with its own tape ------------------------------------------ */
template <int DIM, int NS>
__global__ void __launch_bounds__(64)
k_eval_voxels_jit(JitVoxelArgs j)
{
    __shared__ __attribute__((aligned(64))) uint32_t lds[JIT_LDS_DWORDS];
    const VoxelArgs& a = j.v;
    const int lane = threadIdx.x;
    jit_load_table(lds, lane, 64, false);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    uint32_t* const code = j.code + (size_t)blockIdx.x * j.region_dwords;
    const uint32_t trash_off = (j.region_dwords - 320) * 4u + (uint32_t)lane * 16u;
    const uint64_t* __restrict__ const tro = a.tape_ro;
    const uint64_t head0 = tro[0];
    int cached_tape = -1;

    for (int run = blockIdx.x; run * JIT_RUN < a.count; run += gridDim.x) {
        for (int k = 0; k < JIT_RUN; ++k) {
            const int tile_index = run * JIT_RUN + k;
            if (tile_index >= a.count) break;
            const int position = __builtin_amdgcn_readfirstlane(a.tiles[tile_index].position);
            const int tape = __builtin_amdgcn_readfirstlane(a.tiles[tile_index].tape);
            JitVoxel<DIM> vox;
            if (!vox.setup(a, position, lane)) continue;
            uint32_t fresh = 0;
            if (tape != cached_tape) {
                (void)jit_translate(tro, (uint32_t)(tape + 1), head0, code, trash_off, lds, lane);
                cached_tape = tape;
                fresh = 1;
            }
            {
                /* the next tile's tape travels while this one runs */
                int nt = tile_index + 1;
                if (k + 1 == JIT_RUN) nt = (run + (int)gridDim.x) * JIT_RUN;
                int ntape = tape;
                if (nt < a.count) ntape = __builtin_amdgcn_readfirstlane(a.tiles[nt].tape);
                jit_prefetch_tape(tro, ntape, j.tape_len, (uint32_t)(uintptr_t)(lds + JIT_LDS_PFDUMP), lane);
                fresh |= 2u;                              /* three loads are in flight behind the code's stores */
            }
            const float res = jit_run<NS>(code, fresh, vox.vx, vox.vy, vox.vz);
            vox.finish(a, res);
        }
    }
}

/* ---- group form: a workgroup per group of 64 sibling tiles, one translation, every surviving child runs it ---- */
constexpr int JIT_GROUP_WAVES = 4;
template <int DIM, int NS>
/* registers: slots + 48; the bound keeps the compiler's own values from costing a wavefront of occupancy */
__global__ void __launch_bounds__(64 * JIT_GROUP_WAVES, NS <= 24 ? 6 : NS <= 40 ? 5 : NS <= 96 ? 3 : 2)
k_eval_voxels_jit_groups(JitVoxelArgs j)
{
    __shared__ __attribute__((aligned(64))) uint32_t lds[JIT_LDS_DWORDS];
    const VoxelArgs& a = j.v;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    jit_load_table(lds, (int)threadIdx.x, 64 * JIT_GROUP_WAVES, true);
    __syncthreads();
    /* The workgroup's region is a ring of code slots.  A group's tape is translated into the slot after the one
     * used last, which nothing has been fetched from since the ring was last entered at slot 0 — no instruction
     * cache can hold lines of it — so the invalidate, which throws out everybody's code in the CU pair's cache
     * (measured: with one per group the instruction fetch is the bottleneck again), is needed once per trip round
     * the ring.  Slots are 4 KB apart beyond the longest code: the sequential instruction prefetch that runs past
     * the end of one slot's code must not reach the next slot before that is written (256 bytes were not enough,
     * 1 KB was). */
    uint32_t* const region = j.code + (size_t)blockIdx.x * j.region_dwords;
    __shared__ int next_child_lds;
    __shared__ ulonglong2 more_masks[64];
    int* const next_child = &next_child_lds;
    uint32_t slot = 0;
    bool first_group = true;
    const uint64_t* __restrict__ const tro = a.tape_ro;
    const uint64_t head0 = tro[0];
    const int ngroups = (a.count + 63) / 64;

    __shared__ int next_group;
    /* the groups that still have a tile (j.group_list, in list order: front to back) are handed out one at a time as
     * workgroups come free: their cost varies with the number of surviving children — a fixed stride left a third of the
     * kernel's duration to a tail of long workgroups — and the order is what lets the groups behind a surface find it
     * already drawn (handing out four at a time cost bear 30 %) */
    const int nlisted = j.group_list[ngroups];
    for (;;) {
        int g, position = -1, gtape, nch;
        uint64_t alive;
        if (threadIdx.x == 0) next_group = atomicAdd(j.group_counter, 1);
        __syncthreads();
        const int r = next_group;
        __syncthreads();
        if (r >= nlisted) break;
        g = j.group_list[r];
        /* the 64 siblings as the last compaction left them: position -1 = empty, filled or hidden */
        const int idx = g * 64 + lane;
        if (idx < a.count) position = a.tiles[idx].position;
        alive = ballot(position != -1);
        if (alive == 0) continue;                         /* the same for every wavefront of the workgroup */
        const GroupInfo gi = j.groups[g];
        gtape = __builtin_amdgcn_readfirstlane(gi.tape);
        nch = __builtin_amdgcn_readfirstlane(gi.nchoices);
        if (!first_group) slot = slot + 1 == j.slots ? 0 : slot + 1;
        first_group = false;
        uint32_t* const code = region + (size_t)slot * j.slot_dwords;
        if (wave == 0) {
            if (lane == 0) *next_child = 0;
            const uint32_t trash_off = (uint32_t)((j.region_dwords - 320 - slot * j.slot_dwords) * 4u) + (uint32_t)lane * 16u;
            (void)jit_translate(tro, (uint32_t)(gtape + 1), head0, code, trash_off, lds, lane);
            if (slot == 0 || j.always_invalidate) asm volatile("s_waitcnt vmcnt(0)\n s_icache_inv\n s_nop 7\n s_nop 7\n" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n" ::: "memory");
        }
        /* lane i: the decisions of the group's 64 tiles at its i-th min / max */
        ulonglong2 m = make_ulonglong2(0ull, 0ull);
        auto load_mask = [&](int i) { return j.choice_masks[(size_t)g * j.choice_cap + i]; };
        if (lane < nch) m = load_mask(lane);
        if (nch > 64 && threadIdx.x < 64) {
            /* decisions 64..127 wait in LDS (registers held across the children's loop would cost a wave of occupancy) */
            const int i = 64 + (int)threadIdx.x;
            more_masks[threadIdx.x] = i < nch ? load_mask(i) : make_ulonglong2(0ull, 0ull);
        }
        __syncthreads();
        /* front to back: children with the larger z first (lane = x + 4 y + 16 z); a wavefront takes the next child
         * when it is done with its last one (hidden children cost nothing, the others differ little) */
        const int nalive = __popcll(alive);
        for (;;) {
            int k = 0;
            if (lane == 0) k = atomicAdd(next_child, 1);
            k = __builtin_amdgcn_readfirstlane(k);
            if (k >= nalive) break;
            /* the k-th set bit of `alive` from the top */
            uint64_t rest = alive;
            for (int skip = 0; skip < k; ++skip) rest &= ~(1ull << (63 - __builtin_clzll(rest)));
            const int c = 63 - __builtin_clzll(rest);
            const int cpos = (int)rdlane((uint32_t)position, (uint32_t)c);
            JitVoxel<DIM> vox;
            if (!vox.setup(a, cpos, lane)) continue;
            uint32_t decisions = (uint32_t)((m.x >> c) & 1ull) | ((uint32_t)((m.y >> c) & 1ull) << 1);
            if (nch > 64) {
                const ulonglong2 m1 = more_masks[lane];
                decisions |= ((uint32_t)((m1.x >> c) & 1ull) << 2) | ((uint32_t)((m1.y >> c) & 1ull) << 3);
            }
            const float res = jit_run<NS>(code, 0u, vox.vx, vox.vy, vox.vz, decisions);
            vox.finish(a, res);
        }
        __syncthreads();                                  /* the region is rewritten for the next group */
    }
}

/* ---- the pass on the ROOT tape's host-generated code (voxel_gen.hpp) -------------------------------------------
 * For tapes the host generates code for (at most 24 slots and 64 min / max clauses) in frames whose tile stages kept
 * their tiles' decisions as bits over the root tape's min / max clauses (TileStageArgs::gen_decisions): a smallest
 * tile's own tape is the root tape with the decisions of the 16^3 tile above it (its record) and its own (the group's
 * masks, numbered by the clauses the parent's tape keeps) applied, so every tile runs the ONE piece of code there is
 * per tape — nothing is translated on the device, no code ring, no instruction-cache invalidates, and dead clauses are
 * jumped over by the code's own scalar branches on the decisions. */
#define VG_ASM_TEXT                                                                                    \
    "s_getpc_b64 s[50:51]\n"                                                                           \
    "L_pc_%=:\n"                                                                                       \
    JIT_ROUTINE_ADDR(52, 53, "L_div") JIT_ROUTINE_ADDR(54, 55, "L_sqrt") JIT_ROUTINE_ADDR(56, 57, "L_exp")  \
    JIT_ROUTINE_ADDR(58, 59, "L_log") JIT_ROUTINE_ADDR(60, 61, "L_sin") JIT_ROUTINE_ADDR(62, 63, "L_cos")   \
    JIT_LEAF_ADDR(64, 65, "mpr_fj_asin") JIT_LEAF_ADDR(66, 67, "mpr_fj_acos") JIT_LEAF_ADDR(68, 69, "mpr_fj_atan") \
    "s_mov_b32 s90, 0x260\n"                            /* class mask of the square root's slow path */ \
    "s_mov_b32 s80, 0x42ae0000\n"                       /* 87.0: exp without special cases (voxel_gen.hpp) */ \
    "s_movk_i32 s81, 0x100\n"                           /* class mask: positive normal (log) */         \
    "s_mov_b32 s76, %[dl0]\n s_mov_b32 s77, %[dl1]\n s_mov_b32 s78, %[dr0]\n s_mov_b32 s79, %[dr1]\n"   \
    "v_mov_b32 v32, %[vx]\n v_mov_b32 v33, %[vy]\n v_mov_b32 v34, %[vz]\n"                             \
    "v_mov_b32 v7, 0x2ff\n"                             /* class mask of the inline constant division */ \
    "v_mov_b32 v5, 0x39506967\n v_mov_b32 v6, 0x3d9021bb\n"   /* leading coefficients of the exp / log polynomials (voxel_gen.hpp) */ \
    "v_mov_b32 v46, 0\n v_mov_b32 v47, 1.0\n"           /* L_sin / L_cos: (argument, cosine) seen last */ \
    "s_mov_b32 s74, %[clo]\n s_mov_b32 s75, %[chi]\n"                                                  \
    "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 9, 1), 0\n"  /* MODE.IEEE off while generated code runs (see jt::minmax) */ \
    "s_swappc_b64 s[72:73], s[74:75]\n"                                                                \
    "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 9, 1), 1\n"                                                 \
    "v_mov_b32 %[res], v37\n"                                                                          \
    "s_branch L_end_%=\n"                                                                              \
    "L_div_%=:\n" MPR_ASM_DIV_BODY "s_setpc_b64 s[30:31]\n"                                            \
    "L_sqrt_%=:\n" MPR_ASM_SQRT_BODY "s_setpc_b64 s[30:31]\n" MPR_ASM_SQRT_TAIL                        \
    "L_exp_%=:\n" MPR_ASM_EXP_BODY "s_setpc_b64 s[30:31]\n" MPR_ASM_EXP_TAIL                           \
    "L_log_%=:\n" MPR_ASM_LOG_BODY "s_setpc_b64 s[30:31]\n" MPR_ASM_LOG_TAIL                           \
    "L_sin_%=:\n" MPR_ASM_SINCOS_BODY "v_mov_b32 v46, v35\n v_mov_b32 v47, v36\n s_setpc_b64 s[30:31]\n"     \
    "L_cos_%=:\n"                                                                                      \
    "v_cmp_ne_u32 vcc, v35, v46\n"                                                                      \
    "s_cbranch_vccnz L_cosmiss_%=\n"                                                                    \
    "v_mov_b32 v37, v47\n"                                                                              \
    "s_setpc_b64 s[30:31]\n"                                                                            \
    "L_cosmiss_%=:\n" MPR_ASM_SINCOS_BODY "v_mov_b32 v37, v36\n v_mov_b32 v46, v35\n v_mov_b32 v47, v36\n s_setpc_b64 s[30:31]\n" \
    "L_end_%=:\n"

/* dl / dr: the tile's decisions, bit k = min / max clause k of the root tape decided for the lhs / rhs (wave-uniform) */
DEV float vox_gen_run(const uint32_t* code, float vx, float vy, float vz, uint64_t dl, uint64_t dr)
{
    const uint32_t clo = rdfirst((uint32_t)(uintptr_t)code), chi = rdfirst((uint32_t)((uintptr_t)code >> 32));
    const uint32_t dl0 = rdfirst((uint32_t)dl), dl1 = rdfirst((uint32_t)(dl >> 32)), dr0 = rdfirst((uint32_t)dr), dr1 = rdfirst((uint32_t)(dr >> 32));
    float res;
    asm volatile(VG_ASM_TEXT
                 : [res] "=&v"(res)
                 : [vx] "v"(vx), [vy] "v"(vy), [vz] "v"(vz), [clo] "s"(clo), [chi] "s"(chi), [dl0] "s"(dl0), [dl1] "s"(dl1), [dr0] "s"(dr0), [dr1] "s"(dr1)
                 : JIT_CLOBBER_BASE, "s80", "s81");
    return res;
}

struct GenVoxelArgs {
    VoxelArgs v;                       /* tiles / count: the smallest tiles the last compaction left, front to back */
    const uint32_t* code;              /* the root tape's float walk (executable memory) */
    const int* source;                 /* per smallest tile: its index in the last tile stage's list (group = / 64, child = % 64) */
    const GroupInfo* groups;
    const ulonglong2* choice_masks;
    int choice_cap;
    int* tile_counter;                 /* VG_LISTS counters, VG_COUNTER_STRIDE ints apart, zero when the kernel starts */
    const unsigned long long* parent_records;   /* records of the tiles of the stage above (GEN_RECORD_U64 words each) */
    int nchoices;                      /* min / max clauses of the root tape */
    int run;                           /* consecutive tiles a wavefront takes per atomic */
    int* walked;                       /* development: += tiles walked (null: not counted) */
    const unsigned char* skip;         /* null, or per tile of the last tile stage's list (TileStageArgs::tight_skip): 1 = provably empty, not walked;
                                        * 2 = provably filled, drawn without a walk; 3 = the same (the segments' launch draws those: k_compact_footprints) */
};

/* A wavefront per smallest tile, in list order (front to back: the tiles behind a surface find it drawn).  Nothing is shared
 * between the tiles of a group any more (the group form's workgroup translated one tape for its 64 children and waited for the
 * slowest of its four wavefronts): no barrier, no idle wavefront at the end of a group, and a rank with an eighth of the tiles
 * (multi-GPU) ends when its last handful of tiles does, not its last group.
 * Handing out: runs of VG_RUN consecutive tiles, one atomic each — but not on ONE word: same-address atomics serialise at about
 * 12 ns each on this part, and one per tile (0.58 M for bear 1024^3) made the counter the kernel: 6.7 ms instead of 0.8.  The
 * runs are dealt round robin to VG_LISTS counters a cache line or more apart; a wavefront works through the counter of its own
 * number first and helps with the others when that one runs dry. */
constexpr int VG_RUN = 4, VG_LISTS = 8, VG_COUNTER_STRIDE = 64;      /* (ints between counters; VG_RUN: the default of GenVoxelArgs::run) */
template <int DIM>
__global__ void __launch_bounds__(64, 6)         /* (7 waves per SIMD would take the SGPRs the routines name away: s90..s95) */
k_eval_voxels_gen(GenVoxelArgs j)
{
    const VoxelArgs& a = j.v;
    const int lane = threadIdx.x;
    const unsigned long long all = j.nchoices >= 64 ? ~0ull : ((1ull << j.nchoices) - 1ull);
    const int run_len = j.run;
    const int nruns = (a.count + run_len - 1) / run_len;
    int nwalked = 0;
    for (int turn = 0; turn < VG_LISTS; ++turn) {
        const int list = (int)((blockIdx.x + (unsigned)turn) % VG_LISTS);
        const int list_runs = (nruns - list + VG_LISTS - 1) / VG_LISTS;
        for (;;) {
            int q = 0;
            if (lane == 0) {
                /* a wavefront that has come to HELP with another list looks before it claims: when the frame ends every wavefront asks
                 * every list, and 7168 claims on a word with nothing left take their 9 ns one after the other — a third of a rank's
                 * float pass in an 8-way deal (round 5; a look before EVERY claim cost the hand-outs of the whole frame more) */
                if (turn > 0 && __hip_atomic_load(j.tile_counter + list * VG_COUNTER_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= list_runs) q = list_runs;
                else q = atomicAdd(j.tile_counter + list * VG_COUNTER_STRIDE, 1);
            }
            const int run = __builtin_amdgcn_readfirstlane(q) * VG_LISTS + list;
            if (run >= nruns) break;
            for (int t = run * run_len; t < min(run * run_len + run_len, a.count); ++t) {
                const int position = __builtin_amdgcn_readfirstlane(a.tiles[t].position);
                const int src = __builtin_amdgcn_readfirstlane(j.source[t]);
                const int verdict = j.skip ? __builtin_amdgcn_readfirstlane((int)j.skip[src]) : 0;
                if (verdict == 1) continue;
                JitVoxel<DIM> vox;
                if (!vox.setup(a, position, lane)) continue;
                if (verdict >= 2) {          /* (3: drawn from the second verdict's image where the segments' launch ran; once more does no harm) */
                    vox.finish(a, -1.0f);
                    continue;
                }
                const int g = src >> 6, c = src & 63;
                const GroupInfo gi = j.groups[g];
                /* what the tile above decided (and everything above it), and the min / max clauses its tape keeps: the group's
                 * masks are numbered by those */
                unsigned long long L = 0, R = 0, K = all;
                if (__builtin_amdgcn_readfirstlane(gi.tape) != 0) {
                    const unsigned long long* const rec = j.parent_records + (size_t)__builtin_amdgcn_readfirstlane(gi.parent) * GEN_RECORD_U64;
                    L = rfl64(rec[0]);
                    R = rfl64(rec[1]);
                    K = rfl64(rec[2]);
                }
                /* lane k: root clause k is the i-th clause that tape keeps; its mask's bit c = this tile's decision there */
                const bool kept = (K >> lane) & 1ull;
                const int i = __popcll(K & ((1ull << lane) - 1ull));
                ulonglong2 mk = make_ulonglong2(0ull, 0ull);
                if (kept && i < __builtin_amdgcn_readfirstlane(gi.nchoices)) mk = j.choice_masks[(size_t)g * j.choice_cap + i];
                const uint64_t dl = L | ballot((mk.x >> c) & 1ull), dr = R | ballot((mk.y >> c) & 1ull);
                const float res = vox_gen_run(j.code, vox.vx, vox.vy, vox.vz, dl, dr);
                vox.finish(a, res);
                ++nwalked;
            }
        }
    }
    if (j.walked && lane == 0 && nwalked) atomicAdd(j.walked + (blockIdx.x & 31) * 32, nwalked);
}

int voxel_gen_counter_ints() { return VG_LISTS * VG_COUNTER_STRIDE; }
int voxel_gen_counter_lists() { return VG_LISTS; }
int voxel_gen_grid(int dim, int cus)
{
    int per_cu = 0;
    const hipError_t e = dim == 3 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_eval_voxels_gen<3>, 64, 0)
                                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_eval_voxels_gen<2>, 64, 0);
    if (e != hipSuccess || per_cu <= 0) per_cu = 16;
    return per_cu * cus;
}
void launch_eval_voxels_gen(hipStream_t s, int dim, const VoxelArgs& a, const uint32_t* code, int grid, const int* source, const GroupInfo* groups,
                            const ulonglong2* choice_masks, int choice_cap, int* tile_counter, const unsigned long long* parent_records,
                            int nchoices, int run, int* walked, const unsigned char* skip)
{
    if (a.count <= 0) return;
    GenVoxelArgs j;
    j.walked = walked;
    j.skip = skip;
    j.run = run > 0 ? run : VG_RUN;
    j.v = a;
    j.code = code;
    j.source = source;
    j.groups = groups;
    j.choice_masks = choice_masks;
    j.choice_cap = choice_cap;
    j.tile_counter = tile_counter;
    j.parent_records = parent_records;
    j.nchoices = nchoices;
    const dim3 g(std::min(grid, a.count)), b(64);
    if (dim == 3) hipLaunchKernelGGL(k_eval_voxels_gen<3>, g, b, 0, s, j);
    else hipLaunchKernelGGL(k_eval_voxels_gen<2>, g, b, 0, s, j);
}

/* ---- the same pass by FOOTPRINT SEGMENTS (round 6; kernels.hip: k_compact_footprints) -----------------------------------
 * A segment = the (up to four) surviving tiles over one 4 x 4 footprint of pixels inside one block of 64 siblings, nearest first.
 * One wavefront walks them in that order and stops at the first hidden one; the group's record and masks are fetched once per
 * segment.  v.tiles: the LAST TILE STAGE's list (an item's low 28 bits = block * 64 + footprint, the top 4 = which z). */
struct FpVoxelArgs {
    VoxelArgs v;
    const uint32_t* code;
    const unsigned* items;
    const int* meta;                   /* [1] = number of segments */
    const GroupInfo* groups;
    const ulonglong2* choice_masks;
    int choice_cap;
    int* counter;                      /* VG_LISTS counters, VG_COUNTER_STRIDE ints apart, zero when the kernel starts */
    const unsigned long long* parent_records;
    int nchoices;
    int run;                           /* consecutive segments a wavefront takes per atomic */
    int* walked;                       /* development: += tiles walked (null: not counted) */
    const unsigned char* skip;         /* GenVoxelArgs::skip */
};
template <int DIM>
__global__ void __launch_bounds__(64, 6)
k_eval_voxels_gen_fp(FpVoxelArgs j)
{
    const VoxelArgs& a = j.v;
    const int lane = threadIdx.x;
    const unsigned long long all = j.nchoices >= 64 ? ~0ull : ((1ull << j.nchoices) - 1ull);
    const int run_len = j.run;
    const int total = __builtin_amdgcn_readfirstlane(__hip_atomic_load(j.meta + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const int nruns = (total + run_len - 1) / run_len;
    int nwalked = 0;
    for (int turn = 0; turn < VG_LISTS; ++turn) {
        const int list = (int)((blockIdx.x + (unsigned)turn) % VG_LISTS);
        const int list_runs = (nruns - list + VG_LISTS - 1) / VG_LISTS;
        /* the first run of a wavefront's own list is dealt, not claimed: all the grid's wavefronts asking eight counters at the same moment
         * waited 6 to 9 us for the answer (768 same-address atomics a counter, ~10 ns each).  The counters count from there. */
        const int list_dealt = ((int)gridDim.x - list + VG_LISTS - 1) / VG_LISTS;
        bool dealt = turn == 0;
        for (;;) {
            int q = 0;
            if (dealt) {
                q = (int)blockIdx.x / VG_LISTS;
                dealt = false;
            } else {
                if (lane == 0) {
                    if (turn > 0 && __hip_atomic_load(j.counter + list * VG_COUNTER_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + list_dealt >= list_runs) q = list_runs;
                    else q = atomicAdd(j.counter + list * VG_COUNTER_STRIDE, 1) + list_dealt;
                }
                q = __builtin_amdgcn_readfirstlane(q);
            }
            const int run = q * VG_LISTS + list;
            if (run >= nruns) break;
            for (int t = run * run_len; t < min(run * run_len + run_len, total); ++t) {
                const uint32_t item = __builtin_amdgcn_readfirstlane(j.items[t]);
                const int base = (int)(item & 0x0FFFFFFFu);
                const uint32_t zbits = item >> 28;
                const int g = base >> 6, f = base & 15;
                const GroupInfo gi = j.groups[g];
                unsigned long long L = 0, R = 0, K = all;
                if (__builtin_amdgcn_readfirstlane(gi.tape) != 0) {
                    const unsigned long long* const rec = j.parent_records + (size_t)__builtin_amdgcn_readfirstlane(gi.parent) * GEN_RECORD_U64;
                    L = rfl64(rec[0]);
                    R = rfl64(rec[1]);
                    K = rfl64(rec[2]);
                }
                /* lane k: root clause k is the i-th clause the group's tape keeps; its mask's bit c = tile c's decision there */
                const bool kept = (K >> lane) & 1ull;
                const int i = __popcll(K & ((1ull << lane) - 1ull));
                ulonglong2 mk = make_ulonglong2(0ull, 0ull);
                if (kept && i < __builtin_amdgcn_readfirstlane(gi.nchoices)) mk = j.choice_masks[(size_t)g * j.choice_cap + i];
                for (int z = 3; z >= 0; --z) {
                    if (!((zbits >> z) & 1u)) continue;
                    const int c = z * 16 + f, src = (base & ~63) + c;
                    const int verdict = j.skip ? __builtin_amdgcn_readfirstlane((int)j.skip[src]) : 0;
                    if (verdict == 1) continue;
                    const int position = __builtin_amdgcn_readfirstlane(a.tiles[src].position);
                    if (position < 0) continue;
                    JitVoxel<DIM> vox;
                    if (!vox.setup(a, position, lane)) break;               /* hidden: so is everything behind it */
                    if (verdict >= 2) {          /* (3: drawn from the second verdict's image where the segments' launch ran; once more does no harm) */
                        vox.finish(a, -1.0f);
                        continue;
                    }
                    const uint64_t dl = L | ballot((mk.x >> c) & 1ull), dr = R | ballot((mk.y >> c) & 1ull);
                    const float res = vox_gen_run(j.code, vox.vx, vox.vy, vox.vz, dl, dr);
                    vox.finish(a, res);
                    ++nwalked;
                }
            }
        }
    }
    if (j.walked && lane == 0 && nwalked) atomicAdd(j.walked + (blockIdx.x & 31) * 32, nwalked);
}
int voxel_gen_fp_grid(int cus)
{
    int per_cu = 0;
    const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_eval_voxels_gen_fp<3>, 64, 0);
    if (e != hipSuccess || per_cu <= 0) per_cu = 16;
    return per_cu * cus;
}
void launch_eval_voxels_gen_fp(hipStream_t s, const VoxelArgs& a, const uint32_t* code, int grid, const unsigned* items, const int* meta, const GroupInfo* groups,
                               const ulonglong2* choice_masks, int choice_cap, int* counter, const unsigned long long* parent_records, int nchoices, int run,
                               int* walked, const unsigned char* skip)
{
    FpVoxelArgs j;
    j.v = a;
    j.code = code;
    j.items = items;
    j.meta = meta;
    j.groups = groups;
    j.choice_masks = choice_masks;
    j.choice_cap = choice_cap;
    j.counter = counter;
    j.parent_records = parent_records;
    j.nchoices = nchoices;
    j.run = run > 0 ? run : 2;
    j.walked = walked;
    j.skip = skip;
    hipLaunchKernelGGL(k_eval_voxels_gen_fp<3>, dim3(grid), dim3(64), 0, s, j);
}

#ifdef MPR_TEST_HOOKS
/* one tape through the host-generated code: a and b in the x and y slots, the tile's decisions wave-uniform */
__global__ void __launch_bounds__(64)
k_test_float_gen(const uint32_t* code, int n, const float* a, const float* b, float* out, unsigned long long dl, unsigned long long dr)
{
    const int i = blockIdx.x * 64 + (int)threadIdx.x;
    const float r = vox_gen_run(code, i < n ? a[i] : 0.0f, (i < n && b) ? b[i] : 0.0f, 0.0f, dl, dr);
    if (i < n) out[i] = r;
}
void launch_test_float_gen(hipStream_t s, const uint32_t* code, int n, const float* a, const float* b, float* out, unsigned long long dl,
                           unsigned long long dr)
{
    hipLaunchKernelGGL(k_test_float_gen, dim3((n + 63) / 64), dim3(64), 0, s, code, n, a, b, out, dl, dr);
}

/* a clause of one operand (or with a constant) through the host-generated code on every bit pattern of [first, first + count), against
 * float_clause (device_math.hpp: the float pass's definition): out = {tested, results that differ (two NaNs are the same), an input} */
__global__ void __launch_bounds__(64)
k_test_float_gen_all(const uint32_t* code, int op, float imm, unsigned long long first, unsigned long long count, unsigned long long* out)
{
    const int lane = threadIdx.x;
    unsigned long long tested = 0, bad = 0, example = 0;
    for (unsigned long long base = (unsigned long long)blockIdx.x * 64; base < count; base += (unsigned long long)gridDim.x * 64) {
        const uint32_t bits = (uint32_t)(first + base + lane);
        const float x = mpr_u2f(bits);
        const float got = vox_gen_run(code, x, x, 0.0f, 0ull, 0ull);
        const float want = float_clause((uint32_t)op, x, x, imm);
        if (base + lane >= count) continue;
        ++tested;
        if (mpr_f2u(got) != mpr_f2u(want) && !(got != got && want != want)) { ++bad; example = bits; }
    }
    atomicAdd(&out[0], tested);
    if (bad) { atomicAdd(&out[1], bad); out[2] = example; }
}
void launch_test_float_gen_all(hipStream_t s, const uint32_t* code, int op, float imm, unsigned long long first, unsigned long long count, unsigned long long* out)
{
    hipLaunchKernelGGL(k_test_float_gen_all, dim3(6144), dim3(64), 0, s, code, op, imm, first, count, out);
}

/* one clause through the translator and the generated code: tape3 as for k_test_float_asm */
__global__ void __launch_bounds__(64)
k_test_float_jit(const uint64_t* tape3, uint32_t* code, uint32_t region_dwords, int n, const float* a, const float* b, float* out)
{
    __shared__ __attribute__((aligned(64))) uint32_t lds[JIT_LDS_DWORDS];
    const int lane = threadIdx.x;
    jit_load_table(lds, lane, 64, false);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    const int i = blockIdx.x * 64 + lane;
    uint32_t* const my = code + (size_t)blockIdx.x * region_dwords;
    (void)jit_translate(tape3, 1u, tape3[0], my, (region_dwords - 320) * 4u + (uint32_t)lane * 16u, lds, lane);
    const float r = jit_run<24>(my, 1u, i < n ? a[i] : 0.0f, (i < n && b) ? b[i] : 0.0f, 0.0f);
    if (i < n) out[i] = r;
}
void launch_test_float_jit(hipStream_t s, const uint64_t* tape3, uint32_t* code, uint32_t region_dwords, int n, const float* a,
                           const float* b, float* out)
{
    hipLaunchKernelGGL(k_test_float_jit, dim3((n + 63) / 64), dim3(64), 0, s, tape3, code, region_dwords, n, a, b, out);
}

#endif  /* MPR_TEST_HOOKS */
int jit_max_choices() { return jt::JIT_MAX_CHOICES; }
int jit_slot_class(int nslots)
{
    return nslots <= 24 ? 24 : nslots <= 40 ? 40 : nslots <= 96 ? 96 : nslots <= 192 ? 192 : 0;
}
template <int DIM, int NS>
static int jit_grid_of(int cus, bool group)
{
    int per_cu = 0;
    hipError_t e;
    if (group) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_eval_voxels_jit_groups<DIM, NS>, 64 * JIT_GROUP_WAVES, 0);
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_eval_voxels_jit<DIM, NS>, 64, 0);
    if (e != hipSuccess || per_cu <= 0) per_cu = group ? 1 : 8;
    return per_cu * cus;
}
/* workgroups the device holds of the kernel for this slot class: the grid, and the number of code regions */
int jit_grid(int dim, int nslots, int cus, bool group)
{
    const int ns = jit_slot_class(nslots);
    if (dim == 3)
        return ns == 24 ? jit_grid_of<3, 24>(cus, group) : ns == 40 ? jit_grid_of<3, 40>(cus, group) : ns == 96 ? jit_grid_of<3, 96>(cus, group)
                                                                                                             : jit_grid_of<3, 192>(cus, group);
    return ns == 24 ? jit_grid_of<2, 24>(cus, group) : ns == 40 ? jit_grid_of<2, 40>(cus, group) : ns == 96 ? jit_grid_of<2, 96>(cus, group)
                                                                                                             : jit_grid_of<2, 192>(cus, group);
}
void launch_eval_voxels_jit(hipStream_t s, int dim, const VoxelArgs& a, uint32_t* code, uint32_t region_dwords, int slot_dwords, int slots, int grid,
                            int tape_len, const GroupInfo* groups, const ulonglong2* choice_masks, int choice_cap, int* group_counter,
                            const int* group_list, bool always_invalidate)
{
    if (a.count <= 0) return;
    JitVoxelArgs j;
    j.v = a;
    j.code = code;
    j.region_dwords = region_dwords;
    j.slot_dwords = (uint32_t)slot_dwords;
    j.slots = (uint32_t)slots;
    j.tape_len = tape_len;
    j.groups = groups;
    j.choice_masks = choice_masks;
    j.choice_cap = choice_cap;
    j.group_counter = group_counter;
    j.group_list = group_list;
    j.always_invalidate = always_invalidate ? 1 : 0;
    const int ns = jit_slot_class(a.nslots);
    if (groups) {
        const dim3 g(std::min(grid, (a.count + 63) / 64)), b(64 * JIT_GROUP_WAVES);
#define JIT_LAUNCH(D, N) hipLaunchKernelGGL((k_eval_voxels_jit_groups<D, N>), g, b, 0, s, j)
        if (dim == 3) {
            if (ns == 24) JIT_LAUNCH(3, 24); else if (ns == 40) JIT_LAUNCH(3, 40); else if (ns == 96) JIT_LAUNCH(3, 96); else JIT_LAUNCH(3, 192);
        } else {
            if (ns == 24) JIT_LAUNCH(2, 24); else if (ns == 40) JIT_LAUNCH(2, 40); else if (ns == 96) JIT_LAUNCH(2, 96); else JIT_LAUNCH(2, 192);
        }
#undef JIT_LAUNCH
        return;
    }
    const int runs = (a.count + JIT_RUN - 1) / JIT_RUN;
    const dim3 g(std::min(grid, runs)), b(64);
#define JIT_LAUNCH(D, N) hipLaunchKernelGGL((k_eval_voxels_jit<D, N>), g, b, 0, s, j)
    if (dim == 3) {
        if (ns == 24) JIT_LAUNCH(3, 24); else if (ns == 40) JIT_LAUNCH(3, 40); else if (ns == 96) JIT_LAUNCH(3, 96); else JIT_LAUNCH(3, 192);
    } else {
        if (ns == 24) JIT_LAUNCH(2, 24); else if (ns == 40) JIT_LAUNCH(2, 40); else if (ns == 96) JIT_LAUNCH(2, 96); else JIT_LAUNCH(2, 192);
    }
#undef JIT_LAUNCH
}

}  // namespace mprk

#ifdef MPR_TEST_HOOKS
/* What the translator makes of one clause (host restatement of its `dword = T + (T & mask) + base`, for the test that
 * disassembles the rows): table 0 tile form, 1 group form; row = the opcode, 30 (division by a constant), or 32.. */
extern "C" int mpr_test_jit_row(int32_t table, int32_t row, uint32_t clause_lo, uint32_t imm_bits, int32_t choice, uint32_t* out, int32_t cap)
{
    if (table < 0 || table > 1 || row < 0 || row >= mprk::JIT_ROWS || !out) return -1;
    const uint32_t* const w = mprk::h_jit_table[table].w[row];
    const int n = (int)(w[0] & 15u);
    if (n > cap) return -1;
    const uint32_t regs = ((clause_lo & 0xFFFFFF00u) + 0x30303000u) | ((128u + (uint32_t)choice) & 0xFFu);
    for (int k = 0; k < n; ++k) {
        const int at = 4 + 12 * (k / 4) + (k % 4);
        const uint32_t base = w[at], sel = w[at + 4], mask = w[at + 8];
        uint32_t T = 0;
        for (int b = 0; b < 4; ++b) {
            const uint32_t s = (sel >> (8 * b)) & 0xFFu;
            uint32_t byte = 0;
            if (s < 4) byte = (regs >> (8 * s)) & 0xFFu;
            else if (s < 8) byte = (imm_bits >> (8 * (s - 4))) & 0xFFu;
            T |= byte << (8 * b);
        }
        out[k] = T + (T & mask) + base;
    }
    return n;
}

#endif  /* MPR_TEST_HOOKS */
