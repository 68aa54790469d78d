/* gfx950_ir.hpp — a small instruction-level IR of the GFX9 family (gfx950) for the host-side code generator of the interval
 * forward walk (interval_gen.cpp): instructions over VIRTUAL registers (vector registers, scalar register pairs), so that a
 * clause's routine can be written once, in line, on whatever registers its operands live in, then list-scheduled against its
 * neighbours and given physical registers.  An instruction knows how to encode itself (the words the chip runs) and how to print
 * itself (assembler text): tests/test_interval_gen.py assembles the text with the ROCm assembler and compares the bytes, and runs
 * the same instructions through an emulator against the oracle's interval arithmetic — two independent readings of each
 * instruction, so that a wrong field shows on the CPU. */
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace mpr {
namespace ir {

enum class K : uint8_t {
    NONE,
    V,      /* virtual vector register */
    S,      /* virtual scalar register pair (a lane mask); as a 32-bit source: its first register */
    SH,     /* the second register of a virtual scalar register pair (32-bit sources) */
    PV,     /* physical vector register */
    PS,     /* physical scalar register: a pair's first register for 64-bit operands */
    IMM,    /* inline constant: the hardware's source code (128..248) */
    LIT,    /* the instruction's 32-bit literal */
    VCC,
    EXEC,
};
struct Opnd {
    K k = K::NONE;
    int32_t id = 0;
    bool is_reg() const { return k == K::V || k == K::S || k == K::SH || k == K::PV || k == K::PS || k == K::VCC; }
    bool operator==(const Opnd& o) const { return k == o.k && id == o.id; }
};
inline Opnd V(int id) { return {K::V, id}; }
inline Opnd S(int id) { return {K::S, id}; }
inline Opnd SH(int id) { return {K::SH, id}; }
inline Opnd PV(int r) { return {K::PV, r}; }
inline Opnd PS(int r) { return {K::PS, r}; }
inline Opnd IMM(int code) { return {K::IMM, code}; }
inline Opnd INT(int k) { return {K::IMM, 128 + k}; }             /* 0..64 */
inline Opnd LIT() { return {K::LIT, 0}; }
inline Opnd VCC() { return {K::VCC, 0}; }
inline Opnd NONE() { return {}; }
constexpr int C_ZERO = 128, C_ONE_I = 129, C_MINUS1_I = 193, C_HALF = 240, C_ONE = 242, C_MONE = 243, C_TWO = 244, C_FOUR = 246;

enum class Fmt : uint8_t { VOP1, VOP2, VOP3, VOPC, SOP1, SOP2, SOPC, SOPK, SOPP, DS, LABEL };
enum : uint16_t {
    F_TRANS = 1,        /* quarter-rate transcendental: its result needs a wait state before a non-transcendental reads it */
    F_HALF = 2,         /* issues at half rate (compares, selects) */
    F_WR_SCC = 4,       /* writes SCC as a by-product nobody reads */
    F_DEF_SCC = 8,      /* writes SCC for a reader (compare, bit test) */
    F_RD_SCC = 16,
    F_BARRIER = 32,     /* nothing moves across it (branches, calls, labels) */
    F_B64 = 64,         /* scalar operands are register pairs */
    F_COMMUTES = 128,   /* src0 and src1 may trade places */
    F_CALL = 256,       /* s_swappc_b64: clobbers what the called routine may */
};
enum class Op : uint8_t {
    V_MOV, V_EXP, V_LOG, V_RCP, V_SQRT, V_SIN, V_COS, V_FLOOR,
    V_CNDMASK, V_ADD_F32, V_SUB_F32, V_SUBREV_F32, V_MUL_F32, V_MIN_F32, V_MAX_F32, V_MIN_U32, V_MAX_U32, V_ASHRREV_I32,
    V_AND, V_OR, V_XOR, V_ADD_U32, V_SUB_U32,
    V_MAX3_F32, V_MED3_F32, V_FMA_F32, V_LSHL_OR, V_BFE_U32, V_WRITELANE,
    C_CLASS, C_LT, C_EQ, C_LE, C_GT, C_LG, C_GE, C_O, C_U, C_NGE, C_NLG, C_NGT, C_NLE, C_NEQ, C_NLT,
    C_LT_I32, C_GT_I32, C_LT_U32, C_EQ_U32, C_LE_U32, C_GT_U32, C_NE_U32, C_GE_U32,
    S_MOV_B32, S_MOV_B64, S_NOT_B64, S_SETPC, S_SWAPPC,
    S_AND_B64, S_OR_B64, S_ANDN2_B64, S_ORN2_B64, S_XOR_B64, S_CSELECT_B64,
    S_BITCMP1_B64, S_CMP_LG_U64, S_CMP_EQ_U64,
    S_MOVK,
    DS_WRITE_B128,
    S_NOP, S_BRANCH, S_CBRANCH_SCC0, S_CBRANCH_SCC1, S_CBRANCH_VCCZ, S_CBRANCH_VCCNZ,
    LABEL,
    COUNT
};
struct OpInfo {
    const char* name;
    Fmt fmt;
    uint16_t e32;       /* opcode in the short encoding (VOP1 / VOP2 / VOPC / SOPx) */
    uint16_t e64;       /* opcode in VOP3 */
    uint8_t nsrc;
    uint16_t flags;
};
inline const OpInfo& info(Op op)
{
    static const OpInfo T[(int)Op::COUNT] = {
        {"v_mov_b32", Fmt::VOP1, 1, 0x141, 1, 0},
        {"v_exp_f32", Fmt::VOP1, 0x20, 0x160, 1, F_TRANS},
        {"v_log_f32", Fmt::VOP1, 0x21, 0x161, 1, F_TRANS},
        {"v_rcp_f32", Fmt::VOP1, 0x22, 0x162, 1, F_TRANS},
        {"v_sqrt_f32", Fmt::VOP1, 0x27, 0x167, 1, F_TRANS},
        {"v_sin_f32", Fmt::VOP1, 0x29, 0x169, 1, F_TRANS},      /* argument in revolutions, |x| <= 256 */
        {"v_cos_f32", Fmt::VOP1, 0x2a, 0x16a, 1, F_TRANS},
        {"v_floor_f32", Fmt::VOP1, 0x1f, 0x15f, 1, 0},
        {"v_cndmask_b32", Fmt::VOP2, 0, 0x100, 3, F_HALF},
        {"v_add_f32", Fmt::VOP2, 1, 0x101, 2, F_COMMUTES},
        {"v_sub_f32", Fmt::VOP2, 2, 0x102, 2, 0},
        {"v_subrev_f32", Fmt::VOP2, 3, 0x103, 2, 0},
        {"v_mul_f32", Fmt::VOP2, 5, 0x105, 2, F_COMMUTES},
        {"v_min_f32", Fmt::VOP2, 10, 0x10a, 2, F_COMMUTES},
        {"v_max_f32", Fmt::VOP2, 11, 0x10b, 2, F_COMMUTES},
        {"v_min_u32", Fmt::VOP2, 14, 0x10e, 2, F_COMMUTES},
        {"v_max_u32", Fmt::VOP2, 15, 0x10f, 2, F_COMMUTES},
        {"v_ashrrev_i32", Fmt::VOP2, 17, 0x111, 2, 0},
        {"v_and_b32", Fmt::VOP2, 19, 0x113, 2, F_COMMUTES},
        {"v_or_b32", Fmt::VOP2, 20, 0x114, 2, F_COMMUTES},
        {"v_xor_b32", Fmt::VOP2, 21, 0x115, 2, F_COMMUTES},
        {"v_add_u32", Fmt::VOP2, 52, 0x134, 2, F_COMMUTES},
        {"v_sub_u32", Fmt::VOP2, 53, 0x135, 2, 0},
        {"v_max3_f32", Fmt::VOP3, 0, 0x1d3, 3, 0},
        {"v_med3_f32", Fmt::VOP3, 0, 0x1d6, 3, 0},
        {"v_fma_f32", Fmt::VOP3, 0, 0x1cb, 3, 0},
        {"v_lshl_or_b32", Fmt::VOP3, 0, 0x200, 3, 0},
        {"v_bfe_u32", Fmt::VOP3, 0, 0x1c8, 3, 0},
        /* dst[lane src1] = src0 (a scalar); the other lanes keep their values: src[2] = dst, for the dependencies only */
        {"v_writelane_b32", Fmt::VOP3, 0, 0x28a, 2, 0},
        {"v_cmp_class_f32", Fmt::VOPC, 0x10, 0x10, 2, F_HALF},
        {"v_cmp_lt_f32", Fmt::VOPC, 0x41, 0x41, 2, F_HALF},
        {"v_cmp_eq_f32", Fmt::VOPC, 0x42, 0x42, 2, F_HALF},
        {"v_cmp_le_f32", Fmt::VOPC, 0x43, 0x43, 2, F_HALF},
        {"v_cmp_gt_f32", Fmt::VOPC, 0x44, 0x44, 2, F_HALF},
        {"v_cmp_lg_f32", Fmt::VOPC, 0x45, 0x45, 2, F_HALF},
        {"v_cmp_ge_f32", Fmt::VOPC, 0x46, 0x46, 2, F_HALF},
        {"v_cmp_o_f32", Fmt::VOPC, 0x47, 0x47, 2, F_HALF},
        {"v_cmp_u_f32", Fmt::VOPC, 0x48, 0x48, 2, F_HALF},
        {"v_cmp_nge_f32", Fmt::VOPC, 0x49, 0x49, 2, F_HALF},
        {"v_cmp_nlg_f32", Fmt::VOPC, 0x4a, 0x4a, 2, F_HALF},
        {"v_cmp_ngt_f32", Fmt::VOPC, 0x4b, 0x4b, 2, F_HALF},
        {"v_cmp_nle_f32", Fmt::VOPC, 0x4c, 0x4c, 2, F_HALF},
        {"v_cmp_neq_f32", Fmt::VOPC, 0x4d, 0x4d, 2, F_HALF},
        {"v_cmp_nlt_f32", Fmt::VOPC, 0x4e, 0x4e, 2, F_HALF},
        {"v_cmp_lt_i32", Fmt::VOPC, 0xc1, 0xc1, 2, F_HALF},
        {"v_cmp_gt_i32", Fmt::VOPC, 0xc4, 0xc4, 2, F_HALF},
        {"v_cmp_lt_u32", Fmt::VOPC, 0xc9, 0xc9, 2, F_HALF},
        {"v_cmp_eq_u32", Fmt::VOPC, 0xca, 0xca, 2, F_HALF},
        {"v_cmp_le_u32", Fmt::VOPC, 0xcb, 0xcb, 2, F_HALF},
        {"v_cmp_gt_u32", Fmt::VOPC, 0xcc, 0xcc, 2, F_HALF},
        {"v_cmp_ne_u32", Fmt::VOPC, 0xcd, 0xcd, 2, F_HALF},
        {"v_cmp_ge_u32", Fmt::VOPC, 0xce, 0xce, 2, F_HALF},
        {"s_mov_b32", Fmt::SOP1, 0, 0, 1, 0},
        {"s_mov_b64", Fmt::SOP1, 1, 0, 1, F_B64},
        {"s_not_b64", Fmt::SOP1, 5, 0, 1, F_B64 | F_WR_SCC},
        {"s_setpc_b64", Fmt::SOP1, 0x1d, 0, 1, F_B64 | F_BARRIER},
        {"s_swappc_b64", Fmt::SOP1, 0x1e, 0, 1, F_B64 | F_BARRIER | F_CALL},
        {"s_and_b64", Fmt::SOP2, 13, 0, 2, F_B64 | F_WR_SCC},
        {"s_or_b64", Fmt::SOP2, 15, 0, 2, F_B64 | F_WR_SCC},
        {"s_andn2_b64", Fmt::SOP2, 19, 0, 2, F_B64 | F_WR_SCC},
        {"s_orn2_b64", Fmt::SOP2, 21, 0, 2, F_B64 | F_WR_SCC},
        {"s_xor_b64", Fmt::SOP2, 17, 0, 2, F_B64 | F_WR_SCC},
        {"s_cselect_b64", Fmt::SOP2, 11, 0, 2, F_B64 | F_RD_SCC},
        {"s_bitcmp1_b64", Fmt::SOPC, 0x0f, 0, 2, F_B64 | F_DEF_SCC},
        {"s_cmp_lg_u64", Fmt::SOPC, 0x13, 0, 2, F_B64 | F_DEF_SCC},
        {"s_cmp_eq_u64", Fmt::SOPC, 0x12, 0, 2, F_B64 | F_DEF_SCC},
        {"s_movk_i32", Fmt::SOPK, 0, 0, 0, 0},
        /* src[0] = address, src[1] = the first of four data registers, imm = byte offset; nothing moves across it */
        {"ds_write_b128", Fmt::DS, 0xdf, 0, 2, F_BARRIER},
        {"s_nop", Fmt::SOPP, 0, 0, 0, 0},
        {"s_branch", Fmt::SOPP, 2, 0, 0, F_BARRIER},
        {"s_cbranch_scc0", Fmt::SOPP, 4, 0, 0, F_BARRIER | F_RD_SCC},
        {"s_cbranch_scc1", Fmt::SOPP, 5, 0, 0, F_BARRIER | F_RD_SCC},
        {"s_cbranch_vccz", Fmt::SOPP, 6, 0, 0, F_BARRIER},
        {"s_cbranch_vccnz", Fmt::SOPP, 7, 0, 0, F_BARRIER},
        {"label", Fmt::LABEL, 0, 0, 0, F_BARRIER},
    };
    return T[(int)op];
}

struct Inst {
    Op op = Op::S_NOP;
    Opnd dst;
    Opnd src[3];
    uint8_t neg = 0, abs = 0;       /* bit k: source k is negated / taken by its magnitude (float sources of VOP3 encodings) */
    uint32_t lit = 0;
    int32_t imm = 0;                /* s_nop count, s_movk immediate, LABEL id, branch: label id (target < 0) or dword offset once resolved */
    int32_t clause = 0;             /* the clause the instruction belongs to (the scheduler's window is counted in clauses) */
    bool resolved = false;          /* branches: imm is the offset */
    uint16_t flags() const { return info(op).flags; }
    bool is_valu() const { const Fmt f = info(op).fmt; return f == Fmt::VOP1 || f == Fmt::VOP2 || f == Fmt::VOP3 || f == Fmt::VOPC; }
    bool is_branch() const { return op >= Op::S_BRANCH && op <= Op::S_CBRANCH_VCCNZ; }
    bool has_lit() const { return src[0].k == K::LIT || src[1].k == K::LIT || src[2].k == K::LIT; }
};

/* ---- physical forms ---- */
inline bool phys(const Opnd& o) { return o.k != K::V && o.k != K::S && o.k != K::SH; }
/* the 9-bit source field of a physical operand */
inline uint32_t src_code(const Opnd& o)
{
    switch (o.k) {
        case K::PV: return 256u + (uint32_t)o.id;
        case K::PS: return (uint32_t)o.id;
        case K::IMM: return (uint32_t)o.id;
        case K::LIT: return 255u;
        case K::VCC: return 106u;
        case K::EXEC: return 126u;
        default: return 0u;
    }
}
inline bool is_sgpr_src(const Opnd& o) { return o.k == K::PS || o.k == K::VCC || o.k == K::EXEC; }

/* does the instruction (physical operands) fit its short encoding? */
inline bool fits_e32(const Inst& i)
{
    const OpInfo& f = info(i.op);
    if (i.neg || i.abs) return false;
    switch (f.fmt) {
        case Fmt::VOP1: return i.dst.k == K::PV;
        case Fmt::VOP2:
            if (i.src[1].k != K::PV) return false;
            if (i.op == Op::V_CNDMASK && i.src[2].k != K::VCC) return false;
            return true;
        case Fmt::VOPC: return i.dst.k == K::VCC && i.src[1].k == K::PV;
        default: return true;
    }
}
/* dwords the instruction occupies */
inline int size_dw(const Inst& i)
{
    const OpInfo& f = info(i.op);
    if (f.fmt == Fmt::LABEL) return 0;
    if (f.fmt == Fmt::VOP3 || f.fmt == Fmt::DS) return 2;
    if (f.fmt == Fmt::VOP1 || f.fmt == Fmt::VOP2 || f.fmt == Fmt::VOPC) return fits_e32(i) ? (i.has_lit() ? 2 : 1) : 2;
    return i.has_lit() ? 2 : 1;
}
/* appends the instruction's words; false: it has no encoding (a literal or a second scalar source in a VOP3 form, a virtual register) */
inline bool encode(const Inst& i, std::vector<uint32_t>& out)
{
    const OpInfo& f = info(i.op);
    for (int k = 0; k < 3; ++k)
        if (!phys(i.src[k])) return false;
    if (!phys(i.dst)) return false;
    switch (f.fmt) {
        case Fmt::LABEL: return true;
        case Fmt::VOP1:
        case Fmt::VOP2:
        case Fmt::VOPC:
        case Fmt::VOP3: {
            if (f.fmt != Fmt::VOP3 && fits_e32(i)) {
                if (f.fmt == Fmt::VOP1) out.push_back(0x7E000000u | (uint32_t)i.dst.id << 17 | (uint32_t)f.e32 << 9 | src_code(i.src[0]));
                else if (f.fmt == Fmt::VOP2) out.push_back((uint32_t)f.e32 << 25 | (uint32_t)i.dst.id << 17 | (uint32_t)i.src[1].id << 9 | src_code(i.src[0]));
                else out.push_back(0x7C000000u | (uint32_t)f.e32 << 17 | (uint32_t)i.src[1].id << 9 | src_code(i.src[0]));
                if (i.has_lit()) out.push_back(i.lit);
                return true;
            }
            if (i.has_lit()) return false;
            int nsgpr = 0;
            uint32_t seen = ~0u;
            for (int k = 0; k < 3; ++k)
                if (is_sgpr_src(i.src[k]) && src_code(i.src[k]) != seen) { ++nsgpr; seen = src_code(i.src[k]); }
            if (nsgpr > 1) return false;            /* one scalar operand per vector instruction (the constant bus) */
            uint32_t d;
            if (f.fmt == Fmt::VOPC) d = i.dst.k == K::VCC ? 106u : (uint32_t)i.dst.id;
            else d = (uint32_t)i.dst.id;
            if (f.fmt != Fmt::VOPC && i.dst.k != K::PV) return false;
            out.push_back(0xD0000000u | (uint32_t)f.e64 << 16 | (uint32_t)i.abs << 8 | d);
            out.push_back(src_code(i.src[0]) | src_code(i.src[1]) << 9 | (i.op == Op::V_WRITELANE ? 0u : src_code(i.src[2]) << 18) | (uint32_t)i.neg << 29);
            return true;
        }
        case Fmt::SOP1: {
            const uint32_t d = i.dst.k == K::NONE ? 0u : i.dst.k == K::VCC ? 106u : (uint32_t)i.dst.id;
            out.push_back(0xBE800000u | d << 16 | (uint32_t)f.e32 << 8 | src_code(i.src[0]));
            if (i.has_lit()) out.push_back(i.lit);
            return true;
        }
        case Fmt::SOP2: {
            const uint32_t d = i.dst.k == K::VCC ? 106u : (uint32_t)i.dst.id;
            out.push_back(0x80000000u | (uint32_t)f.e32 << 23 | d << 16 | src_code(i.src[1]) << 8 | src_code(i.src[0]));
            if (i.has_lit()) out.push_back(i.lit);
            return true;
        }
        case Fmt::SOPC:
            out.push_back(0xBF000000u | (uint32_t)f.e32 << 16 | src_code(i.src[1]) << 8 | src_code(i.src[0]));
            return true;
        case Fmt::DS:
            if (i.src[0].k != K::PV || i.src[1].k != K::PV || i.imm < 0 || i.imm > 0xFFFF) return false;
            out.push_back(0xD8000000u | (uint32_t)f.e32 << 17 | (uint32_t)i.imm);
            out.push_back((uint32_t)i.src[0].id | (uint32_t)i.src[1].id << 8);
            return true;
        case Fmt::SOPK:
            out.push_back(0xB0000000u | (uint32_t)i.dst.id << 16 | ((uint32_t)i.imm & 0xFFFFu));
            return true;
        case Fmt::SOPP:
            if (i.is_branch() && !i.resolved) return false;
            out.push_back(0xBF800000u | (uint32_t)f.e32 << 16 | ((uint32_t)i.imm & 0xFFFFu));
            return true;
    }
    return false;
}

/* ---- assembler text (what the ROCm assembler reads back into the same words) ---- */
inline std::string opnd_text(const Opnd& o, bool pair, bool as_float, uint32_t lit)
{
    char b[48];
    switch (o.k) {
        case K::V: snprintf(b, sizeof b, "%%v%d", o.id); return b;
        case K::S: snprintf(b, sizeof b, "%%s%d", o.id); return b;
        case K::SH: snprintf(b, sizeof b, "%%s%d.hi", o.id); return b;
        case K::PV: snprintf(b, sizeof b, "v%d", o.id); return b;
        case K::PS:
            if (pair) snprintf(b, sizeof b, "s[%d:%d]", o.id, o.id + 1);
            else snprintf(b, sizeof b, "s%d", o.id);
            return b;
        case K::VCC: return "vcc";
        case K::EXEC: return "exec";
        case K::LIT: snprintf(b, sizeof b, "0x%x", lit); return b;
        case K::IMM:
            if (o.id >= 128 && o.id <= 192) { snprintf(b, sizeof b, "%d", o.id - 128); return b; }
            if (o.id >= 193 && o.id <= 208) { snprintf(b, sizeof b, "%d", -(o.id - 192)); return b; }
            switch (o.id) {
                case 240: return "0.5";
                case 241: return "-0.5";
                case 242: return "1.0";
                case 243: return "-1.0";
                case 244: return "2.0";
                case 245: return "-2.0";
                case 246: return "4.0";
                case 247: return "-4.0";
                case 248: return "0.15915494";
                default: break;
            }
            snprintf(b, sizeof b, "?%d", o.id);
            return b;
        default: return "";
    }
    (void)as_float;
}
inline std::string text(const Inst& i)
{
    const OpInfo& f = info(i.op);
    char b[64];
    if (f.fmt == Fmt::LABEL) { snprintf(b, sizeof b, "L%d:", i.imm); return b; }
    if (f.fmt == Fmt::SOPP) {
        if (i.is_branch() && !i.resolved) snprintf(b, sizeof b, "%s L%d", f.name, -i.imm - 1);
        else snprintf(b, sizeof b, "%s %d", f.name, (int)(int16_t)i.imm);
        return b;
    }
    if (f.fmt == Fmt::DS) {
        if (i.imm) snprintf(b, sizeof b, "%s v%d, v[%d:%d] offset:%d", f.name, i.src[0].id, i.src[1].id, i.src[1].id + 3, i.imm);
        else snprintf(b, sizeof b, "%s v%d, v[%d:%d]", f.name, i.src[0].id, i.src[1].id, i.src[1].id + 3);
        return b;
    }
    if (f.fmt == Fmt::SOPK) { snprintf(b, sizeof b, "%s s%d, 0x%x", f.name, i.dst.id, (unsigned)i.imm & 0xFFFFu); return b; }
    const bool b64 = (f.flags & F_B64) != 0;
    std::string s = f.name;
    s += " ";
    bool first = true;
    if (i.dst.k != K::NONE) {
        s += opnd_text(i.dst, i.is_valu() ? i.dst.k == K::PS : b64, false, 0);
        first = false;
    }
    for (int k = 0; k < f.nsrc; ++k) {
        if (i.src[k].k == K::NONE) continue;
        if (!first) s += ", ";
        first = false;
        /* scalar operands of vector instructions are single registers, except the lane masks (a select's third source) */
        const bool pair = i.is_valu() ? (i.op == Op::V_CNDMASK && k == 2) : (b64 && !(i.op == Op::S_BITCMP1_B64 && k == 1));
        std::string o = opnd_text(i.src[k], pair, true, i.lit);
        if (i.abs >> k & 1) o = "|" + o + "|";
        if (i.neg >> k & 1) o = "-" + o;
        s += o;
    }
    return s;
}

}  // namespace ir
}  // namespace mpr
