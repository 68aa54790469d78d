/* voxel_gen.cpp — see voxel_gen.hpp.  Instruction encodings: GFX9 family (gfx950); tests/test_voxel_gen.py disassembles what
 * this file emits with the ROCm assembler and compares it with the instructions it is meant to be.  The inline bodies are
 * asm_float_bodies.hpp's with the registers renamed (argument A instead of v35, result O instead of v37): keep the two in
 * step — tests/test_gpu_primitives.py runs every opcode on every class of bit pattern through both and the oracle. */
#include "voxel_gen.hpp"

#include <algorithm>
#include <cstring>

#include "../../include/mpr_clause.h"
#include "../../include/mpr_amd_test.h"
#include "gfx950_emit.hpp"

namespace mpr {
namespace {

using namespace gfx;

/* VOP1 / VOP2 / VOPC / VOP3 opcodes this file adds to gfx950_emit.hpp's */
constexpr int V1_CVT_F32_I32 = 5, V1_RSQ_F32 = 36, V1_BFREV = 44;
constexpr int V_MIN_F32 = 10, V_MAX_F32 = 11, V_LSHRREV = 16, V_FMAMK = 23, V_FMAAK = 24, V_FMAC = 59;
constexpr int VC_CLASS_F32 = 0x10, VC_GT_F32 = 0x44, VC_NLE_F32 = 0x4C, VC_GT_U32 = 0xCC;
constexpr int V3_FMA_F32 = 0x1CB, V3_LSHL_ADD = 0x1FD;
constexpr uint32_t K_HALF = 240, K_MHALF = 241, K_ONE = 242, K_MONE = 243;

struct Fix { size_t at; int target; };          /* a branch word and what it jumps to */

struct Gen {
    std::vector<uint32_t> c;                    /* the code */
    std::vector<uint32_t> st;                   /* the stubs behind it */
    std::vector<Fix> to_stub;                   /* branch at c[at] -> stub at st[target] */
    std::vector<Fix> to_code;                   /* branch at st[at] -> c[target] */
    std::vector<Fix> in_code;                   /* branch at c[at] -> c[target] */
    int nstubs = 0;

    static int R(int slot) { return VG_SLOT_BASE + slot; }
    static uint32_t V(int r) { return 256u + (uint32_t)r; }
    /* a 32-bit constant as a source operand: the inline constant where there is one (same bits, one dword less) */
    static bool inline_const(uint32_t k, uint32_t* src)
    {
        if (k <= 64u) { *src = 128u + k; return true; }
        if (k >= 0xFFFFFFF0u) { *src = 193u + (0xFFFFFFFFu - k); return true; }
        switch (k) {
            case 0x3f000000u: *src = 240; return true;
            case 0xbf000000u: *src = 241; return true;
            case 0x3f800000u: *src = 242; return true;
            case 0xbf800000u: *src = 243; return true;
            case 0x40000000u: *src = 244; return true;
            case 0xc0000000u: *src = 245; return true;
            case 0x40800000u: *src = 246; return true;
            case 0xc0800000u: *src = 247; return true;
            case 0x3e22f983u: *src = 248; return true;
            default: return false;
        }
    }
    void vop1(std::vector<uint32_t>& o, int op, int vdst, uint32_t src0) { o.push_back(0x7E000000u | (uint32_t)vdst << 17 | (uint32_t)op << 9 | src0); }
    void vop2(int op, int vdst, uint32_t src0, int vsrc1) { c.push_back((uint32_t)op << 25 | (uint32_t)vdst << 17 | (uint32_t)vsrc1 << 9 | src0); }
    void vop2_k(int op, int vdst, uint32_t k, int vsrc1)                       /* src0 = the constant k */
    {
        uint32_t s;
        if (inline_const(k, &s)) vop2(op, vdst, s, vsrc1);
        else { vop2(op, vdst, 255, vsrc1); c.push_back(k); }
    }
    void vop2_lit(int op, int vdst, uint32_t lit, int vsrc1) { vop2(op, vdst, 255, vsrc1); c.push_back(lit); }
    void fmamk(int vdst, int v0, uint32_t k, int v1) { vop2(V_FMAMK, vdst, V(v0), v1); c.push_back(k); }      /* v0 * k + v1 */
    void fmaak(int vdst, int v0, int v1, uint32_t k) { vop2(V_FMAAK, vdst, V(v0), v1); c.push_back(k); }      /* v0 * v1 + k */
    void vopc(int op, uint32_t src0, int vsrc1) { c.push_back(0x7C000000u | (uint32_t)op << 17 | (uint32_t)vsrc1 << 9 | src0); }
    void vopc_lit(int op, uint32_t lit, int vsrc1) { vopc(op, 255, vsrc1); c.push_back(lit); }
    void vop3(int op, uint32_t dst, uint32_t s0, uint32_t s1, uint32_t s2, uint32_t neg = 0, uint32_t abs = 0)
    {
        c.push_back(0xD0000000u | (uint32_t)op << 16 | abs << 8 | dst);
        c.push_back(s0 | s1 << 9 | s2 << 18 | neg << 29);
    }
    void mov(std::vector<uint32_t>& o, int vdst, uint32_t src0) { vop1(o, 1, vdst, src0); }
    void mov_k(std::vector<uint32_t>& o, int vdst, uint32_t k)
    {
        uint32_t s;
        if (inline_const(k, &s)) mov(o, vdst, s);
        else { mov(o, vdst, 255); o.push_back(k); }
    }
    void nop(int n) { c.push_back(0xBF800000u | (uint32_t)n); }
    void cmp_vcc_exec() { c.push_back(0xBF127E6Au); }                                     /* s_cmp_eq_u64 vcc, exec */
    void bitcmp1(int spair, int bit) { c.push_back(0xBF0F0000u | (uint32_t)(128 + bit) << 8 | (uint32_t)spair); }
    static void call(std::vector<uint32_t>& o, int pair) { o.push_back(0xBE9E1E00u | (uint32_t)pair); }   /* s_swappc_b64 s[30:31], s[pair:pair+1] */
    enum { BR = 2, SCC0 = 4, SCC1 = 5, VCCNZ = 7 };
    /* the stub being written starts here; returns its index */
    int stub_begin() { ++nstubs; return (int)st.size(); }
    void branch_to_stub(int sopp, int stub) { to_stub.push_back({c.size(), stub}); c.push_back(0xBF800000u | (uint32_t)sopp << 16); }
    /* ends a stub: back into the code at the dword that is emitted next */
    void stub_end_here() { to_code.push_back({st.size(), (int)c.size()}); st.push_back(0xBF820000u); }
    /* stubs are written before the code they return to exists: remember the branch, give the target later */
    size_t stub_end_later() { st.push_back(0xBF820000u); return st.size() - 1; }

    /* rare operands: the interpreters' full routine on (v35 [, v36]) -> v37 */
    int stub_call1(int routine, int A, int O, std::vector<size_t>& backs)
    {
        const int s = stub_begin();
        mov(st, 35, V(A));
        call(st, routine);
        mov(st, O, V(37));
        backs.push_back(stub_end_later());
        return s;
    }
    void finish_backs(const std::vector<size_t>& backs)
    {
        for (size_t at : backs) to_code.push_back({at, (int)c.size()});
    }

    void sqrt_inline(int A, int O)
    {
        std::vector<size_t> backs;
        const int s = stub_call1(VG_RT_SQRT, A, O, backs);
        vop2_lit(V_ADD_U32, 38, 0xf0800000u, A);              /* bits - 0x0f800000 */
        vopc_lit(VC_GT_U32, 0x70000000u, 38);                 /* positive, normal, finite, not tiny: in every lane? */
        cmp_vcc_exec();
        branch_to_stub(SCC0, s);
        vop1(c, V1_RSQ_F32, 39, V(A));
        nop(0);
        vop2(V_MUL_F32, 40, V(A), 39);                        /* g */
        vop2(V_MUL_F32, 39, K_HALF, 39);                      /* h */
        vop3(V3_FMA_F32, 41, V(39), V(40), K_HALF, 1);        /* r = 0.5 - h g */
        vop2(V_FMAC, 40, V(40), 41);
        vop2(V_FMAC, 39, V(39), 41);
        vop3(V3_FMA_F32, 41, V(40), V(40), V(A), 1);          /* x - g g */
        vop3(V3_FMA_F32, (uint32_t)O, V(41), V(39), V(40));
        finish_backs(backs);
    }
    void exp_inline(int A, int O)
    {
        std::vector<size_t> backs;
        const int s = stub_call1(VG_RT_EXP, A, O, backs);
        vop3(VC_NLE_F32, 106, V(A), VG_K_87, 0, 0, 1);        /* vcc = !(|x| <= 87): some special case applies */
        branch_to_stub(VCCNZ, s);
        vop2_lit(V_MUL_F32, 38, 0x3fb8aa3bu, A);
        vop2_lit(V_ADD_F32, 42, 0x4b400000u, 38);             /* t + 1.5 * 2^23 */
        vop2_lit(V_ADD_F32, 38, 0xcb400000u, 42);             /* kf */
        fmamk(39, 38, 0xbf318000u, A);
        fmamk(39, 38, 0x395e8083u, 39);                       /* r */
        fmaak(40, VG_K_EXP_C5, 39, 0x3ab743ceu);              /* (v_mov + v_fmac in the routine: the same fused operation) */
        fmaak(40, 40, 39, 0x3c088908u);
        fmaak(40, 40, 39, 0x3d2aa9c1u);
        fmaak(40, 40, 39, 0x3e2aaaaau);
        vop2(V_MUL_F32, 41, V(39), 39);
        vop3(V3_FMA_F32, 40, V(40), V(39), K_HALF);
        vop2(V_FMAC, 39, V(40), 41);
        vop2(V_ADD_F32, 39, K_ONE, 39);                       /* p */
        vop3(V3_LSHL_ADD, (uint32_t)O, V(42), 128 + 23, V(39));   /* p * 2^k for a normal result */
        finish_backs(backs);
    }
    void log_inline(int A, int O)
    {
        std::vector<size_t> backs;
        const int s = stub_call1(VG_RT_LOG, A, O, backs);
        vop3(VC_CLASS_F32, 106, V(A), VG_K_POSNORMAL, 0);     /* every lane a positive normal number? */
        cmp_vcc_exec();
        branch_to_stub(SCC0, s);
        vop2(V_LSHRREV, 39, 128 + 23, A);
        vop2_lit(V_ADD_U32, 39, 0xffffff82u, 39);             /* e */
        vop2_lit(V_AND, 38, 0x7fffffu, A);
        vop2(V_OR, 38, K_HALF, 38);                           /* m in [0.5, 1) */
        vopc_lit(VC_GT_F32, 0x3f3504f3u, 38);
        vop1(c, V1_BFREV, 41, 129);
        nop(0);
        c.push_back(0xD11E0000u | 92u << 8 | 39u);            /* v_subbrev_co_u32 v39, s[92:93], 0, v39, vcc */
        c.push_back(128u | V(39) << 9 | 106u << 18);
        vop2(V_CNDMASK, 41, V(41), 38);
        vop2(V_ADD_F32, 38, V(41), 38);                       /* m + m, or m + (-0) */
        vop2(V_ADD_F32, 38, K_MONE, 38);
        fmaak(40, VG_K_LOG_C8, 38, 0xbdebd1b8u);
        fmaak(40, 40, 38, 0x3def251au);
        fmaak(40, 40, 38, 0xbdfe5d4fu);
        fmaak(40, 40, 38, 0x3e11e9bfu);
        fmaak(40, 40, 38, 0xbe2aae50u);
        fmaak(40, 40, 38, 0x3e4cceacu);
        fmaak(40, 40, 38, 0xbe7ffffcu);
        vop1(c, V1_CVT_F32_I32, 39, V(39));                   /* fe */
        fmaak(40, 40, 38, 0x3eaaaaaau);
        vop2(V_MUL_F32, 41, V(38), 38);                       /* z */
        vop2(V_MUL_F32, 40, V(40), 38);
        vop2(V_MUL_F32, 40, V(40), 41);
        fmamk(40, 39, 0xb95e8083u, 40);
        vop2(V_FMAC, 40, K_MHALF, 41);
        vop2(V_ADD_F32, 40, V(38), 40);
        fmamk(O, 39, 0x3f318000u, 40);
        finish_backs(backs);
    }
    /* x / c for a constant c with 2^-30 <= |c| <= 2^30 that is not a power of two; y = RN(1 / c).  kernels_voxel_jit.hip,
     * row 30: inline when x * x is a positive normal number in every lane, else the general division */
    void divc_inline(int A, int O, uint32_t cbits)
    {
        float cf, yf;
        std::memcpy(&cf, &cbits, 4);
        yf = 1.0f / cf;                                       /* IEEE: correctly rounded, as the device's sequence is */
        uint32_t y;
        std::memcpy(&y, &yf, 4);
        const uint32_t negc = cbits ^ 0x80000000u;
        const int s = stub_begin();
        mov(st, 35, V(A));
        mov_k(st, 36, cbits);
        call(st, VG_RT_DIV);
        mov(st, O, V(37));
        const size_t back = stub_end_later();
        vop2(V_MUL_F32, 39, V(A), A);
        c.push_back(0x7C200F27u);                             /* v_cmp_class_f32 vcc, v39, v7: anything but a positive normal */
        branch_to_stub(VCCNZ, s);
        vop2_lit(V_MUL_F32, 37, y, A);                        /* q = y x */
        fmamk(39, 37, negc, A);                               /* r = x - c q */
        fmamk(37, 39, y, 37);                                 /* q += r y */
        fmamk(39, 37, negc, A);
        fmamk(O, 39, y, 37);
        to_code.push_back({back, (int)c.size()});
    }
    void call_unary(int routine, int A, int O)
    {
        mov(c, 35, V(A));
        call(c, routine);
        mov(c, O, V(37));
    }
    void call_leaf(int routine, int A, int O)
    {
        mov(c, 0, V(A));
        call(c, routine);
        mov(c, O, V(0));
        mov_k(c, 7, 0x2ffu);                                  /* the compiled leaves may use v0..v7 */
        mov_k(c, VG_K_EXP_C5, 0x39506967u);
        mov_k(c, VG_K_LOG_C8, 0x3d9021bbu);
    }
    /* l / r: slot registers, r < 0: the immediate */
    void minmax(int op, int k, int A, int B, uint32_t K, int O)
    {
        /* decided for the lhs / rhs: the chosen operand, raw (what COPY_LHS / COPY_RHS / COPY_IMM of the tile's own tape do) */
        const size_t after_fix_l = in_code.size();
        bitcmp1(VG_DEC_L, k);
        int sl = -1;
        if (O != A) {
            sl = stub_begin();
            mov(st, O, V(A));
        }
        size_t back_l = 0;
        if (sl >= 0) { back_l = stub_end_later(); branch_to_stub(SCC1, sl); }
        else { in_code.push_back({c.size(), -1}); c.push_back(0xBF850000u); }
        bitcmp1(VG_DEC_R, k);
        int sr = -1;
        if (B < 0 || O != B) {
            sr = stub_begin();
            if (B >= 0) mov(st, O, V(B));
            else mov_k(st, O, K);
        }
        size_t back_r = 0;
        if (sr >= 0) { back_r = stub_end_later(); branch_to_stub(SCC1, sr); }
        else { in_code.push_back({c.size(), -1}); c.push_back(0xBF850000u); }
        if (B >= 0) vop2(op, O, V(A), B);
        else vop2_k(op, O, K, A);
        const int here = (int)c.size();
        if (sl >= 0) to_code.push_back({back_l, here});
        if (sr >= 0) to_code.push_back({back_r, here});
        for (size_t i = after_fix_l; i < in_code.size(); ++i)
            if (in_code[i].target == -1) in_code[i].target = here;
    }
};

bool uses_lhs(uint32_t op)
{
    return (op >= MPR_OP_SQUARE_LHS && op <= MPR_OP_LOG_LHS) || op == MPR_OP_ADD_LHS_IMM || op == MPR_OP_ADD_LHS_RHS ||
           op == MPR_OP_MUL_LHS_IMM || op == MPR_OP_MUL_LHS_RHS || mpr_op_is_minmax(op) || op == MPR_OP_SUB_LHS_IMM ||
           op == MPR_OP_SUB_LHS_RHS || op == MPR_OP_DIV_LHS_IMM || op == MPR_OP_DIV_LHS_RHS || op == MPR_OP_COPY_LHS;
}
bool uses_rhs(uint32_t op)
{
    return op == MPR_OP_ADD_LHS_RHS || op == MPR_OP_MUL_LHS_RHS || op == MPR_OP_MIN_LHS_RHS || op == MPR_OP_MAX_LHS_RHS ||
           op == MPR_OP_SUB_IMM_RHS || op == MPR_OP_SUB_LHS_RHS || op == MPR_OP_DIV_IMM_RHS || op == MPR_OP_DIV_LHS_RHS ||
           op == MPR_OP_COPY_RHS;
}

}  // namespace

/* The clauses only one operand of a min / max clause reaches, as runs (voxel_gen.hpp). */
std::vector<DeadRun> tape_dead_runs(const uint64_t* cl, int end, int min_run)
{
    std::vector<DeadRun> runs;
    if (min_run <= 0) return runs;
    std::vector<int> ldef(end, -1), rdef(end, -1), choice_of(end, -1);
    int cur[256];
    for (int& x : cur) x = -1;
    int nch = 0;
    for (int i = 1; i < end; ++i) {
        const uint32_t op = mpr_cl_op(cl[i]);
        if (uses_lhs(op)) ldef[i] = cur[mpr_cl_lhs(cl[i])];
        if (uses_rhs(op)) rdef[i] = cur[mpr_cl_rhs(cl[i])];
        if (mpr_op_is_minmax(op)) choice_of[i] = nch++;
        cur[mpr_cl_out(cl[i])] = i;
    }
    const int root = cur[mpr_cl_out(cl[end])];
    auto reach = [&](int skip_clause, int skip_side, std::vector<char>& seen) {
        seen.assign(end, 0);
        std::vector<int> stack;
        if (root >= 1) stack.push_back(root);
        while (!stack.empty()) {
            const int c = stack.back();
            stack.pop_back();
            if (seen[c]) continue;
            seen[c] = 1;
            if (ldef[c] >= 1 && !(c == skip_clause && skip_side == 0)) stack.push_back(ldef[c]);
            if (rdef[c] >= 1 && !(c == skip_clause && skip_side == 1)) stack.push_back(rdef[c]);
        }
    };
    std::vector<char> all, part;
    reach(-1, -1, all);
    for (int m = 1; m < end; ++m) {
        if (choice_of[m] < 0 || !all[m]) continue;
        for (int side = 0; side < 2; ++side) {
            if ((side == 0 ? ldef[m] : rdef[m]) < 1) continue;
            reach(m, side, part);
            /* clauses that die with this operand; clauses the result never reaches are not in any run (they run: harmless) */
            int i = 1;
            while (i < end) {
                if (!(all[i] && !part[i])) { ++i; continue; }
                int j = i;
                while (j + 1 < end && all[j + 1] && !part[j + 1]) ++j;
                /* dead when m does not take this operand: decided for the OTHER side */
                if (j - i + 1 >= min_run) runs.push_back({i, j, choice_of[m], side == 1});
                i = j + 1;
            }
        }
    }
    return runs;
}

VoxelGen voxel_gen_build(const uint64_t* clauses, int len, int min_run)
{
    VoxelGen g;
    if (!clauses || len < 2) return g;
    int end = -1, nch = 0;
    for (int i = 1; i < len; ++i) {
        const uint32_t op = mpr_cl_op(clauses[i]);
        const int o = (int)mpr_cl_out(clauses[i]), l = (int)mpr_cl_lhs(clauses[i]), r = (int)mpr_cl_rhs(clauses[i]);
        if (o >= VG_MAX_SLOTS || l >= VG_MAX_SLOTS || r >= VG_MAX_SLOTS) return g;
        if (op == MPR_OP_INVALID) {
            end = i;
            break;
        }
        if (op == MPR_OP_JUMP || op >= MPR_OP_COUNT || o == 0) return g;
        if ((uses_lhs(op) && l == 0) || (uses_rhs(op) && r == 0)) return g;      /* slot 0 is nobody's */
        if (mpr_op_is_minmax(op)) ++nch;
    }
    if (end < 0 || nch > VG_MAX_CHOICES) return g;
    const int hx = (int)mpr_cl_out(clauses[0]), hy = (int)mpr_cl_lhs(clauses[0]), hz = (int)mpr_cl_rhs(clauses[0]);
    if (hx >= VG_MAX_SLOTS || hy >= VG_MAX_SLOTS || hz >= VG_MAX_SLOTS) return g;

    std::vector<DeadRun> runs = tape_dead_runs(clauses, end, min_run);
    /* at a clause: outer runs first */
    std::stable_sort(runs.begin(), runs.end(), [](const DeadRun& a, const DeadRun& b) { return a.first != b.first ? a.first < b.first : a.last > b.last; });

    Gen e;
    e.mov(e.c, Gen::R(hx), Gen::V(32));
    e.mov(e.c, Gen::R(hy), Gen::V(33));
    e.mov(e.c, Gen::R(hz), Gen::V(34));
    std::vector<int> pos(end + 1, 0);
    std::vector<Fix> guards;                                  /* branch at c[at] -> the clause after the run (an index into pos) */
    size_t next_run = 0;
    int choice = 0;
    for (int i = 1; i < end; ++i) {
        pos[i] = (int)e.c.size();
        for (; next_run < runs.size() && runs[next_run].first == i; ++next_run) {
            const DeadRun& r = runs[next_run];
            e.bitcmp1(r.by_lhs ? VG_DEC_L : VG_DEC_R, r.choice);
            guards.push_back({e.c.size(), r.last + 1});
            e.c.push_back(0xBF850000u);                       /* s_cbranch_scc1 */
        }
        const uint64_t w = clauses[i];
        const uint32_t op = mpr_cl_op(w), K = mpr_cl_immbits(w);
        const int O = Gen::R((int)mpr_cl_out(w)), A = Gen::R((int)mpr_cl_lhs(w)), B = Gen::R((int)mpr_cl_rhs(w));
        switch (op) {
            case MPR_OP_SQUARE_LHS: e.vop2(V_MUL_F32, O, Gen::V(A), A); break;
            case MPR_OP_SQRT_LHS: e.sqrt_inline(A, O); break;
            case MPR_OP_NEG_LHS: e.vop2_lit(V_XOR, O, 0x80000000u, A); break;
            case MPR_OP_SIN_LHS: e.call_unary(VG_RT_SIN, A, O); break;
            case MPR_OP_COS_LHS: e.call_unary(VG_RT_COS, A, O); break;
            case MPR_OP_ASIN_LHS: e.call_leaf(VG_RT_ASIN, A, O); break;
            case MPR_OP_ACOS_LHS: e.call_leaf(VG_RT_ACOS, A, O); break;
            case MPR_OP_ATAN_LHS: e.call_leaf(VG_RT_ATAN, A, O); break;
            case MPR_OP_EXP_LHS: e.exp_inline(A, O); break;
            case MPR_OP_ABS_LHS: e.vop2_lit(V_AND, O, 0x7fffffffu, A); break;
            case MPR_OP_LOG_LHS: e.log_inline(A, O); break;
            case MPR_OP_ADD_LHS_IMM: e.vop2_k(V_ADD_F32, O, K, A); break;
            case MPR_OP_ADD_LHS_RHS: e.vop2(V_ADD_F32, O, Gen::V(A), B); break;
            case MPR_OP_MUL_LHS_IMM: e.vop2_k(V_MUL_F32, O, K, A); break;
            case MPR_OP_MUL_LHS_RHS: e.vop2(V_MUL_F32, O, Gen::V(A), B); break;
            case MPR_OP_MIN_LHS_IMM: e.minmax(V_MIN_F32, choice, A, -1, K, O); break;
            case MPR_OP_MIN_LHS_RHS: e.minmax(V_MIN_F32, choice, A, B, K, O); break;
            case MPR_OP_MAX_LHS_IMM: e.minmax(V_MAX_F32, choice, A, -1, K, O); break;
            case MPR_OP_MAX_LHS_RHS: e.minmax(V_MAX_F32, choice, A, B, K, O); break;
            case MPR_OP_SUB_LHS_IMM: e.vop2_k(V_SUBREV_F32, O, K, A); break;      /* lhs - imm */
            case MPR_OP_SUB_IMM_RHS: e.vop2_k(V_SUB_F32, O, K, B); break;         /* imm - rhs */
            case MPR_OP_SUB_LHS_RHS: e.vop2(V_SUB_F32, O, Gen::V(A), B); break;
            case MPR_OP_DIV_LHS_IMM: {
                /* the translator's three cases (kernels_voxel_jit.hip: jit_translate) */
                const uint32_t ex = (K >> 23) & 0xFFu, man = K & 0x7FFFFFu, mag = K & 0x7FFFFFFFu;
                if (ex >= 1 && ex <= 253 && man == 0) {
                    e.vop2_k(V_MUL_F32, O, (K & 0x80000000u) | ((254u - ex) << 23), A);     /* +-2^-k: the same real number */
                } else if (mag >= 0x30800000u && mag <= 0x4e800000u) {
                    e.divc_inline(A, O, K);
                } else {
                    e.mov(e.c, 35, Gen::V(A));
                    e.mov_k(e.c, 36, K);
                    Gen::call(e.c, VG_RT_DIV);
                    e.mov(e.c, O, Gen::V(37));
                }
                break;
            }
            case MPR_OP_DIV_IMM_RHS:
                e.mov_k(e.c, 35, K);
                e.mov(e.c, 36, Gen::V(B));
                Gen::call(e.c, VG_RT_DIV);
                e.mov(e.c, O, Gen::V(37));
                break;
            case MPR_OP_DIV_LHS_RHS:
                e.mov(e.c, 35, Gen::V(A));
                e.mov(e.c, 36, Gen::V(B));
                Gen::call(e.c, VG_RT_DIV);
                e.mov(e.c, O, Gen::V(37));
                break;
            case MPR_OP_COPY_IMM: e.mov_k(e.c, O, K); break;
            case MPR_OP_COPY_LHS: if (O != A) e.mov(e.c, O, Gen::V(A)); break;
            case MPR_OP_COPY_RHS: if (O != B) e.mov(e.c, O, Gen::V(B)); break;
            default: return g;
        }
        if (mpr_op_is_minmax(op)) ++choice;
    }
    pos[end] = (int)e.c.size();
    e.mov(e.c, 37, Gen::V(Gen::R((int)mpr_cl_out(clauses[end]))));
    e.c.push_back(0xBE801D00u | (uint32_t)VG_RET);            /* s_setpc_b64 s[72:73] */

    /* branches: simm16 = target - (branch + 1), in dwords; the stubs start behind the code */
    const int base = (int)e.c.size();
    auto patch = [](uint32_t& w, long from, long to) -> bool {
        const long d = to - (from + 1);
        if (d < -32768 || d > 32767) return false;
        w = (w & 0xFFFF0000u) | (uint32_t)(d & 0xFFFF);
        return true;
    };
    bool fits = true;
    for (const Fix& f : guards) fits = patch(e.c[f.at], (long)f.at, pos[f.target]) && fits;
    for (const Fix& f : e.in_code) fits = patch(e.c[f.at], (long)f.at, f.target) && fits;
    for (const Fix& f : e.to_stub) fits = patch(e.c[f.at], (long)f.at, base + f.target) && fits;
    for (const Fix& f : e.to_code) fits = patch(e.st[f.at], base + (long)f.at, f.target) && fits;
    if (!fits) return g;
    g.code = std::move(e.c);
    g.code.insert(g.code.end(), e.st.begin(), e.st.end());
    g.nchoices = nch;
    g.runs = (int)runs.size();
    g.stubs = e.nstubs;
    g.ok = true;
    return g;
}

}  // namespace mpr

#ifdef MPR_TEST_HOOKS
extern "C" int mpr_test_voxel_gen(const uint64_t* clauses, int32_t len, int32_t min_run, uint32_t* out, int32_t cap, int32_t* info)
{
    const mpr::VoxelGen g = mpr::voxel_gen_build(clauses, len, min_run);
    if (!g.ok) return -1;
    if (info) {
        info[0] = g.nchoices;
        info[1] = g.runs;
        info[2] = g.stubs;
    }
    if (out && (int)g.code.size() <= cap)
        for (size_t i = 0; i < g.code.size(); ++i) out[i] = g.code[i];
    return (int)g.code.size();
}
#endif  /* MPR_TEST_HOOKS */
