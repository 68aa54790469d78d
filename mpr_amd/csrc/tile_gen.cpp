/* tile_gen.cpp — see tile_gen.hpp.  Instruction encodings: GFX9 family (gfx950); tests/test_tile_gen.py disassembles what
 * this file emits with the ROCm assembler and compares it with the instructions it is meant to be. */
#include "tile_gen.hpp"
#include <future>

#include <algorithm>
#include <cstdlib>
#include <utility>

#include "../../include/mpr_clause.h"
#include "../../include/mpr_amd_test.h"
#include "gfx950_emit.hpp"
#include "internal.hpp"
#include "interval_gen.hpp"
#include "voxel_gen.hpp"

namespace mpr {
namespace {

using namespace gfx;


/* ---- backward (reference :351-458; the compiled restatement: kernels.hip, "backward walk") ---- */
/* pos -= the lanes' emit flag (VGPR ef); lanes whose chunk is full move to the next one of their run */
void take_word(Emit& e, int ef)
{
    e.vop2(V_SUB_U32, 61, Emit::V(61), ef);
    e.vopc(VC_EQ_U32, Emit::V(61), 62);
    e.cbranch_vccz(1);
    e.swappc(TG_RET_ROUTINE, TG_RT_CHUNK);                   /* clears v32 / v35 of a lane that ran out of chunks */
}
void store_word(Emit& e, int ef, uint32_t W1, uint32_t& cur_hi)
{
    if (W1 != cur_hi) {
        e.mov_lit(47, W1);
        cur_hi = W1;
    }
    e.vopc(VC_NE_U32, Emit::I(0), ef);
    e.vop2(V_LSHLREV, 44, Emit::I(3), 61);
    e.mov_exec(VCC);
    e.store_x2(44, 46, 76);
    e.mov_exec(193);                                          /* -1 */
}

void backward_clause(Emit& e, uint32_t op, int o, int l, int r, uint32_t W0, uint32_t W1, int choice, uint32_t& cur_hi)
{
    e.vop3(V3_BFE_U32, 32, Emit::V(60), Emit::I(o), Emit::I(1));                 /* is the out slot active for this lane? */
    if (!mpr_op_is_minmax(op)) {
        take_word(e, 32);
        /* active[o] = 0, active[l] = active[r] = 1 for those lanes (an operand that is the out slot: bit unchanged) */
        if (l != o && r != o) e.vop2_lit(V_AND, 60, ~(1u << o), 60);
        if (l != 0 && l != o) e.vop3(V3_LSHL_OR, 60, Emit::V(32), Emit::I(l), Emit::V(60));
        if (r != 0 && r != o && r != l) e.vop3(V3_LSHL_OR, 60, Emit::V(32), Emit::I(r), Emit::V(60));
        e.mov_lit(46, W0);
        store_word(e, 32, W1, cur_hi);
        return;
    }
    /* min / max: lanes that chose a side keep that operand only and get a COPY — or nothing, when it would copy a slot
     * onto itself */
    e.vop3(V3_BFE_U32, 33, Emit::V(56 + (choice >> 5)), Emit::I(choice & 31), Emit::I(1));   /* chose lhs */
    e.vop3(V3_BFE_U32, 34, Emit::V(58 + (choice >> 5)), Emit::I(choice & 31), Emit::I(1));   /* chose rhs */
    e.vop2(V_OR, 36, Emit::V(33), 34);
    e.vop2(V_XOR, 36, Emit::I(1), 36);
    e.vop2(V_AND, 36, Emit::V(36), 32);                                          /* undecided and active: keeps the min / max */
    e.vop2(V_ADD_U32, 54, Emit::V(54), 36);
    e.vop3(V3_LSHL_OR, 41 + (choice >> 5), Emit::V(36), Emit::I(choice & 31), Emit::V(41 + (choice >> 5)));   /* ... and says so in v41 / v42 */
    const bool dropL = l == o, dropR = r != 0 && r == o;
    if (!dropL && !dropR) {
        e.mov(35, Emit::V(32));
    } else if (dropL && dropR) {
        e.mov(35, Emit::V(36));
    } else {
        e.vop2(V_XOR, 35, Emit::I(1), dropL ? 33 : 34);
        e.vop2(V_AND, 35, Emit::V(35), 32);
    }
    take_word(e, 35);
    e.vop2(V_XOR, 37, Emit::I(1), 34);
    e.vop2(V_AND, 37, Emit::V(37), 32);                                          /* lhs stays live: active, did not choose rhs */
    if (r != 0) {
        e.vop2(V_XOR, 38, Emit::I(1), 33);
        e.vop2(V_AND, 38, Emit::V(38), 32);
    }
    e.vop2_lit(V_AND, 60, ~(1u << o), 60);
    if (l != 0) e.vop3(V3_LSHL_OR, 60, Emit::V(37), Emit::I(l), Emit::V(60));
    if (r != 0) e.vop3(V3_LSHL_OR, 60, Emit::V(38), Emit::I(r), Emit::V(60));
    e.vopc(VC_NE_U32, Emit::I(0), 33);
    e.mov_lit(46, W0);
    e.mov_lit(39, (W0 & ~0xFFu) | MPR_OP_COPY_LHS);
    e.vop2(V_CNDMASK, 46, Emit::V(46), 39);
    e.vopc(VC_NE_U32, Emit::I(0), 34);
    e.mov_lit(40, (W0 & ~0xFFu) | (r != 0 ? MPR_OP_COPY_RHS : MPR_OP_COPY_IMM));
    e.nop(0);
    e.vop2(V_CNDMASK, 46, Emit::V(46), 40);
    store_word(e, 35, W1, cur_hi);
}

/* ---- the backward walk of a FIRST stage for tapes beyond 24 slots / 64 min / max clauses (tile_gen.hpp: tile_gen_build_big_backward) ----
 * backward_clause with the active slots in three registers (slot s: bit s & 31 of v60 / v64 / v65) and the lanes' choices taken from
 * where the forward walk — the interpreter's, or the generated one that records its choices the same way (interval_gen.hpp:
 * IW_FIRST_MASKS) — left them: 16 bytes of LDS per min / max clause, the lanes that chose the lhs and the lanes that chose the rhs.
 * Every lane reads the clause's entry (v[66:69]; the read for the next min / max clause is issued as soon as this one's bits are out),
 * takes the word of its half (s[64:65] = lanes 32..63) and its own bit of it (v74 = lane & 31).  v75 = LDS address of choice 0. */
int big_active(int slot) { return slot < 32 ? 60 : slot < 64 ? 64 : 65; }
void big_read_choice(Emit& e, int choice)
{
    e.d(0xD9FE0000u | (uint32_t)(16 * choice));              /* ds_read_b128 v[66:69], v75 offset:16 * choice */
    e.d(75u | 66u << 24);
}
void big_backward_clause(Emit& e, uint32_t op, int o, int l, int r, uint32_t W0, uint32_t W1, int choice, uint32_t& cur_hi)
{
    const int AO = big_active(o), AL = big_active(l), AR = big_active(r);
    e.vop3(V3_BFE_U32, 32, Emit::V(AO), Emit::I(o & 31), Emit::I(1));            /* is the out slot active for this lane? */
    if (!mpr_op_is_minmax(op)) {
        take_word(e, 32);
        if (l != o && r != o) e.vop2_lit(V_AND, AO, ~(1u << (o & 31)), AO);
        if (l != 0 && l != o) e.vop3(V3_LSHL_OR, AL, Emit::V(32), Emit::I(l & 31), Emit::V(AL));
        if (r != 0 && r != o && r != l) e.vop3(V3_LSHL_OR, AR, Emit::V(32), Emit::I(r & 31), Emit::V(AR));
        e.mov_lit(46, W0);
        store_word(e, 32, W1, cur_hi);
        return;
    }
    e.d(0xBF8CC07Fu);                                                            /* s_waitcnt lgkmcnt(0): the clause's entry */
    e.vop3(V3_CNDMASK, 36, Emit::V(66), Emit::V(67), 64);
    e.vop3(V3_BFE_U32, 33, Emit::V(36), Emit::V(74), Emit::I(1));                /* chose lhs */
    e.vop3(V3_CNDMASK, 37, Emit::V(68), Emit::V(69), 64);
    e.vop3(V3_BFE_U32, 34, Emit::V(37), Emit::V(74), Emit::I(1));                /* chose rhs */
    if (choice > 0) big_read_choice(e, choice - 1);
    e.vop2(V_OR, 36, Emit::V(33), 34);
    e.vop2(V_XOR, 36, Emit::I(1), 36);
    e.vop2(V_AND, 36, Emit::V(36), 32);                                          /* undecided and active: keeps the min / max */
    e.vop2(V_ADD_U32, 54, Emit::V(54), 36);
    const bool dropL = l == o, dropR = r != 0 && r == o;
    if (!dropL && !dropR) {
        e.mov(35, Emit::V(32));
    } else if (dropL && dropR) {
        e.mov(35, Emit::V(36));
    } else {
        e.vop2(V_XOR, 35, Emit::I(1), dropL ? 33 : 34);
        e.vop2(V_AND, 35, Emit::V(35), 32);
    }
    take_word(e, 35);
    e.vop2(V_XOR, 37, Emit::I(1), 34);
    e.vop2(V_AND, 37, Emit::V(37), 32);                                          /* lhs stays live: active, did not choose rhs */
    if (r != 0) {
        e.vop2(V_XOR, 38, Emit::I(1), 33);
        e.vop2(V_AND, 38, Emit::V(38), 32);
    }
    e.vop2_lit(V_AND, AO, ~(1u << (o & 31)), AO);
    if (l != 0) e.vop3(V3_LSHL_OR, AL, Emit::V(37), Emit::I(l & 31), Emit::V(AL));
    if (r != 0) e.vop3(V3_LSHL_OR, AR, Emit::V(38), Emit::I(r & 31), Emit::V(AR));
    e.vopc(VC_NE_U32, Emit::I(0), 33);
    e.mov_lit(46, W0);
    e.mov_lit(39, (W0 & ~0xFFu) | MPR_OP_COPY_LHS);
    e.vop2(V_CNDMASK, 46, Emit::V(46), 39);
    e.vopc(VC_NE_U32, Emit::I(0), 34);
    e.mov_lit(40, (W0 & ~0xFFu) | (r != 0 ? MPR_OP_COPY_RHS : MPR_OP_COPY_IMM));
    e.nop(0);
    e.vop2(V_CNDMASK, 46, Emit::V(46), 40);
    store_word(e, 35, W1, cur_hi);
}

/* ---- the backward walk for tapes that are shortened again, and for the stages that shorten them ----
 * The tape a stage below the first shortens is its parent's: not the root tape under the parent's decisions, but the very
 * words the parent's walk emitted — a clause it turned into a COPY keeps its unused operand field, the reference's walk (and
 * the interpreter's) marks that operand's slot active like any plain clause's, and whichever earlier clause OF THAT TAPE writes
 * the slot is kept.  So a record carries, besides the decisions, one PRESENCE bit per clause of the root tape (bit i - 1:
 * clause i is on the tape), this code skips the rows of absent clauses (s[0..23]: the parent's bits, all set for the first
 * stage), walks the clauses decided above as the plain COPYs they are there (s[64:65] / s[66:67]), and collects the presence
 * bits of the tape it writes in v64.. (one register per 32 clauses). */
void full_store(Emit& b, int ef, uint32_t W0, uint32_t W1, int pw, int pbit)
{
    b.mov_lit(46, W0);
    b.mov_lit(47, W1);
    b.vop3(V3_LSHL_OR, 64 + pw, Emit::V(ef), Emit::I(pbit), Emit::V(64 + pw));
    b.vopc(VC_NE_U32, Emit::I(0), ef);
    b.vop2(V_LSHLREV, 44, Emit::I(3), 61);
    b.mov_exec(VCC);
    b.store_x2(44, 46, 76);
    b.mov_exec(193);
}
void full_plain(Emit& b, int o, int l, int r, uint32_t W0, uint32_t W1, int pw, int pbit)
{
    b.vop3(V3_BFE_U32, 32, Emit::V(60), Emit::I(o), Emit::I(1));
    take_word(b, 32);
    if (l != o && r != o) b.vop2_lit(V_AND, 60, ~(1u << o), 60);
    if (l != 0 && l != o) b.vop3(V3_LSHL_OR, 60, Emit::V(32), Emit::I(l), Emit::V(60));
    if (r != 0 && r != o && r != l) b.vop3(V3_LSHL_OR, 60, Emit::V(32), Emit::I(r), Emit::V(60));
    full_store(b, 32, W0, W1, pw, pbit);
}
void backward_full_clause(Emit& e, uint32_t op, int o, int l, int r, uint32_t W0, uint32_t W1, int choice, int index)
{
    const int pw = (index - 1) >> 5, pbit = (index - 1) & 31;
    std::vector<uint32_t> body;
    Emit b{body};
    if (!mpr_op_is_minmax(op)) {
        full_plain(b, o, l, r, W0, W1, pw, pbit);
    } else {
        std::vector<uint32_t> body_own, body_l, body_r;
        {
            Emit x{body_l};
            if (l != o) full_plain(x, o, l, r, (W0 & ~0xFFu) | MPR_OP_COPY_LHS, W1, pw, pbit);
        }
        {
            Emit x{body_r};
            if (!(r != 0 && r == o)) full_plain(x, o, l, r, (W0 & ~0xFFu) | (r != 0 ? MPR_OP_COPY_RHS : MPR_OP_COPY_IMM), W1, pw, pbit);
        }
        {
            Emit x{body_own};
            x.vop3(V3_BFE_U32, 32, Emit::V(60), Emit::I(o), Emit::I(1));
            x.vop3(V3_BFE_U32, 33, Emit::V(56 + (choice >> 5)), Emit::I(choice & 31), Emit::I(1));
            x.vop3(V3_BFE_U32, 34, Emit::V(58 + (choice >> 5)), Emit::I(choice & 31), Emit::I(1));
            x.vop2(V_OR, 36, Emit::V(33), 34);
            x.vop2(V_XOR, 36, Emit::I(1), 36);
            x.vop2(V_AND, 36, Emit::V(36), 32);
            x.vop2(V_ADD_U32, 54, Emit::V(54), 36);
            x.vop3(V3_LSHL_OR, 41 + (choice >> 5), Emit::V(36), Emit::I(choice & 31), Emit::V(41 + (choice >> 5)));
            const bool dropL = l == o, dropR = r != 0 && r == o;
            if (!dropL && !dropR) {
                x.mov(35, Emit::V(32));
            } else if (dropL && dropR) {
                x.mov(35, Emit::V(36));
            } else {
                x.vop2(V_XOR, 35, Emit::I(1), dropL ? 33 : 34);
                x.vop2(V_AND, 35, Emit::V(35), 32);
            }
            take_word(x, 35);
            x.vop2(V_XOR, 37, Emit::I(1), 34);
            x.vop2(V_AND, 37, Emit::V(37), 32);
            if (r != 0) {
                x.vop2(V_XOR, 38, Emit::I(1), 33);
                x.vop2(V_AND, 38, Emit::V(38), 32);
            }
            x.vop2_lit(V_AND, 60, ~(1u << o), 60);
            if (l != 0) x.vop3(V3_LSHL_OR, 60, Emit::V(37), Emit::I(l), Emit::V(60));
            if (r != 0) x.vop3(V3_LSHL_OR, 60, Emit::V(38), Emit::I(r), Emit::V(60));
            x.vopc(VC_NE_U32, Emit::I(0), 33);
            x.mov_lit(46, W0);
            x.mov_lit(39, (W0 & ~0xFFu) | MPR_OP_COPY_LHS);
            x.vop2(V_CNDMASK, 46, Emit::V(46), 39);
            x.vopc(VC_NE_U32, Emit::I(0), 34);
            x.mov_lit(40, (W0 & ~0xFFu) | (r != 0 ? MPR_OP_COPY_RHS : MPR_OP_COPY_IMM));
            x.nop(0);
            x.vop2(V_CNDMASK, 46, Emit::V(46), 40);
            x.mov_lit(47, W1);
            x.vop3(V3_LSHL_OR, 64 + pw, Emit::V(35), Emit::I(pbit), Emit::V(64 + pw));
            x.vopc(VC_NE_U32, Emit::I(0), 35);
            x.vop2(V_LSHLREV, 44, Emit::I(3), 61);
            x.mov_exec(VCC);
            x.store_x2(44, 46, 76);
            x.mov_exec(193);
        }
        /* s_bitcmp1_b64 s[64:65], k ; s_cbranch_scc1 L ; s_bitcmp1_b64 s[66:67], k ; s_cbranch_scc1 R ; own ; s_branch E ;
         * L: the COPY_LHS it is above ; s_branch E ; R: the COPY_RHS / COPY_IMM ; E: */
        const int n_own = (int)body_own.size(), n_l = (int)body_l.size(), n_r = (int)body_r.size();
        b.d(0xBF0F0040u | (uint32_t)(128 + choice) << 8);
        b.d(0xBF850000u | (uint32_t)((n_own + 3) & 0xFFFF));
        b.d(0xBF0F0042u | (uint32_t)(128 + choice) << 8);
        b.d(0xBF850000u | (uint32_t)((n_own + n_l + 2) & 0xFFFF));
        for (uint32_t w : body_own) b.d(w);
        b.d(0xBF820000u | (uint32_t)((n_l + 1 + n_r) & 0xFFFF));
        for (uint32_t w : body_l) b.d(w);
        b.d(0xBF820000u | (uint32_t)(n_r & 0xFFFF));
        for (uint32_t w : body_r) b.d(w);
    }
    e.d(0xBF0D0000u | (uint32_t)(128 + pbit) << 8 | (uint32_t)pw);          /* s_bitcmp1_b32 s[pw], pbit: on the parent's tape? */
    e.d(0xBF840000u | (uint32_t)(body.size() & 0xFFFF));                     /* s_cbranch_scc0: no */
    for (uint32_t w : body) e.d(w);
}

/* ---- the Deriv walk of the normals pass (reference :1067-1121; the interpreter's handlers: kernels_normals_asm.hip) ---- */
/* Lane = pixel * 4 + component (dx, dy, dz, value); slot s = v[50 + s]; s[98:99] = the value lanes; the shared float routines
 * take v35 (, v36) and return v37 through s[70:71] (entry points: TileGenReg TD_*).  Every row computes exactly the handler's
 * expression.  v74 / v75 (v76 / v77): bit k set = the pixel's tile decided min / max clause k for the lhs (rhs). */
struct DerivEmit {
    Emit& e;
    static int slot(int s) { return 50 + s; }
    void v2(int op, int vdst, uint32_t src0, int vsrc1) { e.vop2(op, vdst, src0, vsrc1); e.wrote(vdst); }
    void v2l(int op, int vdst, uint32_t lit, int vsrc1) { e.vop2_lit(op, vdst, lit, vsrc1); e.wrote(vdst); }
    void mv(int vdst, uint32_t src0) { e.mov(vdst, src0); e.wrote(vdst); }
    void mvl(int vdst, uint32_t lit) { e.mov_lit(vdst, lit); e.wrote(vdst); }
    void v3(int op, uint32_t dst, uint32_t s0, uint32_t s1, uint32_t s2, uint32_t neg = 0) { e.vop3(op, dst, s0, s1, s2, neg); e.wrote((int)dst); }
    void sel_value(int vdst, int other, int value) { v3(V3_CNDMASK, vdst, Emit::V(other), Emit::V(value), 98); }    /* value lanes: `value` */
    void call(int pair) { e.swappc(70, pair); e.wrote(-1); e.wrote(-1); }
    void nop(int n) { e.nop(n); e.wrote(-1); }
};
bool deriv_clause(DerivEmit& g, uint32_t op, int o, int l, int r, uint32_t K, int choice)
{
    const int O = DerivEmit::slot(o), A = DerivEmit::slot(l), B = DerivEmit::slot(r);
    Emit& e = g.e;
    switch (op) {
        case MPR_OP_SQUARE_LHS:                               /* isv ? a a : a av + a av */
            e.vop2_q3(V_MUL_F32, 38, A, A);
            g.v2(V_MUL_F32, 39, Emit::V(A), A);
            g.v2(V_ADD_F32, 38, Emit::V(38), 38);
            g.sel_value(O, 38, 39);
            break;
        case MPR_OP_SQRT_LHS:                                 /* s = sqrt(av); isv ? s : a / (2 s) */
            g.mv(43, Emit::V(A));
            e.mov_q3(35, A);
            g.call(TD_RT_SQRT);
            g.mv(44, Emit::V(37));
            g.v2(V_MUL_F32, 36, 244, 37);                     /* 2.0 */
            g.mv(35, Emit::V(43));
            g.call(TD_RT_DIV);
            g.sel_value(O, 37, 44);
            break;
        case MPR_OP_NEG_LHS: g.v2l(V_XOR, O, SIGN, A); break;
        case MPR_OP_SIN_LHS:                                  /* isv ? sin(av) : cos(av) a */
            g.mv(48, Emit::V(A));
            e.mov_q3(35, A);
            g.call(TD_RT_SINCOS);
            g.v2(V_MUL_F32, 38, Emit::V(36), 48);
            g.sel_value(O, 38, 37);
            break;
        case MPR_OP_COS_LHS:                                  /* isv ? cos(av) : -sin(av) a */
            g.mv(48, Emit::V(A));
            e.mov_q3(35, A);
            g.call(TD_RT_SINCOS);
            g.v3(V3_MUL_F32, 38, Emit::V(37), Emit::V(48), 0, 1);
            g.sel_value(O, 38, 36);
            break;
        case MPR_OP_ASIN_LHS:
        case MPR_OP_ACOS_LHS:
        case MPR_OP_ATAN_LHS:                                 /* compiled routines: a in v0, "value lane" in v1, result v37 */
            g.mv(0, Emit::V(A));
            g.v3(V3_CNDMASK, 1, Emit::I(0), Emit::I(1), 98);
            g.call(op == MPR_OP_ASIN_LHS ? TD_RT_ASIN : op == MPR_OP_ACOS_LHS ? TD_RT_ACOS : TD_RT_ATAN);
            g.mv(O, Emit::V(37));
            break;
        case MPR_OP_EXP_LHS:                                  /* e = exp(av); isv ? e : e a */
            g.mv(43, Emit::V(A));
            e.mov_q3(35, A);
            g.call(TD_RT_EXP);
            g.v2(V_MUL_F32, 38, Emit::V(37), 43);
            g.sel_value(O, 38, 37);
            break;
        case MPR_OP_ABS_LHS:                                  /* av < 0 ? -a : a */
            g.mv(39, Emit::I(0));
            g.v2l(V_XOR, 38, SIGN, A);
            e.mov_q3(46, A);
            e.vopc(VC_LT_F32, Emit::V(46), 39);
            e.wrote(-1);
            g.nop(1);
            g.v2(V_CNDMASK, O, Emit::V(A), 38);
            break;
        case MPR_OP_LOG_LHS:                                  /* isv ? log(av) : a / av */
            g.mv(43, Emit::V(A));
            e.mov_q3(35, A);
            g.nop(0);
            g.mv(45, Emit::V(35));
            g.call(TD_RT_LOG);
            g.mv(44, Emit::V(37));
            g.mv(35, Emit::V(43));
            g.mv(36, Emit::V(45));
            g.call(TD_RT_DIV);
            g.sel_value(O, 37, 44);
            break;
        case MPR_OP_ADD_LHS_IMM:                              /* only the value */
            g.v2l(V_ADD_F32, 38, K, A);
            g.sel_value(O, A, 38);
            break;
        case MPR_OP_ADD_LHS_RHS: g.v2(V_ADD_F32, O, Emit::V(A), B); break;
        case MPR_OP_MUL_LHS_IMM: g.v2l(V_MUL_F32, O, K, A); break;
        case MPR_OP_MUL_LHS_RHS:                              /* isv ? a b : a bv + b av */
            e.vop2_q3(V_MUL_F32, 38, B, A);
            e.vop2_q3(V_MUL_F32, 39, A, B);
            g.v2(V_MUL_F32, 40, Emit::V(A), B);
            g.v2(V_ADD_F32, 38, Emit::V(38), 39);
            g.sel_value(O, 38, 40);
            break;
        case MPR_OP_MIN_LHS_IMM:
        case MPR_OP_MIN_LHS_RHS:
        case MPR_OP_MAX_LHS_IMM:
        case MPR_OP_MAX_LHS_RHS: {
            const bool is_min = op <= MPR_OP_MIN_LHS_RHS;
            const bool has_rhs = op == MPR_OP_MIN_LHS_RHS || op == MPR_OP_MAX_LHS_RHS;
            int Bv = B;
            if (has_rhs) {
                e.mov_q3(39, B);                              /* bv */
            } else {
                g.mvl(39, K);                                 /* b = (0, 0, 0, imm) */
                g.v3(V3_CNDMASK, 36, Emit::I(0), Emit::V(39), 98);
                Bv = 36;
            }
            e.mov_q3(46, A);                                  /* av */
            e.vopc(is_min ? VC_LT_F32 : VC_GE_F32, Emit::V(46), 39);          /* vcc: take a */
            e.wrote(-1);
            /* the tile's decision overrides the comparison */
            g.v3(V3_BFE_U32, 40, Emit::V(74 + (choice >> 5)), Emit::I(choice & 31), Emit::I(1));
            g.v3(V3_BFE_U32, 41, Emit::V(76 + (choice >> 5)), Emit::I(choice & 31), Emit::I(1));
            e.vop3(0xCD, 42, Emit::I(0), Emit::V(40), 0);     /* v_cmp_ne_u32 s[42:43], 0, v40 */
            e.vop3(0xCD, 44, Emit::I(0), Emit::V(41), 0);
            e.wrote(-1);
            e.wrote(-1);
            g.nop(0);
            e.sop2_vcc(15, 42);                               /* vcc |= decided lhs */
            e.sop2_vcc(19, 44);                               /* vcc &= ~decided rhs */
            e.wrote(-1);
            e.wrote(-1);
            g.v2(V_CNDMASK, O, Emit::V(Bv), A);
            break;
        }
        case MPR_OP_SUB_LHS_IMM:                              /* only the value */
            g.v2l(V_SUBREV_F32, 38, K, A);
            g.sel_value(O, A, 38);
            break;
        case MPR_OP_SUB_IMM_RHS:                              /* isv ? imm - b : -b */
            g.v2l(V_SUB_F32, 38, K, B);
            g.v2l(V_XOR, 39, SIGN, B);
            g.sel_value(O, 39, 38);
            break;
        case MPR_OP_SUB_LHS_RHS: g.v2(V_SUB_F32, O, Emit::V(A), B); break;
        case MPR_OP_DIV_LHS_IMM:                              /* every component by imm */
            g.mv(35, Emit::V(A));
            g.mvl(36, K);
            g.call(TD_RT_DIV);
            g.mv(O, Emit::V(37));
            break;
        case MPR_OP_DIV_IMM_RHS:                              /* isv ? imm / b : (-imm b) / (bv bv) */
            e.mov_q3(41, B);
            g.mvl(40, K);
            g.v3(V3_MUL_F32, 39, Emit::V(40), Emit::V(B), 0, 1);
            g.v2(V_MUL_F32, 38, Emit::V(41), 41);
            g.v3(V3_CNDMASK, 35, Emit::V(39), Emit::V(40), 98);
            g.v3(V3_CNDMASK, 36, Emit::V(38), Emit::V(B), 98);
            g.call(TD_RT_DIV);
            g.mv(O, Emit::V(37));
            break;
        case MPR_OP_DIV_LHS_RHS:                              /* isv ? a / b : (bv a - av b) / (bv bv) */
            e.vop2_q3(V_MUL_F32, 38, B, A);
            e.vop2_q3(V_MUL_F32, 39, A, B);
            e.mov_q3(41, B);
            g.v2(V_SUB_F32, 38, Emit::V(38), 39);
            g.nop(0);
            g.v2(V_MUL_F32, 40, Emit::V(41), 41);
            g.v3(V3_CNDMASK, 35, Emit::V(38), Emit::V(A), 98);
            g.v3(V3_CNDMASK, 36, Emit::V(40), Emit::V(B), 98);
            g.call(TD_RT_DIV);
            g.mv(O, Emit::V(37));
            break;
        case MPR_OP_COPY_IMM:
            g.mvl(39, K);
            g.v3(V3_CNDMASK, O, Emit::I(0), Emit::V(39), 98);
            break;
        case MPR_OP_COPY_LHS: if (o != l) g.mv(O, Emit::V(A)); break;
        case MPR_OP_COPY_RHS: if (o != r) g.mv(O, Emit::V(B)); break;
        default: return false;
    }
    return true;
}

}  // namespace

TileGen tile_gen_build(const uint64_t* clauses, int len)
{
    TileGen g;
    if (!clauses || len < 2) return g;
    int end = -1, nch = 0;
    for (int i = 1; i < len; ++i) {
        const uint32_t op = (uint32_t)clauses[i] & 0xFF;
        const int o = (int)(clauses[i] >> 8) & 0xFF, l = (int)(clauses[i] >> 16) & 0xFF, r = (int)(clauses[i] >> 24) & 0xFF;
        if (o >= TILE_GEN_MAX_SLOTS || l >= TILE_GEN_MAX_SLOTS || r >= TILE_GEN_MAX_SLOTS) return g;
        if (op == MPR_OP_INVALID) {
            end = i;
            break;
        }
        if (op == MPR_OP_JUMP || op >= MPR_OP_COUNT || o == 0) return g;
        if (mpr_op_is_minmax(op)) ++nch;
    }
    if (end < 0 || nch > TILE_GEN_MAX_CHOICES) return g;
    {
        const int hx = (int)(clauses[0] >> 8) & 0xFF, hy = (int)(clauses[0] >> 16) & 0xFF, hz = (int)(clauses[0] >> 24) & 0xFF;
        if (hx >= TILE_GEN_MAX_SLOTS || hy >= TILE_GEN_MAX_SLOTS || hz >= TILE_GEN_MAX_SLOTS) return g;
    }
    Emit b{g.bwd};
    int choice = nch;
    g.result_slot = (int)(clauses[end] >> 8) & 0xFF;
    for (int i = 1; i < end; ++i) {
        const uint32_t op = (uint32_t)clauses[i] & 0xFF;
        if (op == MPR_OP_INVALID || op == MPR_OP_JUMP || op >= MPR_OP_COUNT) return g;
    }

    {
        Emit de{g.deriv};
        DerivEmit dg{de};
        int ch = 0;
        for (int i = 1; i < end; ++i) {
            const uint64_t w = clauses[i];
            const uint32_t op = (uint32_t)w & 0xFF;
            if (!deriv_clause(dg, op, (int)(w >> 8) & 0xFF, (int)(w >> 16) & 0xFF, (int)(w >> 24) & 0xFF, (uint32_t)(w >> 32), ch)) {
                g.deriv.clear();
                return g;
            }
            if (mpr_op_is_minmax(op)) ++ch;
        }
        de.mov(37, Emit::V(DerivEmit::slot(g.result_slot)));
        de.setpc(TG_RET_CODE);
    }
    {
        /* ... and with the dead runs guarded (the 16 pixels of a footprint mostly share their tiles' decisions) */
        std::vector<DeadRun> runs = tape_dead_runs(clauses, end, 4);
        std::stable_sort(runs.begin(), runs.end(), [](const DeadRun& a, const DeadRun& b) { return a.first != b.first ? a.first < b.first : a.last > b.last; });
        Emit de{g.deriv_guarded};
        DerivEmit dg{de};
        std::vector<int> pos((size_t)end + 1, 0);
        std::vector<std::pair<size_t, int>> fix;
        size_t next_run = 0;
        int ch = 0;
        for (int i = 1; i < end; ++i) {
            pos[(size_t)i] = (int)g.deriv_guarded.size();
            for (; next_run < runs.size() && runs[next_run].first == i; ++next_run) {
                const DeadRun& r = runs[next_run];
                de.d(0xBF0F0000u | (uint32_t)(128 + r.choice) << 8 | (r.by_lhs ? 64u : 66u));      /* s_bitcmp1_b64 s[64:65] / s[66:67], choice */
                fix.emplace_back(g.deriv_guarded.size(), r.last + 1);
                de.d(0xBF850000u);                                   /* s_cbranch_scc1 */
                de.wrote(-1);                                        /* (a jump target: no DPP hazard is carried across it) */
                de.wrote(-1);
            }
            const uint64_t w = clauses[i];
            const uint32_t op = (uint32_t)w & 0xFF;
            (void)deriv_clause(dg, op, (int)(w >> 8) & 0xFF, (int)(w >> 16) & 0xFF, (int)(w >> 24) & 0xFF, (uint32_t)(w >> 32), ch);
            if (mpr_op_is_minmax(op)) ++ch;
        }
        pos[(size_t)end] = (int)g.deriv_guarded.size();
        de.mov(37, Emit::V(DerivEmit::slot(g.result_slot)));
        de.setpc(TG_RET_CODE);
        bool fits = true;
        for (const auto& fx : fix) {
            const long d = (long)pos[(size_t)fx.second] - ((long)fx.first + 1);
            if (d < 0 || d > 32767) fits = false;
            g.deriv_guarded[fx.first] |= (uint32_t)(d & 0xFFFF);
        }
        if (!fits || runs.empty()) g.deriv_guarded.clear();
    }

    uint32_t cur_hi = 0;                                      /* the harness enters with v47 = 0 */
    for (int i = end - 1; i >= 1; --i) {
        const uint64_t w = clauses[i];
        const uint32_t op = (uint32_t)w & 0xFF;
        if (mpr_op_is_minmax(op)) --choice;
        backward_clause(b, op, (int)(w >> 8) & 0xFF, (int)(w >> 16) & 0xFF, (int)(w >> 24) & 0xFF, (uint32_t)w, (uint32_t)(w >> 32), choice, cur_hi);
    }
    b.setpc(TG_RET_CODE);
    if (end - 1 <= 32 * TILE_GEN_PRESENCE_WORDS) {
        Emit bf{g.bwd_full};
        int ch = nch;
        for (int i = end - 1; i >= 1; --i) {
            const uint64_t w = clauses[i];
            const uint32_t op = (uint32_t)w & 0xFF;
            if (mpr_op_is_minmax(op)) --ch;
            backward_full_clause(bf, op, (int)(w >> 8) & 0xFF, (int)(w >> 16) & 0xFF, (int)(w >> 24) & 0xFF, (uint32_t)w, (uint32_t)(w >> 32), ch, i);
        }
        bf.setpc(TG_RET_CODE);
    }
    g.words = end;
    g.nchoices = nch;
    g.ok = true;
    return g;
}

std::vector<uint32_t> tile_gen_build_big_backward(const uint64_t* clauses, int len, int* nchoices)
{
    std::vector<uint32_t> out;
    if (!clauses || len < 2) return out;
    int end = -1, nch = 0;
    for (int i = 1; i < len; ++i) {
        const uint64_t w = clauses[i];
        const uint32_t op = (uint32_t)w & 0xFF;
        if (op == MPR_OP_INVALID) { end = i; break; }
        if (op == MPR_OP_JUMP || op >= MPR_OP_COUNT) return out;
        const int o = (int)(w >> 8) & 0xFF, l = (int)(w >> 16) & 0xFF, r = (int)(w >> 24) & 0xFF;
        if (o == 0 || o >= 96 || l >= 96 || r >= 96) return out;
        if (mpr_op_is_minmax(op)) ++nch;
    }
    if (end < 0 || nch > 4096 || ((int)(clauses[end] >> 8) & 0xFF) >= 96) return out;
    Emit b{out};
    if (nch > 0) big_read_choice(b, nch - 1);
    int choice = nch;
    uint32_t cur_hi = 0;                                      /* the harness enters with v47 = 0 */
    for (int i = end - 1; i >= 1; --i) {
        const uint64_t w = clauses[i];
        const uint32_t op = (uint32_t)w & 0xFF;
        if (mpr_op_is_minmax(op)) --choice;
        big_backward_clause(b, op, (int)(w >> 8) & 0xFF, (int)(w >> 16) & 0xFF, (int)(w >> 24) & 0xFF, (uint32_t)w, (uint32_t)(w >> 32), choice, cur_hi);
    }
    b.setpc(TG_RET_CODE);
    if (nchoices) *nchoices = nch;
    return out;
}

}  // namespace mpr

namespace mpr {
std::shared_ptr<const TapeCode> build_tape_code(const uint64_t* clauses, int len, int vox_min_run)
{
    const TileGen g = tile_gen_build(clauses, len);
    if (!g.ok || g.deriv.empty()) return nullptr;
    const VoxelGen v = voxel_gen_build(clauses, len, vox_min_run);
    auto c = std::make_shared<TapeCode>();
    c->words.reserve(g.bwd.size() + g.deriv.size() + g.bwd_full.size() + v.code.size() + g.deriv_guarded.size() + 40000);
    c->words.insert(c->words.end(), g.bwd.begin(), g.bwd.end());
    c->words.insert(c->words.end(), g.deriv.begin(), g.deriv.end());
    c->words.insert(c->words.end(), g.bwd_full.begin(), g.bwd_full.end());
    if (v.ok) c->words.insert(c->words.end(), v.code.begin(), v.code.end());
    c->words.insert(c->words.end(), g.deriv_guarded.begin(), g.deriv_guarded.end());
    c->derivg_dw = (int)g.deriv_guarded.size();
    c->fwd_dw = 0;                  /* (round 4's forward walk stood here; the forward walks are the iw_* pieces behind the rest) */
    c->bwd_dw = (int)g.bwd.size();
    c->deriv_dw = (int)g.deriv.size();
    c->full_dw = (int)g.bwd_full.size();
    c->vox_dw = v.ok ? (int)v.code.size() : 0;
    c->walk_words = g.words;
    c->nchoices = g.nchoices;
    c->vox_min_run = vox_min_run;
    const int window = 0;
    /* the nine scheduled forward walks — three kinds, each exact, loose and loose + tight — are independent of one another and 3 to 8 ms
     * each for a tape of bear's size (list scheduling, register allocation, wait states): side by side, put together in a fixed order */
    std::future<IntervalCode> walks[3][3];
    for (int kind = 0; kind < 3; ++kind)
        for (int loose = 0; loose < 3; ++loose)           /* (2: loose and tight) */
            walks[kind][loose] = std::async(std::launch::async, [=]() {
                return interval_gen_build(clauses, len, kind, loose != 0, window, 3, false, loose == 2 ? IGEN_TIGHT_VGPRS : loose ? IGEN_LEAN_VGPRS : 0, false, loose == 2);
            });
    for (int kind = 0; kind < 3; ++kind)
        for (int loose = 0; loose < 3; ++loose) {
            const IntervalCode ic = walks[kind][loose].get();
            if (!ic.ok) continue;
            c->iw_at[kind][loose] = (int)c->words.size();
            c->iw_dw[kind][loose] = (int)ic.words.size();
            c->iw_instructions[kind][loose] = ic.instructions;
            c->words.insert(c->words.end(), ic.words.begin(), ic.words.end());
        }
    return c;
}
}  // namespace mpr

#ifdef MPR_TEST_HOOKS
extern "C" int mpr_test_tile_gen(const uint64_t* clauses, int32_t len, int32_t which, uint32_t* out, int32_t cap)
{
    if (which == 6) {                    /* the backward walk for tapes of up to 96 slots (tile_gen_build_big_backward) */
        const std::vector<uint32_t> c = mpr::tile_gen_build_big_backward(clauses, len);
        if (c.empty()) return -1;
        if (out && (int)c.size() <= cap)
            for (size_t i = 0; i < c.size(); ++i) out[i] = c[i];
        return (int)c.size();
    }
    const mpr::TileGen g = mpr::tile_gen_build(clauses, len);
    if (!g.ok || which < 1 || which > 5 || which == 4) return -1;        /* (0 / 4 were round 4's forward walks: mpr_test_interval_gen) */
    const std::vector<uint32_t>& c = which == 5 ? g.deriv_guarded : which == 3 ? g.bwd_full : which == 2 ? g.deriv : g.bwd;
    if (out && (int)c.size() <= cap)
        for (size_t i = 0; i < c.size(); ++i) out[i] = c[i];
    return (int)c.size();
}
#endif  /* MPR_TEST_HOOKS */
