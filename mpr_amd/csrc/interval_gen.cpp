/* interval_gen.cpp — see interval_gen.hpp.  tests/test_interval_gen.py assembles the text of what this file emits with the ROCm
 * assembler (the same words?), runs it through an emulator against the oracle's interval arithmetic (exact: equal; loose:
 * encloses, decides no more), and checks the wait states; tests/test_gpu_*.py run it on the chip. */
#include "interval_gen.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>

#include "../../include/mpr_clause.h"
#include "../../include/mpr_amd_test.h"
#include "gfx950_ir.hpp"
#include "tile_gen.hpp"
#include "voxel_gen.hpp"

namespace mpr {
namespace {

using namespace ir;

constexpr uint32_t SIGN = 0x80000000u;
/* fixed registers of the code (interval_gen.hpp) */
constexpr int R_OUT_LO = 36, R_OUT_HI = 37, R_ARG = 36, R_RES = 40, R_MAG = 42, R_DEC = 56;
/* tight code (interval_gen.hpp): the second result and the NaN accumulator of the values that lead to it */
constexpr int R_TOUT_LO = 38, R_TOUT_HI = 39, R_TNAN = 40;
/* |sin x - v_sin_f32(x / 2 pi)| and the float pass's own |sinf x - sin x| (include/mpr_fmath.h) together, with room to spare: the
 * instruction's error is measured on every float of the domain by tests/test_gpu_primitives.py::test_tight_sin_cos_code_on_every_float */
constexpr uint32_t TIGHT_TRIG_EPS = 0x37000000u;          /* 2^-17 */
constexpr int S_RET_ROUTINE = 36, S_RET_CODE = 38, S_BAD = 40, S_REDO = 60, S_DEC_L = 72, S_DEC_R = 74;
/* IW_FIRST_MASKS: the lanes that decided anything, the register with the LDS address of the lane's entry of the current group of 64 */
constexpr int S_ANY = 78, R_CHOICE_ADDR = 60;
/* tight code: the lanes in which an operand of the tight values left its routine's domain (no second verdict for them) */
constexpr int S_TBAD = 76;

struct IV {
    Opnd a, b;          /* exact: lo, hi; loose: -lo, hi */
};

float bits_f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
uint32_t f_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
/* 1 / c rounded down and up to float (c a finite non-zero float) */
void recip_bounds(float c, float* dn, float* up)
{
    const float r = (float)(1.0 / (double)c);
    float lo, hi;
    /* lo <= 1 / c <= hi, told by exact products in double (24 x 24 bits fit) */
    auto below = [&](float t) { const double p = (double)t * (double)c; return c > 0 ? p <= 1.0 : p >= 1.0; };   /* t <= 1 / c */
    auto above = [&](float t) { const double p = (double)t * (double)c; return c > 0 ? p >= 1.0 : p <= 1.0; };   /* t >= 1 / c */
    lo = r;
    while (!below(lo)) lo = std::nextafter(lo, -INFINITY);
    hi = r;
    while (!above(hi)) hi = std::nextafter(hi, INFINITY);
    *dn = lo;
    *up = hi;
}

struct Gen {
    std::vector<Inst> code;
    int nv = 0, ns = 0;
    int clause = 0;
    bool loose = false;
    int kind = IW_FIRST;
    int mag_next = 0;
    bool quiet = false;                 /* the tight twin of a clause: an operand outside its routine's domain or a NaN costs the lane its
                                         * second verdict (s[76:77], v40), not the wavefront its walk */
    int imposed_k = -1;                 /* the clause whose imposed decisions imp_l / imp_r hold */
    Opnd imp_l, imp_r;
    std::map<uint32_t, Opnd> kregs;     /* loose: constants kept in registers (defined in the prologue) */
    std::vector<Inst> prologue_consts;

    Opnd v() { return V(nv++); }
    Opnd s() { return S(ns++); }
    Inst& e(Op op, Opnd dst, Opnd a = NONE(), Opnd b = NONE(), Opnd c = NONE(), uint8_t neg = 0, uint8_t abs = 0, uint32_t lit = 0)
    {
        Inst i;
        i.op = op; i.dst = dst; i.src[0] = a; i.src[1] = b; i.src[2] = c; i.neg = neg; i.abs = abs; i.lit = lit; i.clause = clause;
        code.push_back(i);
        return code.back();
    }
    /* dst = op(literal, b) in the short encoding */
    Opnd lit2(Op op, uint32_t k, Opnd b) { Opnd d = v(); e(op, d, const_src(k), b, NONE(), 0, 0, k); return d; }
    Opnd op2(Op op, Opnd a, Opnd b, uint8_t neg = 0, uint8_t abs = 0) { Opnd d = v(); e(op, d, a, b, NONE(), neg, abs); return d; }
    Opnd op3(Op op, Opnd a, Opnd b, Opnd c, uint8_t neg = 0, uint8_t abs = 0) { Opnd d = v(); e(op, d, a, b, c, neg, abs); return d; }
    Opnd op1(Op op, Opnd a, uint8_t neg = 0) { Opnd d = v(); e(op, d, a, NONE(), NONE(), neg); return d; }
    Opnd cmp(Op op, Opnd a, Opnd b, uint8_t neg = 0) { Opnd d = s(); e(op, d, a, b, NONE(), neg); return d; }
    Opnd sop(Op op, Opnd a, Opnd b) { Opnd d = s(); e(op, d, a, b); return d; }
    Opnd sel(Opnd f, Opnd t, Opnd mask, uint8_t neg = 0) { Opnd d = v(); e(Op::V_CNDMASK, d, f, t, mask, neg); return d; }   /* mask ? t : f */
    Opnd movk(uint32_t k)
    {
        Opnd d = v();
        const Opnd c = const_src(k);
        e(Op::V_MOV, d, c, NONE(), NONE(), 0, 0, k);
        return d;
    }
    static Opnd const_src(uint32_t k)
    {
        if (k <= 64u) return INT((int)k);
        switch (k) {
            case 0x3f000000u: return IMM(240);
            case 0xbf000000u: return IMM(241);
            case 0x3f800000u: return IMM(242);
            case 0xbf800000u: return IMM(243);
            case 0x40000000u: return IMM(244);
            case 0xc0000000u: return IMM(245);
            case 0x40800000u: return IMM(246);
            case 0xc0800000u: return IMM(247);
            case 0x3e22f983u: return IMM(248);        /* 1 / 2 pi */
            default: return LIT();
        }
    }
    /* a constant in a register for the whole walk */
    Opnd kreg(uint32_t k)
    {
        auto it = kregs.find(k);
        if (it != kregs.end()) return it->second;
        Opnd d = v();
        Inst i;
        i.op = Op::V_MOV; i.dst = d; i.src[0] = const_src(k); i.lit = k; i.clause = 0;
        prologue_consts.push_back(i);
        kregs[k] = d;
        return d;
    }

    /* ---------------- exact arithmetic: the interpreter's sequences on renamed registers ---------------- */
    IV x_add(IV A, IV B)
    {
        Opnd t = op2(Op::V_ADD_F32, A.a, B.a, 3);
        IV o; o.b = op2(Op::V_ADD_F32, A.b, B.b);
        o.a = lit2(Op::V_XOR, SIGN, t);
        return o;
    }
    IV x_add_imm(IV A, uint32_t K)
    {
        Opnd t = lit2(Op::V_SUB_F32, K ^ SIGN, A.a);          /* (-K) - lo */
        IV o; o.b = lit2(Op::V_ADD_F32, K, A.b);
        o.a = lit2(Op::V_XOR, SIGN, t);
        return o;
    }
    IV x_sub_lhs_imm(IV A, uint32_t K)
    {
        Opnd t = lit2(Op::V_SUB_F32, K, A.a);                 /* K - lo */
        IV o; o.b = lit2(Op::V_SUBREV_F32, K, A.b);           /* hi - K */
        o.a = lit2(Op::V_XOR, SIGN, t);
        return o;
    }
    IV x_sub_imm_rhs(IV B, uint32_t K)
    {
        Opnd t = lit2(Op::V_SUBREV_F32, K, B.b);              /* hi - K */
        IV o; o.b = lit2(Op::V_SUB_F32, K, B.a);              /* K - lo */
        o.a = lit2(Op::V_XOR, SIGN, t);
        return o;
    }
    IV x_sub(IV A, IV B)
    {
        Opnd t = op2(Op::V_SUB_F32, B.b, A.a);
        IV o; o.b = op2(Op::V_SUB_F32, A.b, B.a);
        o.a = lit2(Op::V_XOR, SIGN, t);
        return o;
    }
    IV x_mul_imm(IV A, uint32_t K)
    {
        const bool ng = bits_f(K) < 0.0f;                     /* (false for -0 and NaN: the interpreter's run-time comparison) */
        Opnd k = movk(K);
        Opnd t = op2(Op::V_MUL_F32, ng ? A.b : A.a, k, 1);
        IV o; o.b = op2(Op::V_MUL_F32, ng ? A.a : A.b, k);
        o.a = lit2(Op::V_XOR, SIGN, t);
        return o;
    }
    IV x_neg(IV A)
    {
        IV o; o.a = lit2(Op::V_XOR, SIGN, A.b);
        o.b = lit2(Op::V_XOR, SIGN, A.a);
        return o;
    }
    IV x_square(IV A)                                           /* L_square */
    {
        Opnd big = cmp(Op::C_LT, A.b, A.a, 2);
        Opnd a = op2(Op::V_MUL_F32, A.a, A.a);
        Opnd b = op2(Op::V_MUL_F32, A.b, A.b);
        Opnd nc = op2(Op::V_MUL_F32, A.a, A.a, 1);
        Opnd pos = cmp(Op::C_LT, INT(0), A.a);
        Opnd ngm = cmp(Op::C_GT, INT(0), A.b);
        Opnd t = sel(b, a, big);
        Opnd lo = sel(INT(0), nc, pos, 2);
        Opnd t2 = sel(t, b, pos);
        Opnd nd = op2(Op::V_MUL_F32, A.b, A.b, 1);
        IV o; o.b = sel(t2, a, ngm);
        o.a = sel(lo, nd, ngm, 2);
        return o;
    }
    IV x_abs(IV A)                                              /* L_abs */
    {
        Opnd t42 = op2(Op::V_MAX_F32, A.b, A.b);
        Opnd t43 = op2(Op::V_MAX_F32, A.a, A.a, 3);
        Opnd m = op2(Op::V_MAX_F32, t43, t42);
        Opnd ng = cmp(Op::C_GT, INT(0), A.b);
        Opnd nn = cmp(Op::C_LE, INT(0), A.a);
        Opnd l = sel(INT(0), A.b, ng, 2);
        Opnd h = sel(m, A.a, ng, 2);
        IV o; o.a = sel(l, A.a, nn);
        o.b = sel(h, A.b, nn);
        return o;
    }
    IV x_mul(IV A, IV B)                                        /* L_mul: the sign-case table without branches */
    {
        Opnd xn = cmp(Op::C_GT, INT(0), A.a);
        Opnd nxp = cmp(Op::C_NLT, INT(0), A.b);
        Opnd nyn = cmp(Op::C_NGT, INT(0), B.a);
        Opnd yp = cmp(Op::C_LT, INT(0), B.b);
        Opnd xp = cmp(Op::C_LT, INT(0), A.b);
        Opnd xN = sop(Op::S_AND_B64, xn, nxp);
        Opnd yP = sop(Op::S_AND_B64, nyn, yp);
        Opnd nyN = sop(Op::S_OR_B64, nyn, yp);
        Opnd nxn = cmp(Op::C_NGT, INT(0), A.a);
        Opnd yn = cmp(Op::C_GT, INT(0), B.a);
        Opnd xM = sop(Op::S_AND_B64, xn, xp);
        Opnd nyN_xN = sop(Op::S_AND_B64, nyN, xN);
        Opnd xP = sop(Op::S_AND_B64, nxn, xp);
        Opnd yM = sop(Op::S_AND_B64, yn, yp);
        Opnd mp = sop(Op::S_OR_B64, yP, nyN_xN);                /* p is x.lo */
        Opnd xM_yP = sop(Op::S_AND_B64, xM, yP);
        Opnd p = sel(A.b, A.a, mp);
        Opnd mq = sop(Op::S_OR_B64, xN, xM_yP);                 /* q is y.hi */
        Opnd xP_yM = sop(Op::S_AND_B64, xP, yM);
        Opnd q = sel(B.a, B.b, mq);
        Opnd mr = sop(Op::S_OR_B64, yP, xP_yM);                 /* r is x.hi */
        Opnd r = sel(A.a, A.b, mr);
        Opnd ms = sop(Op::S_OR_B64, xP, xM_yP);                 /* s is y.hi */
        Opnd s_ = sel(B.a, B.b, ms);
        Opnd nlo = op2(Op::V_MUL_F32, p, q, 1);
        Opnd hi = op2(Op::V_MUL_F32, r, s_);
        Opnd nlo2 = op2(Op::V_MUL_F32, A.a, B.b, 1);
        Opnd hi2 = op2(Op::V_MUL_F32, A.b, B.b);
        Opnd mm = sop(Op::S_AND_B64, xM, yM);
        Opnd t46 = op2(Op::V_MAX_F32, nlo, nlo);
        Opnd t44 = op2(Op::V_MAX_F32, nlo2, nlo2);
        Opnd t44b = op2(Op::V_MAX_F32, t44, t46);
        Opnd o40 = sel(nlo, t44b, mm);
        Opnd t44c = op2(Op::V_MAX_F32, hi, hi);
        Opnd t45 = op2(Op::V_MAX_F32, hi2, hi2);
        Opnd t45b = op2(Op::V_MAX_F32, t44c, t45);
        Opnd xs = sop(Op::S_OR_B64, xn, xp);
        Opnd ys = sop(Op::S_OR_B64, yn, yp);
        Opnd o40b = lit2(Op::V_XOR, SIGN, o40);
        Opnd o41 = sel(hi, t45b, mm);
        Opnd nz = sop(Op::S_AND_B64, xs, ys);
        IV o; o.a = sel(INT(0), o40b, nz);
        o.b = sel(INT(0), o41, nz);
        return o;
    }
    /* the lanes' decisions at min / max clause k: bit k of v56 / v57 (lhs), v58 / v59 (rhs) */
    void record(Opnd not_lhs, Opnd rhs, int k)
    {
        Opnd t = sel(INT(1), INT(0), not_lhs);
        e(Op::V_LSHL_OR, PV(R_DEC + (k >> 5)), t, INT(k & 31), PV(R_DEC + (k >> 5)));
        Opnd u = sel(INT(0), INT(1), rhs);
        e(Op::V_LSHL_OR, PV(R_DEC + 2 + (k >> 5)), u, INT(k & 31), PV(R_DEC + 2 + (k >> 5)));
    }
    /* what was decided above for clause k, as lane masks (all lanes or none) */
    void imposed(int k, Opnd* all_l, Opnd* all_r)
    {
        e(Op::S_BITCMP1_B64, NONE(), PS(S_DEC_L), INT(k));
        *all_l = sop(Op::S_CSELECT_B64, IMM(C_MINUS1_I), INT(0));
        e(Op::S_BITCMP1_B64, NONE(), PS(S_DEC_R), INT(k));
        *all_r = sop(Op::S_CSELECT_B64, IMM(C_MINUS1_I), INT(0));
    }
    IV x_minmax(bool is_min, IV A, IV B, int k)                 /* L_gmin / L_gmax + L_gsel */
    {
        Opnd t42 = op2(Op::V_MAX_F32, B.a, B.a);
        Opnd t43 = op2(Op::V_MAX_F32, A.a, A.a);
        Opnd t44 = op2(Op::V_MAX_F32, B.b, B.b);
        Opnd nc1, c2a;
        if (is_min) {
            nc1 = cmp(Op::C_NLT, A.b, B.a);                     /* !c1, c1: x.hi < y.lo */
            c2a = cmp(Op::C_GT, A.a, B.b);                      /* y.hi < x.lo */
        } else {
            nc1 = cmp(Op::C_NGT, A.a, B.b);                     /* !c1, c1: x.lo > y.hi */
            c2a = cmp(Op::C_LT, A.b, B.a);                      /* y.lo > x.hi */
        }
        Opnd c2 = sop(Op::S_AND_B64, nc1, c2a);
        Opnd lo = op2(is_min ? Op::V_MIN_F32 : Op::V_MAX_F32, t43, t42);
        Opnd t43b = op2(Op::V_MAX_F32, A.b, A.b);
        Opnd hi = op2(is_min ? Op::V_MIN_F32 : Op::V_MAX_F32, t43b, t44);
        if (kind != IW_FIRST) {
            Opnd al, ar;
            imposed(k, &al, &ar);
            nc1 = sop(Op::S_ANDN2_B64, nc1, al);
            c2 = sop(Op::S_ANDN2_B64, c2, al);
            nc1 = sop(Op::S_OR_B64, nc1, ar);
            c2 = sop(Op::S_OR_B64, c2, ar);
        }
        Opnd o40 = sel(lo, B.a, c2);
        Opnd o41 = sel(hi, B.b, c2);
        IV o; o.a = sel(A.a, o40, nc1);
        o.b = sel(A.b, o41, nc1);
        record(nc1, c2, k);
        return o;
    }
    /* a routine the code calls: operands into v[36:39], result from v[40:41] (exact code only) */
    IV x_call(int routine, IV A, const IV* B)
    {
        e(Op::V_MOV, PV(R_ARG), A.a);
        e(Op::V_MOV, PV(R_ARG + 1), A.b);
        if (B) {
            e(Op::V_MOV, PV(R_ARG + 2), B->a);
            e(Op::V_MOV, PV(R_ARG + 3), B->b);
        }
        e(Op::S_SWAPPC, PS(S_RET_ROUTINE), PS(routine));
        IV o; o.a = op1(Op::V_MOV, PV(R_RES));
        o.b = op1(Op::V_MOV, PV(R_RES + 1));
        return o;
    }

    /* ---------------- loose arithmetic on (-lo, hi) ---------------- */
    /* the sum of the widths of what the products gave: a NaN as soon as one of them is (v42..v45 in turn) */
    IV nan_checked(IV o)
    {
        Opnd t = op2(Op::V_ADD_F32, o.a, o.b);
        if (quiet) {
            e(Op::V_ADD_F32, PV(R_TNAN), PV(R_TNAN), t);
            return o;
        }
        const int r = R_MAG + (mag_next++ & 3);
        e(Op::V_ADD_F32, PV(r), PV(r), t);
        return o;
    }
    void bad_if(Opnd lanes) { const int r = quiet ? S_TBAD : S_BAD; e(Op::S_OR_B64, PS(r), PS(r), lanes); }
    IV l_add(IV A, IV B)
    {
        IV o; o.a = op2(Op::V_ADD_F32, A.a, B.a);
        o.b = op2(Op::V_ADD_F32, A.b, B.b);
        return o;
    }
    IV l_add_imm(IV A, uint32_t K)              /* a + K */
    {
        IV o; o.a = k2(Op::V_ADD_F32, K ^ SIGN, A.a);
        o.b = k2(Op::V_ADD_F32, K, A.b);
        return o;
    }
    IV l_sub(IV A, IV B)                        /* [a.lo - b.hi, a.hi - b.lo] */
    {
        IV o; o.a = op2(Op::V_ADD_F32, A.a, B.b);
        o.b = op2(Op::V_ADD_F32, A.b, B.a);
        return o;
    }
    IV l_neg(IV A) { IV o; o.a = A.b; o.b = A.a; return o; }
    /* op(constant, b): the inline constant where there is one */
    Opnd k2(Op op, uint32_t k, Opnd b)
    {
        Opnd d = v();
        e(op, d, const_src(k), b, NONE(), 0, 0, k);
        return d;
    }
    IV l_mul_imm(IV A, uint32_t K)
    {
        const float kf = bits_f(K);
        const uint32_t mag_k = K & 0x7fffffffu;
        IV o;
        if (kf < 0.0f) {                        /* [hi K, lo K]: -(hi K) = hi |K|, lo K = (-lo) |K| */
            o.a = k2(Op::V_MUL_F32, mag_k, A.b);
            o.b = k2(Op::V_MUL_F32, mag_k, A.a);
        } else {
            o.a = k2(Op::V_MUL_F32, mag_k, A.a);
            o.b = k2(Op::V_MUL_F32, mag_k, A.b);
        }
        return o;
    }
    IV l_square(IV A)
    {
        Opnd c = op3(Op::V_MED3_F32, A.a, INT(0), A.b, 1);          /* the point of [lo, hi] nearest zero */
        IV o; o.a = op2(Op::V_MUL_F32, c, c, 1);                    /* -(c c), rounded up */
        Opnd m = op2(Op::V_MAX_F32, A.a, A.b, 0, 3);
        o.b = op2(Op::V_MUL_F32, m, m);
        return o;
    }
    IV l_abs(IV A)
    {
        Opnd c = op3(Op::V_MED3_F32, A.a, INT(0), A.b, 1);
        IV o; o.a = lit2(Op::V_OR, SIGN, c);                        /* -|c| */
        o.b = op2(Op::V_MAX_F32, A.a, A.b);                         /* max(-lo, hi) */
        return o;
    }
    IV l_mul_raw(IV A, IV B)
    {
        /* the four products of the ends, each rounded up, and their negations likewise */
        Opnd h1 = op2(Op::V_MUL_F32, A.a, B.a);                     /* lo lo' */
        Opnd h2 = op2(Op::V_MUL_F32, A.a, B.b, 1);                  /* lo hi' */
        Opnd h3 = op2(Op::V_MUL_F32, A.b, B.a, 1);                  /* hi lo' */
        Opnd h4 = op2(Op::V_MUL_F32, A.b, B.b);                     /* hi hi' */
        Opnd n1 = op2(Op::V_MUL_F32, A.a, B.a, 1);
        Opnd n2 = op2(Op::V_MUL_F32, A.a, B.b);
        Opnd n3 = op2(Op::V_MUL_F32, A.b, B.a);
        Opnd n4 = op2(Op::V_MUL_F32, A.b, B.b, 1);
        Opnd hm = op3(Op::V_MAX3_F32, h1, h2, h3);
        Opnd nm = op3(Op::V_MAX3_F32, n1, n2, n3);
        IV o; o.b = op2(Op::V_MAX_F32, hm, h4);
        o.a = op2(Op::V_MAX_F32, nm, n4);
        return o;
    }
    IV l_mul(IV A, IV B) { return nan_checked(l_mul_raw(A, B)); }
    IV l_const(uint32_t K)
    {
        IV o; o.a = movk(K ^ SIGN);
        o.b = movk(K);
        return o;
    }
    IV l_minmax(bool is_min, IV A, IV B, int k)
    {
        IV o;
        Opnd c1, c2;
        if (is_min) {
            o.a = op2(Op::V_MAX_F32, A.a, B.a);
            o.b = op2(Op::V_MIN_F32, A.b, B.b);
            c1 = cmp(Op::C_LT, A.b, B.a, 2);                        /* x.hi < y.lo */
            c2 = cmp(Op::C_LT, B.b, A.a, 2);                        /* y.hi < x.lo */
        } else {
            o.a = op2(Op::V_MIN_F32, A.a, B.a);
            o.b = op2(Op::V_MAX_F32, A.b, B.b);
            c1 = cmp(Op::C_GT, A.a, B.b, 1);                        /* x.lo > y.hi */
            c2 = cmp(Op::C_GT, B.a, A.b, 1);                        /* y.lo > x.hi */
        }
        if (kind != IW_FIRST && kind != IW_FIRST_MASKS) {
            /* decided above: the chosen operand as it is — the other one may never have been computed (a guarded run) */
            Opnd al, ar;
            imposed(k, &al, &ar);
            imposed_k = k; imp_l = al; imp_r = ar;
            Opnd a1 = sel(o.a, A.a, al), b1 = sel(o.b, A.b, al);
            o.a = sel(a1, B.a, ar);
            o.b = sel(b1, B.b, ar);
        }
        if (kind == IW_FIRST_MASKS) {
            /* the interpreters' record of a choice (tile_interp_asm.hpp: 16 bytes in LDS, the lanes that chose the lhs / the rhs): lane
             * k & 63 of v56..v59 takes the four halves, 64 choices go out with one store */
            e(Op::V_WRITELANE, PV(R_DEC), c1, INT(k & 63), PV(R_DEC));
            e(Op::V_WRITELANE, PV(R_DEC + 1), SH(c1.id), INT(k & 63), PV(R_DEC + 1));
            e(Op::V_WRITELANE, PV(R_DEC + 2), c2, INT(k & 63), PV(R_DEC + 2));
            e(Op::V_WRITELANE, PV(R_DEC + 3), SH(c2.id), INT(k & 63), PV(R_DEC + 3));
            e(Op::S_OR_B64, PS(S_ANY), PS(S_ANY), c1);
            e(Op::S_OR_B64, PS(S_ANY), PS(S_ANY), c2);
            if ((k & 63) == 63) flush_choices(k >> 6);
            return o;
        }
        Opnd t = sel(INT(0), INT(1), c1);
        e(Op::V_LSHL_OR, PV(R_DEC + (k >> 5)), t, INT(k & 31), PV(R_DEC + (k >> 5)));
        Opnd u = sel(INT(0), INT(1), c2);
        e(Op::V_LSHL_OR, PV(R_DEC + 2 + (k >> 5)), u, INT(k & 31), PV(R_DEC + 2 + (k >> 5)));
        return o;
    }
    /* the tight twin of a min / max clause: the enclosure of min / max (whatever this tile's wide walk decided: a sound decision
     * is a fact about the values, and the enclosure holds either way), with what was decided ABOVE imposed as the wide walk imposes
     * it — the float pass walks the tape with those decisions applied, facts or not */
    IV l_minmax_tight(bool is_min, IV A, IV B, int k)
    {
        IV o;
        if (is_min) {
            o.a = op2(Op::V_MAX_F32, A.a, B.a);
            o.b = op2(Op::V_MIN_F32, A.b, B.b);
        } else {
            o.a = op2(Op::V_MIN_F32, A.a, B.a);
            o.b = op2(Op::V_MAX_F32, A.b, B.b);
        }
        if (kind != IW_FIRST && kind != IW_FIRST_MASKS) {
            Opnd al = imp_l, ar = imp_r;
            if (imposed_k != k) { imposed(k, &al, &ar); imposed_k = k; imp_l = al; imp_r = ar; }
            Opnd a1 = sel(o.a, A.a, al), b1 = sel(o.b, A.b, al);
            o.a = sel(a1, B.a, ar);
            o.b = sel(b1, B.b, ar);
        }
        return o;
    }
    /* sin / cos of an interval (tight code only; the reference's and the loose code's enclosure is [-1, 1] whatever the argument:
     * inc/gpu_interval.hpp:353, the range reduction behind it, :355-375, is dead code).  The functions are monotone between their
     * extrema: [min, max] of the values at the ends unless a maximum (a minimum) lies inside, which the integer parts of the ends
     * in revolutions tell.  The hardware's v_sin_f32 / v_cos_f32 take revolutions; everything is padded: the ends by what the
     * product with 1 / 2 pi and the sums below can be off (so that an extremum near an end counts as inside), the values by
     * TIGHT_TRIG_EPS + |x| 2^-22 (the instruction's error, the argument's, and the float pass's own sinf / cosf against the real
     * function).  Beyond |x| = 1024 the pad grows by |x| - 1024: both extrema count as inside, the result is [-1, 1]; an end that
     * is a NaN is caught by the accumulator. */
    IV l_sincos(IV A, bool is_sin)
    {
        Opnd w = op2(Op::V_ADD_F32, A.a, A.b);
        e(Op::V_ADD_F32, PV(R_TNAN), PV(R_TNAN), w);
        Opnd nvl = lit2(Op::V_MUL_F32, 0x3e22f983u, A.a);           /* -lo / 2 pi */
        Opnd vh = lit2(Op::V_MUL_F32, 0x3e22f983u, A.b);
        Opnd am = op2(Op::V_MAX_F32, A.a, A.b, 0, 3);               /* max(|lo|, |hi|) */
        Opnd ex0 = lit2(Op::V_SUBREV_F32, 0x44800000u, am);         /* |x| - 1024 */
        Opnd ex = op2(Op::V_MAX_F32, INT(0), ex0);
        Opnd p0 = lit2(Op::V_MUL_F32, 0x34800000u, am);             /* |x| 2^-22 */
        Opnd p1 = lit2(Op::V_ADD_F32, 0x35000000u, p0);             /* + 2^-21 */
        Opnd pad = op2(Op::V_ADD_F32, p1, ex);
        Opnd E = lit2(Op::V_ADD_F32, TIGHT_TRIG_EPS, p0);
        Opnd th = op2(Op::V_ADD_F32, vh, pad);                      /* the upper end in revolutions, padded */
        Opnd tl = op2(Op::V_ADD_F32, nvl, pad);                     /* minus the lower end, padded */
        /* maxima at k + omax revolutions, minima at k + omin: one inside <=> floor(hi - o) > floor(lo - o) */
        Opnd hmax_arg, lmax_arg, hmin_arg, lmin_arg;
        if (is_sin) {                                               /* maxima at 1/4, minima at 3/4 */
            hmax_arg = lit2(Op::V_ADD_F32, 0xbe800000u, th);
            lmax_arg = lit2(Op::V_ADD_F32, 0x3e800000u, tl);
            hmin_arg = lit2(Op::V_ADD_F32, 0x3e800000u, th);
            lmin_arg = lit2(Op::V_ADD_F32, 0xbe800000u, tl);
        } else {                                                    /* maxima at 0, minima at 1/2 */
            hmax_arg = th;
            lmax_arg = tl;
            hmin_arg = op2(Op::V_ADD_F32, IMM(C_HALF), th);
            lmin_arg = op2(Op::V_ADD_F32, IMM(241), tl);
        }
        Opnd fh_max = op1(Op::V_FLOOR, hmax_arg), fl_max = op1(Op::V_FLOOR, lmax_arg, 1);
        Opnd fh_min = op1(Op::V_FLOOR, hmin_arg), fl_min = op1(Op::V_FLOOR, lmin_arg, 1);
        Opnd has_max = cmp(Op::C_LT, fl_max, fh_max);
        Opnd has_min = cmp(Op::C_LT, fl_min, fh_min);
        const Op f = is_sin ? Op::V_SIN : Op::V_COS;
        Opnd c_lo = op1(f, nvl, 1);
        Opnd c_hi = op1(f, vh);
        Opnd mx0 = op2(Op::V_MAX_F32, c_lo, c_hi);
        Opnd nm0 = op2(Op::V_MAX_F32, c_lo, c_hi, 3);
        Opnd mx1 = op2(Op::V_ADD_F32, mx0, E);
        Opnd nm1 = op2(Op::V_ADD_F32, nm0, E);
        Opnd mx2 = op2(Op::V_MIN_F32, IMM(C_ONE), mx1);
        Opnd nm2 = op2(Op::V_MIN_F32, IMM(C_ONE), nm1);
        IV o; o.b = sel(mx2, IMM(C_ONE), has_max);
        o.a = sel(nm2, IMM(C_ONE), has_min);
        return o;
    }
    void flush_choices(int group)
    {
        Inst& w = e(Op::DS_WRITE_B128, NONE(), PV(R_CHOICE_ADDR), PV(R_DEC));
        w.imm = group * 1024;      /* (an LDS store has read its data registers when the next instruction issues: no wait before they are written again) */
    }
    IV l_sqrt(IV A)
    {
        /* r = v_sqrt_f32(x) is within an ulp (two assumed): [RD(r - r 2^-22), RU(r + r 2^-22)].  A negative lower end: the float
         * pass's NaN is near -> redo.  The instruction flushes denormal arguments: harmless below, and the upper end is taken of
         * at least 2^-126 */
        bad_if(cmp(Op::C_NLE, A.a, INT(0)));                        /* !(-lo <= 0) */
        Opnd rl = op1(Op::V_SQRT, A.a, 1);
        Opnd hx = lit2(Op::V_MAX_F32, 0x00800000u, A.b);
        Opnd rh = op1(Op::V_SQRT, hx);
        const Opnd c = kreg(0x34800000u);                           /* 2^-22 */
        IV o; o.a = op3(Op::V_FMA_F32, rl, c, rl, 4);               /* RU(r c - r) = -RD(r - r c) */
        o.b = op3(Op::V_FMA_F32, rh, c, rh);
        return o;
    }
    IV l_exp(IV A)
    {
        /* tile_gen_asm.hpp: TG_FEXP_CORE on the negated lower end.  No range test: an end that overflows is an infinity (upper) or
         * the largest finite number (lower: exp(x) is beyond it), an end that underflows [0, 2^-120] */
        Opnd tn = lit2(Op::V_MUL_F32, 0x3fb8aa3bu, A.a);            /* -t_lo = RU((-lo) log2 e) */
        Opnd th = lit2(Op::V_MUL_F32, 0x3fb8aa3bu, A.b);
        Opnd rl0 = op1(Op::V_EXP, tn, 1);
        Opnd rh = op1(Op::V_EXP, th);
        Opnd rl = lit2(Op::V_MIN_F32, 0x7f7fffffu, rl0);
        Opnd kl = op2(Op::V_ADD_F32, tn, IMM(C_FOUR), 0, 1);
        Opnd kh = op2(Op::V_ADD_F32, th, IMM(C_FOUR), 0, 1);
        Opnd kl2 = lit2(Op::V_MUL_F32, 0x34000000u, kl);            /* (|t| + 4) 2^-23 */
        Opnd kh2 = lit2(Op::V_MUL_F32, 0x34000000u, kh);
        Opnd a0 = op3(Op::V_FMA_F32, rl, kl2, rl, 4);               /* RU(r k - r) (a NaN for r = 0, k = inf: dropped by the min below) */
        Opnd b0 = op3(Op::V_FMA_F32, rh, kh2, rh);
        /* results below 2^-120 (the instruction flushes, or loses bits in, what is not a normal number): [0, 2^-120] */
        Opnd a1 = lit2(Op::V_ADD_F32, 0x04800000u, a0);             /* the lower end 2^-118 lower ... */
        IV o; o.a = op2(Op::V_MIN_F32, INT(0), a1);                 /* ... and not below 0 */
        o.b = lit2(Op::V_MAX_F32, 0x03800000u, b0);
        return o;
    }
    IV l_log(IV A)
    {
        /* TG_FLOG_CORE; a lower end that is not a positive normal number -> redo (the exact routine is not isotone there) */
        bad_if(cmp(Op::C_NLE, A.a, kreg(0x80800000u)));             /* !(-lo <= -2^-126) */
        Opnd pl = op1(Op::V_LOG, A.a, 1);
        Opnd ph = op1(Op::V_LOG, A.b);
        Opnd pl2 = lit2(Op::V_MUL_F32, 0x3f317218u, pl);
        Opnd ph2 = lit2(Op::V_MUL_F32, 0x3f317218u, ph);
        Opnd el = op2(Op::V_ADD_F32, pl2, IMM(C_ONE), 0, 1);
        Opnd eh = op2(Op::V_ADD_F32, ph2, IMM(C_ONE), 0, 1);
        Opnd el2 = lit2(Op::V_MUL_F32, 0x35000000u, el);            /* (|p| + 1) 2^-21 */
        Opnd eh2 = lit2(Op::V_MUL_F32, 0x35000000u, eh);
        IV o; o.a = op2(Op::V_SUB_F32, el2, pl2);                   /* RU(e - p) */
        o.b = op2(Op::V_ADD_F32, ph2, eh2);
        return o;
    }
    /* x / c, c a constant with 2^-100 <= |c| <= 2^100: 1 / c lies in [yd, yu] (rounded on the host), both of c's sign */
    IV l_div_imm(IV A, uint32_t K)
    {
        const float c = bits_f(K);
        float yd, yu;
        recip_bounds(c, &yd, &yu);
        const uint32_t ad = f_bits(std::fabs(yd)), au = f_bits(std::fabs(yu));
        /* c > 0: -lo' = max((-lo) yd, (-lo) yu), hi' = max(hi yd, hi yu); c < 0: the ends trade places and the factors are |y| */
        const Opnd lo_src = c > 0 ? A.a : A.b, hi_src = c > 0 ? A.b : A.a;
        IV o;
        if (ad == au) {
            o.a = k2(Op::V_MUL_F32, ad, lo_src);
            o.b = k2(Op::V_MUL_F32, ad, hi_src);
        } else {
            Opnd p1 = k2(Op::V_MUL_F32, ad, lo_src), p2 = k2(Op::V_MUL_F32, au, lo_src);
            Opnd q1 = k2(Op::V_MUL_F32, ad, hi_src), q2 = k2(Op::V_MUL_F32, au, hi_src);
            o.a = op2(Op::V_MAX_F32, p1, p2);
            o.b = op2(Op::V_MAX_F32, q1, q2);
        }
        return o;
    }
    IV l_div(IV A, IV B)
    {
        /* 1 / B = [1 / b.hi, 1 / b.lo] from v_rcp_f32 (1 ulp; 2 assumed), widened, times A.  A divisor that holds zero, or an end
         * that is not a normal number (the instruction flushes it): [-inf, inf], which encloses whatever the exact walk has */
        const Opnd tiny = kreg(0x80800000u);                        /* -2^-126 */
        Opnd c1 = cmp(Op::C_LT, B.a, tiny);                         /* b.lo > 2^-126 */
        Opnd c2 = cmp(Op::C_LT, B.b, tiny);                         /* b.hi < -2^-126 */
        Opnd okm = sop(Op::S_OR_B64, c1, c2);
        Opnd y1 = op1(Op::V_RCP, B.b);                              /* 1 / b.hi: the lower end */
        Opnd y2 = op1(Op::V_RCP, B.a, 1);                           /* 1 / b.lo */
        const Opnd c = kreg(0x34800000u);
        /* (a reciprocal below 2^-126 in magnitude — a divisor beyond 2^126 — comes back as zero: 2^-126 either side covers what
         * was flushed; found by the sweep over every float, tests/test_gpu_primitives.py) */
        Opnd ra = op3(Op::V_FMA_F32, y1, c, y1, 4, 1);              /* RU(|y| c - y) */
        Opnd rb = op3(Op::V_FMA_F32, y2, c, y2, 0, 1);              /* RU(|y| c + y) */
        IV R; R.a = lit2(Op::V_ADD_F32, 0x00800000u, ra);
        R.b = lit2(Op::V_ADD_F32, 0x00800000u, rb);
        IV q = l_mul_raw(A, R);
        const Opnd inf = kreg(0x7f800000u);
        IV o; o.a = sel(inf, q.a, okm);
        o.b = sel(inf, q.b, okm);
        return nan_checked(o);
    }
};

bool uses_l(uint32_t op)
{
    return (op >= MPR_OP_SQUARE_LHS && op <= MPR_OP_LOG_LHS) || op == MPR_OP_ADD_LHS_IMM || op == MPR_OP_ADD_LHS_RHS ||
           op == MPR_OP_MUL_LHS_IMM || op == MPR_OP_MUL_LHS_RHS || mpr_op_is_minmax(op) || op == MPR_OP_SUB_LHS_IMM ||
           op == MPR_OP_SUB_LHS_RHS || op == MPR_OP_DIV_LHS_IMM || op == MPR_OP_DIV_LHS_RHS || op == MPR_OP_COPY_LHS;
}
bool uses_r(uint32_t op)
{
    return op == MPR_OP_ADD_LHS_RHS || op == MPR_OP_MUL_LHS_RHS || op == MPR_OP_MIN_LHS_RHS || op == MPR_OP_MAX_LHS_RHS ||
           op == MPR_OP_SUB_IMM_RHS || op == MPR_OP_SUB_LHS_RHS || op == MPR_OP_DIV_IMM_RHS || op == MPR_OP_DIV_LHS_RHS ||
           op == MPR_OP_COPY_RHS;
}

/* ---------------- scheduling ---------------- */
/* register keys of the dependence tracker */
constexpr int KEY_V = 0, KEY_S = 1 << 20, KEY_PV = 2 << 20, KEY_PS = 3 << 20, KEY_VCC = (3 << 20) + 106;
int key_of(const Opnd& o)
{
    switch (o.k) {
        case K::V: return KEY_V + o.id;
        case K::S: return KEY_S + o.id;
        case K::SH: return KEY_S + o.id;
        case K::PV: return KEY_PV + o.id;
        case K::PS: return KEY_PS + o.id;
        case K::VCC: return KEY_VCC;
        default: return -1;
    }
}
int issue_cycles(const Inst& i)
{
    const uint16_t f = i.flags();
    if (!i.is_valu()) return 2;
    if (f & F_TRANS) return 16;
    if (f & F_HALF) return 8;
    return 4;
}
int result_latency(const Inst& d, const Inst& u)
{
    const uint16_t f = d.flags();
    if (!d.is_valu()) return u.is_valu() ? 8 : 4;
    if (f & F_TRANS) return 28;
    if (info(d.op).fmt == Fmt::VOPC) return u.is_valu() ? 20 : 14;        /* a lane mask through the scalar registers */
    if (f & F_HALF) return 12;
    return 8;
}

struct Region {
    int begin, end;     /* instructions [begin, end) may trade places; code[end] (if any) is the barrier that closes it */
};

/* orders code[b, e) in place; returns the schedule's length in cycles */
/* (A tape of bear's size is 120 regions of 40 instructions per walk, nine walks per tape: the containers are flat and live in a scratch
 * object per thread — as std::map / vector of vectors this function was two thirds of a tape's 70 ms.  Successors are kept in the order
 * the edges were found: the order in which instructions become ready breaks ties below.) */
struct SchedScratch {
    struct Edge { int to, lat, next; };
    std::vector<Edge> edges;
    std::vector<int> head, tail, npred, prio, earliest, order, ready;
    std::vector<std::pair<int, int>> regs;        /* (key, track) sorted by key once the region's keys are known */
    std::vector<int> keys;
    std::vector<int> track_def;
    std::vector<std::vector<int>> track_readers;
    std::vector<int> scc_readers, scc_junk;
    std::vector<std::pair<int, int>> left;        /* (clause, instructions not yet issued), by clause */
    std::vector<Inst> out;
};
int schedule_region(std::vector<Inst>& code, int b, int e, int window)
{
    const int n = e - b;
    if (n <= 1) return n ? issue_cycles(code[b]) : 0;
    thread_local SchedScratch S;
    S.edges.clear();
    S.head.assign((size_t)n, -1);
    S.tail.assign((size_t)n, -1);
    S.npred.assign((size_t)n, 0);
    auto edge = [&](int from, int to, int lat) {
        if (from == to) return;
        const int id = (int)S.edges.size();
        S.edges.push_back({to, lat, -1});
        if (S.tail[(size_t)from] < 0) S.head[(size_t)from] = id;
        else S.edges[(size_t)S.tail[(size_t)from]].next = id;
        S.tail[(size_t)from] = id;
        ++S.npred[(size_t)to];
    };
    /* the registers the region names, numbered */
    S.keys.clear();
    for (int j = 0; j < n; ++j) {
        const Inst& in = code[b + j];
        for (int k = 0; k < 3; ++k) {
            const int key = key_of(in.src[k]);
            if (key >= 0) S.keys.push_back(key);
        }
        const int dk = key_of(in.dst);
        if (dk >= 0) S.keys.push_back(dk);
    }
    std::sort(S.keys.begin(), S.keys.end());
    S.keys.erase(std::unique(S.keys.begin(), S.keys.end()), S.keys.end());
    const size_t ntracks = S.keys.size();
    S.track_def.assign(ntracks, -1);
    if (S.track_readers.size() < ntracks) S.track_readers.resize(ntracks);
    for (size_t t = 0; t < ntracks; ++t) S.track_readers[t].clear();
    auto track_of = [&](int key) { return (size_t)(std::lower_bound(S.keys.begin(), S.keys.end(), key) - S.keys.begin()); };
    int scc_def = -1;
    S.scc_readers.clear();
    S.scc_junk.clear();                                          /* writers nobody reads, since the last real definition */
    for (int j = 0; j < n; ++j) {
        const Inst& in = code[b + j];
        const uint16_t f = in.flags();
        for (int k = 0; k < 3; ++k) {
            const int key = key_of(in.src[k]);
            if (key < 0) continue;
            const size_t t = track_of(key);
            if (S.track_def[t] >= 0) edge(S.track_def[t], j, result_latency(code[b + S.track_def[t]], in));
            S.track_readers[t].push_back(j);
        }
        if (f & F_RD_SCC) {
            if (scc_def >= 0) edge(scc_def, j, 4);
            S.scc_readers.push_back(j);
        }
        const int dk = key_of(in.dst);
        if (dk >= 0) {
            const size_t t = track_of(dk);
            if (S.track_def[t] >= 0) edge(S.track_def[t], j, 1);
            for (int r : S.track_readers[t]) edge(r, j, 1);
            S.track_def[t] = j;
            S.track_readers[t].clear();
        }
        if (f & F_DEF_SCC) {
            for (int r : S.scc_readers) edge(r, j, 1);
            for (int w : S.scc_junk) edge(w, j, 1);
            if (scc_def >= 0) edge(scc_def, j, 1);
            scc_def = j;
            S.scc_readers.clear();
            S.scc_junk.clear();
        } else if (f & F_WR_SCC) {
            for (int r : S.scc_readers) edge(r, j, 1);
            if (scc_def >= 0 && S.scc_readers.empty()) edge(scc_def, j, 1);      /* (never between a definition and its reader) */
            S.scc_junk.push_back(j);
        }
    }
    /* priority: the longest way to the end of the region */
    S.prio.assign((size_t)n, 0);
    for (int j = n - 1; j >= 0; --j) {
        int p = issue_cycles(code[b + j]);
        for (int id = S.head[(size_t)j]; id >= 0; id = S.edges[(size_t)id].next) p = std::max(p, S.edges[(size_t)id].lat + S.prio[(size_t)S.edges[(size_t)id].to]);
        S.prio[(size_t)j] = p;
    }
    S.earliest.assign((size_t)n, 0);
    S.order.clear();
    S.ready.clear();
    for (int j = 0; j < n; ++j)
        if (!S.npred[(size_t)j]) S.ready.push_back(j);
    /* the window: clauses in tape order; an instruction may issue while its clause is within `window` of the oldest unfinished one */
    S.left.clear();
    for (int j = 0; j < n; ++j) S.left.push_back({code[b + j].clause, 0});
    std::sort(S.left.begin(), S.left.end());
    S.left.erase(std::unique(S.left.begin(), S.left.end()), S.left.end());
    auto left_of = [&](int clause) -> int& {
        return std::lower_bound(S.left.begin(), S.left.end(), std::make_pair(clause, 0), [](const std::pair<int, int>& x, const std::pair<int, int>& y) { return x.first < y.first; })->second;
    };
    for (int j = 0; j < n; ++j) ++left_of(code[b + j].clause);
    size_t oldest_at = 0;
    int now = 0;
    while ((int)S.order.size() < n) {
        while (S.left[oldest_at].second == 0) ++oldest_at;
        const int oldest = S.left[oldest_at].first;
        int best = -1;
        bool best_now = false;
        for (int r : S.ready) {
            if (code[b + r].clause > oldest + window - 1) continue;
            const bool is_now = S.earliest[(size_t)r] <= now;
            bool better;
            if (best < 0) better = true;
            else if (is_now != best_now) better = is_now;
            else if (is_now) better = S.prio[(size_t)r] > S.prio[(size_t)best] || (S.prio[(size_t)r] == S.prio[(size_t)best] && r < best);
            else better = S.earliest[(size_t)r] < S.earliest[(size_t)best] || (S.earliest[(size_t)r] == S.earliest[(size_t)best] && S.prio[(size_t)r] > S.prio[(size_t)best]);
            if (better) { best = r; best_now = is_now; }
        }
        if (best < 0) {             /* (cannot happen: the oldest clause's next instruction is always eligible) */
            for (int r : S.ready)
                if (best < 0 || code[b + r].clause < code[b + best].clause) best = r;
        }
        now = std::max(now, S.earliest[(size_t)best]);
        S.order.push_back(best);
        S.ready.erase(std::find(S.ready.begin(), S.ready.end(), best));
        --left_of(code[b + best].clause);
        for (int id = S.head[(size_t)best]; id >= 0; id = S.edges[(size_t)id].next) {
            const int to = S.edges[(size_t)id].to;
            S.earliest[(size_t)to] = std::max(S.earliest[(size_t)to], now + S.edges[(size_t)id].lat);
            if (--S.npred[(size_t)to] == 0) S.ready.push_back(to);
        }
        now += issue_cycles(code[b + best]);
    }
    S.out.clear();
    for (int j : S.order) S.out.push_back(code[b + j]);
    std::copy(S.out.begin(), S.out.end(), code.begin() + b);
    return now;
}

/* ---------------- register allocation ---------------- */
struct Pools {
    std::vector<int> v_unsafe, v_safe;      /* vector registers a call may / may not clobber */
    std::vector<int> s_pairs;
};
Pools pools_for(bool loose, int vgpr_limit, bool big, bool tight)
{
    Pools p;
    /* (vgpr_limit: the harness that runs the code names only the vector registers below it: a wavefront with fewer registers) */
    auto range = [vgpr_limit](std::vector<int>& v, int a, int b, int step = 1) { for (int r = a; r <= b; r += step) if (step != 1 || vgpr_limit <= 0 || r < vgpr_limit) v.push_back(r); };
    if (loose && big) {
        /* IW_FIRST_MASKS: the registers tile_gen_asm.hpp: tile_gen_forward_big names (those of the interpreter with 93 slots in
         * registers: the kernel that runs it has them anyway) but the output pair, the accumulators, v56..v59 and the address in v60 */
        range(p.v_safe, 0, 35);
        range(p.v_safe, 38, 41);
        range(p.v_safe, 46, 55);
        range(p.v_safe, 64, 253);      /* (v61..v63, v254, v255: the compiler's, around the harness) */
        range(p.s_pairs, 0, 30, 2);    /* (lane masks live a few instructions: sixteen pairs are plenty, the rest stay the compiler's) */
    } else if (loose) {
        /* no calls: everything but the output pair, the magnitude accumulators and the decision words */
        range(p.v_safe, 60, 117);
        range(p.v_safe, 0, 35);
        range(p.v_safe, tight ? 41 : 38, 41);     /* (tight code: v38 / v39 the second result, v40 its NaN accumulator) */
        range(p.v_safe, 46, 55);
        range(p.s_pairs, 0, 30, 2);
        range(p.s_pairs, 42, 58, 2);
        range(p.s_pairs, 62, 70, 2);
        range(p.s_pairs, 80, 98, 2);
    } else {
        /* the compiled routines may use v0..v39, v48..v55, v64..v69 and s0..s31, the assembly ones v42..v55, s40..s59, s92..s95;
         * s68.., s82.., s96.. hold entry points */
        range(p.v_safe, 70, 117);
        range(p.v_safe, 60, 63);
        range(p.v_unsafe, 0, 31);
        range(p.v_unsafe, 64, 69);
        range(p.s_pairs, 0, 28, 2);
        range(p.s_pairs, 62, 66, 2);
        p.s_pairs.push_back(70);
        p.s_pairs.push_back(80);
    }
    return p;
}

bool allocate(std::vector<Inst>& code, int nv, int ns, bool loose, int vgpr_limit, bool big, bool tight, int* max_v, int* max_s)
{
    const int n = (int)code.size();
    std::vector<int> vdef(nv, -1), vlast(nv, -1), sdef(ns, -1), slast(ns, -1);
    std::vector<int> calls;
    std::vector<int> pv_last(256, -1);                      /* last read of a physical vector register as an operand */
    std::vector<char> pv_written(256, 0);
    for (int j = 0; j < n; ++j) {
        const Inst& in = code[j];
        if (in.flags() & F_CALL) calls.push_back(j);
        for (int k = 0; k < 3; ++k) {
            const Opnd& o = in.src[k];
            if (o.k == K::V) vlast[o.id] = j;
            else if (o.k == K::S || o.k == K::SH) slast[o.id] = j;
            else if (o.k == K::PV && !pv_written[o.id]) pv_last[o.id] = j;
        }
        if (in.dst.k == K::V) { if (vdef[in.dst.id] < 0) vdef[in.dst.id] = j; vlast[in.dst.id] = std::max(vlast[in.dst.id], j); }
        else if (in.dst.k == K::S) { if (sdef[in.dst.id] < 0) sdef[in.dst.id] = j; slast[in.dst.id] = std::max(slast[in.dst.id], j); }
        else if (in.dst.k == K::PV) pv_written[in.dst.id] = 1;
    }
    Pools pools = pools_for(loose, vgpr_limit, big, tight);
    std::vector<char> safe_reg(256, 0), pool_reg(256, 0);
    for (int r : pools.v_safe) safe_reg[r] = pool_reg[r] = 1;
    for (int r : pools.v_unsafe) pool_reg[r] = 1;
    /* live-in physical registers (the axes' intervals) are taken until their last read */
    std::vector<std::vector<int>> release_at(n + 1);
    std::vector<char> busy(256, 0);
    for (int r = 0; r < 256; ++r)
        if (pool_reg[r] && pv_last[r] >= 0) { busy[r] = 1; release_at[pv_last[r]].push_back(r); }
    /* free lists kept sorted so that the lowest register goes first (deterministic code) */
    std::vector<int> free_safe, free_unsafe, free_s;
    for (int r : pools.v_safe) if (!busy[r]) free_safe.push_back(r);
    for (int r : pools.v_unsafe) if (!busy[r]) free_unsafe.push_back(r);
    free_s = pools.s_pairs;
    auto crosses_call = [&](int d, int l) {
        auto it = std::upper_bound(calls.begin(), calls.end(), d);
        return it != calls.end() && *it < l;
    };
    std::vector<int> vphys(nv, -1), sphys(ns, -1);
    int live_v = 0, live_s = 0;
    *max_v = *max_s = 0;
    auto give_back_v = [&](int r) {
        if (!pool_reg[r]) return;
        std::vector<int>& fl = (safe_reg[r] && !loose) ? free_safe : (loose ? free_safe : free_unsafe);
        fl.push_back(r);
    };
    for (int j = 0; j < n; ++j) {
        Inst& in = code[j];
        /* sources: rename, then release those that end here */
        std::vector<int> rel_v, rel_s;
        for (int k = 0; k < 3; ++k) {
            Opnd& o = in.src[k];
            if (o.k == K::V) {
                if (vphys[o.id] < 0) return false;
                const int id = o.id;
                o = PV(vphys[id]);
                if (vlast[id] == j && std::find(rel_v.begin(), rel_v.end(), id) == rel_v.end()) rel_v.push_back(id);
            } else if (o.k == K::S || o.k == K::SH) {
                if (sphys[o.id] < 0) return false;
                const int id = o.id;
                o = PS(sphys[id] + (o.k == K::SH ? 1 : 0));
                if (slast[id] == j && std::find(rel_s.begin(), rel_s.end(), id) == rel_s.end()) rel_s.push_back(id);
            }
        }
        for (int id : rel_v) { give_back_v(vphys[id]); --live_v; }
        for (int id : rel_s) { free_s.push_back(sphys[id]); --live_s; }
        for (int r : release_at[j]) give_back_v(r);
        if (in.dst.k == K::V) {
            const int id = in.dst.id;
            if (vphys[id] < 0) {
                const bool need_safe = !loose && crosses_call(vdef[id], vlast[id]);
                int r = -1;
                std::vector<int>* fl = nullptr;
                if (!need_safe && !free_unsafe.empty()) fl = &free_unsafe;
                else if (!free_safe.empty()) fl = &free_safe;
                if (!fl) return false;
                auto it = std::min_element(fl->begin(), fl->end());
                r = *it;
                fl->erase(it);
                vphys[id] = r;
                ++live_v;
                *max_v = std::max(*max_v, live_v);
            }
            in.dst = PV(vphys[id]);
            if (vlast[id] == j) { give_back_v(vphys[id]); --live_v; }        /* a result nobody reads */
        } else if (in.dst.k == K::S) {
            const int id = in.dst.id;
            if (sphys[id] < 0) {
                if (!loose && crosses_call(sdef[id], slast[id])) return false;
                if (free_s.empty()) return false;
                auto it = std::min_element(free_s.begin(), free_s.end());
                sphys[id] = *it;
                free_s.erase(it);
                ++live_s;
                *max_s = std::max(*max_s, live_s);
            }
            in.dst = PS(sphys[id]);
            if (slast[id] == j) { free_s.push_back(sphys[id]); --live_s; }
        }
    }
    return true;
}

/* ---------------- wait states the chip does not interlock ---------------- */
/* VALU writes an SGPR / vcc -> a VALU reads it: 2 wait states; a transcendental's result -> a non-transcendental VALU: 1.
 * Along every path: a branch hands its state to its label. */
void insert_wait_states(std::vector<Inst>& code, int* nops)
{
    /* per register: the count of instructions issued so far when a VALU last wrote the scalar register / a transcendental last wrote the
     * vector register (NEVER: no such write in sight); an age is `now - written` */
    struct State {
        enum { NEVER = -1000000 };
        int now = 0;
        std::vector<int> s_at, t_at;
        State() : s_at(128, NEVER), t_at(256, NEVER) {}
        int s_age(int r) const { return now - s_at[r]; }
        int t_age(int r) const { return now - t_at[r]; }
        /* the state another way into this place brings along (taken `o.now` instructions into ITS count): the younger write of the two */
        void merge(const State& o)
        {
            const int shift = now - o.now;
            for (size_t k = 0; k < s_at.size(); ++k) s_at[k] = std::max(s_at[k], o.s_at[k] == NEVER ? (int)NEVER : o.s_at[k] + shift);
            for (size_t k = 0; k < t_at.size(); ++k) t_at[k] = std::max(t_at[k], o.t_at[k] == NEVER ? (int)NEVER : o.t_at[k] + shift);
        }
    };
    auto sreg = [](const Opnd& o) { return o.k == K::VCC ? 106 : o.k == K::PS ? o.id : -1; };
    std::vector<Inst> out;
    out.reserve(code.size() + 64);
    State st;
    std::map<int, State> at_label;
    *nops = 0;
    for (const Inst& in : code) {
        if (in.op == Op::LABEL) {
            auto it = at_label.find(in.imm);
            if (it != at_label.end()) st.merge(it->second);
            out.push_back(in);
            continue;
        }
        int need = 0;
        if (in.is_valu()) {
            for (int k = 0; k < 3; ++k) {
                const int r = sreg(in.src[k]);
                if (r >= 0) {
                    need = std::max(need, 2 - st.s_age(r));
                    if (r + 1 < 128 && in.op == Op::V_CNDMASK && k == 2) need = std::max(need, 2 - st.s_age(r + 1));
                }
                if (in.src[k].k == K::PV && !(in.flags() & F_TRANS)) need = std::max(need, 1 - st.t_age(in.src[k].id));
            }
        }
        if (need > 0) {
            Inst nop;
            nop.op = Op::S_NOP;
            nop.imm = need - 1;
            nop.clause = in.clause;
            out.push_back(nop);
            st.now += need;
            ++*nops;
        }
        out.push_back(in);
        st.now += 1;
        if (in.is_valu()) {
            const int r = sreg(in.dst);
            if (r >= 0) {
                st.s_at[r] = st.now;
                if (r + 1 < 128) st.s_at[r + 1] = st.now;
            }
            if (in.dst.k == K::PV) st.t_at[in.dst.id] = (in.flags() & F_TRANS) ? st.now : (int)State::NEVER;
        } else {
            const int r = sreg(in.dst);                 /* a scalar instruction's result needs no wait */
            if (r >= 0) {
                st.s_at[r] = State::NEVER;
                if (r + 1 < 128) st.s_at[r + 1] = State::NEVER;
            }
        }
        if (in.is_branch() && in.imm < 0) {
            const int label = -in.imm - 1;
            auto it = at_label.find(label);
            if (it == at_label.end()) at_label[label] = st;
            else {
                /* (two ways to the same label: both as seen from the later one's count) */
                State older = it->second;
                it->second = st;
                it->second.merge(older);
            }
        }
    }
    code.swap(out);
}

}  // namespace

IntervalCode interval_gen_build(const uint64_t* cl, int len, int kind, bool loose, int window, int min_run, bool keep_text, int vgpr_limit, bool report_only,
                                bool tight)
{
    IntervalCode g;
    if (!cl || len < 2 || kind < IW_FIRST || kind > IW_FIRST_MASKS) return g;
    if (kind == IW_FIRST_MASKS && !loose) return g;           /* (tapes with asin / acos / atan: the interpreter) */
    if (kind == IW_FIRST_MASKS) report_only = true;
    if (tight && (!loose || kind == IW_FIRST_MASKS)) return g;
    int end = -1, nch = 0, ntrig = 0;
    for (int i = 1; i < len; ++i) {
        const uint32_t op = mpr_cl_op(cl[i]);
        if (op == MPR_OP_INVALID) { end = i; break; }
        if (op == MPR_OP_JUMP || op >= MPR_OP_COUNT || mpr_cl_out(cl[i]) == 0) return g;
        if ((uses_l(op) && mpr_cl_lhs(cl[i]) == 0) || (uses_r(op) && mpr_cl_rhs(cl[i]) == 0)) return g;
        if (mpr_op_is_minmax(op)) ++nch;
        if (op == MPR_OP_SIN_LHS || op == MPR_OP_COS_LHS) ++ntrig;
        if (loose) {
            if (op == MPR_OP_ASIN_LHS || op == MPR_OP_ACOS_LHS || op == MPR_OP_ATAN_LHS) return g;
            if (op == MPR_OP_DIV_LHS_IMM) {
                const uint32_t m = mpr_cl_immbits(cl[i]) & 0x7fffffffu;
                if (m < 0x0D800000u || m > 0x71800000u) return g;
            }
            if (op == MPR_OP_MUL_LHS_IMM) {                       /* 0 x inf: the exact table's business */
                const uint32_t m = mpr_cl_immbits(cl[i]) & 0x7fffffffu;
                if (m == 0 || m >= 0x7f800000u) return g;
            }
        }
    }
    if (end < 0 || nch > (kind == IW_FIRST_MASKS ? IGEN_MAX_CHOICES_MASKS : IGEN_MAX_CHOICES)) return g;
    if (tight && ntrig == 0) return g;                      /* nothing to tighten */

    std::vector<DeadRun> runs;
    if (kind == IW_BELOW_GUARDED) {
        runs = tape_dead_runs(cl, end, min_run);
        std::stable_sort(runs.begin(), runs.end(), [](const DeadRun& a, const DeadRun& b) { return a.first != b.first ? a.first < b.first : a.last > b.last; });
    }
    std::vector<char> is_target((size_t)end + 2, 0);
    for (const DeadRun& r : runs) is_target[(size_t)r.last + 1] = 1;

    Gen e;
    e.loose = loose;
    e.kind = kind;
    std::vector<IV> slot(256), tslot(256);                  /* tslot / has_t: the tight value of a slot that depends on a sin / cos */
    std::vector<char> defined(256, 0), has_t(256, 0);
    /* prologue: the axes */
    e.clause = 0;
    {
        const int hs[3] = {(int)mpr_cl_out(cl[0]), (int)mpr_cl_lhs(cl[0]), (int)mpr_cl_rhs(cl[0])};
        if (loose) {
            /* hi - lo of each axis: a NaN if an end is one (or the interval is inside out at infinity) */
            e.e(Op::V_SUB_F32, PV(R_MAG), PV(1), PV(0));
            e.e(Op::V_SUB_F32, PV(R_MAG + 1), PV(3), PV(2));
            e.e(Op::V_SUB_F32, PV(R_MAG + 2), PV(5), PV(4));
            e.e(Op::V_MOV, PV(R_MAG + 3), INT(0));
            e.e(Op::S_MOV_B64, PS(S_BAD), INT(0));
            if (kind == IW_FIRST_MASKS) e.e(Op::S_MOV_B64, PS(S_ANY), INT(0));
            if (tight) {
                e.e(Op::V_MOV, PV(R_TNAN), INT(0));
                e.e(Op::S_MOV_B64, PS(S_TBAD), INT(0));
            }
        }
        for (int a = 0; a < 3; ++a) {
            IV v;
            if (loose) v.a = e.lit2(Op::V_XOR, SIGN, PV(2 * a));
            else v.a = e.op1(Op::V_MOV, PV(2 * a));
            v.b = e.op1(Op::V_MOV, PV(2 * a + 1));
            /* (a later axis in the same slot replaces an earlier one, as the interpreters' stores do) */
            slot[(size_t)hs[a]] = v;
            defined[(size_t)hs[a]] = 1;
            has_t[(size_t)hs[a]] = 0;
        }
    }
    const size_t consts_at = e.code.size();
    size_t next_run = 0;
    int choice = 0;
    for (int i = 1; i < end; ++i) {
        e.clause = i;
        if (is_target[(size_t)i]) { Inst& l = e.e(Op::LABEL, NONE()); l.imm = i; }
        for (; next_run < runs.size() && runs[next_run].first == i; ++next_run) {
            const DeadRun& r = runs[next_run];
            e.e(Op::S_BITCMP1_B64, NONE(), PS(r.by_lhs ? S_DEC_L : S_DEC_R), INT(r.choice));
            Inst& br = e.e(Op::S_CBRANCH_SCC1, NONE());
            br.imm = -(r.last + 1) - 1;
        }
        const uint64_t w = cl[i];
        const uint32_t op = mpr_cl_op(w), K = mpr_cl_immbits(w);
        const int o = (int)mpr_cl_out(w), l = (int)mpr_cl_lhs(w), r = (int)mpr_cl_rhs(w);
        /* an operand nobody wrote (a tape that reads a slot before writing it): zero, as a cleared slot file would hold.  The
         * interpreters read whatever the registers hold there; no tape the builder makes does this */
        auto get = [&](int s) -> IV {
            if (!defined[(size_t)s]) {
                IV z; z.a = e.movk(0); z.b = z.a;
                slot[(size_t)s] = z;
                defined[(size_t)s] = 1;
            }
            return slot[(size_t)s];
        };
        IV A, B, O;
        if (uses_l(op)) A = get(l);
        if (uses_r(op)) B = get(r);
        bool okc = true;
        if (loose) {
            /* twin: the clause again on the operands' tight values (tight code, clauses that depend on a sin / cos) */
            auto loose_clause = [&](IV A, IV B, bool twin, IV* O) -> bool {
                switch (op) {
                    case MPR_OP_SQUARE_LHS: *O = e.l_square(A); break;
                    case MPR_OP_SQRT_LHS: *O = e.l_sqrt(A); break;
                    case MPR_OP_NEG_LHS: *O = e.l_neg(A); break;
                    case MPR_OP_SIN_LHS:
                    case MPR_OP_COS_LHS:
                        if (twin) *O = e.l_sincos(A, op == MPR_OP_SIN_LHS);
                        else { O->a = e.movk(0x3f800000u); O->b = O->a; }
                        break;
                    case MPR_OP_EXP_LHS: *O = e.l_exp(A); break;
                    case MPR_OP_ABS_LHS: *O = e.l_abs(A); break;
                    case MPR_OP_LOG_LHS: *O = e.l_log(A); break;
                    case MPR_OP_ADD_LHS_IMM: *O = e.l_add_imm(A, K); break;
                    case MPR_OP_ADD_LHS_RHS: *O = e.l_add(A, B); break;
                    case MPR_OP_MUL_LHS_IMM: *O = e.l_mul_imm(A, K); break;
                    case MPR_OP_MUL_LHS_RHS: *O = e.l_mul(A, B); break;
                    case MPR_OP_MIN_LHS_IMM: *O = twin ? e.l_minmax_tight(true, A, e.l_const(K), choice) : e.l_minmax(true, A, e.l_const(K), choice); break;
                    case MPR_OP_MIN_LHS_RHS: *O = twin ? e.l_minmax_tight(true, A, B, choice) : e.l_minmax(true, A, B, choice); break;
                    case MPR_OP_MAX_LHS_IMM: *O = twin ? e.l_minmax_tight(false, A, e.l_const(K), choice) : e.l_minmax(false, A, e.l_const(K), choice); break;
                    case MPR_OP_MAX_LHS_RHS: *O = twin ? e.l_minmax_tight(false, A, B, choice) : e.l_minmax(false, A, B, choice); break;
                    case MPR_OP_SUB_LHS_IMM: *O = e.l_add_imm(A, K ^ SIGN); break;
                    case MPR_OP_SUB_IMM_RHS: *O = e.l_add_imm(e.l_neg(B), K); break;
                    case MPR_OP_SUB_LHS_RHS: *O = e.l_sub(A, B); break;
                    case MPR_OP_DIV_LHS_IMM: *O = e.l_div_imm(A, K); break;
                    case MPR_OP_DIV_IMM_RHS: *O = e.l_div(e.l_const(K), B); break;
                    case MPR_OP_DIV_LHS_RHS: *O = e.l_div(A, B); break;
                    case MPR_OP_COPY_IMM: *O = e.l_const(K); break;
                    case MPR_OP_COPY_LHS: *O = A; break;
                    case MPR_OP_COPY_RHS: *O = B; break;
                    default: return false;
                }
                return true;
            };
            okc = loose_clause(A, B, false, &O);
            if (okc && tight) {
                const bool tl = uses_l(op) && has_t[(size_t)l], tr = uses_r(op) && has_t[(size_t)r];
                if (tl || tr || op == MPR_OP_SIN_LHS || op == MPR_OP_COS_LHS) {
                    IV T;
                    e.quiet = true;
                    okc = loose_clause(tl ? tslot[(size_t)l] : A, tr ? tslot[(size_t)r] : B, true, &T);
                    e.quiet = false;
                    tslot[(size_t)o] = T;
                    has_t[(size_t)o] = 1;
                } else {
                    has_t[(size_t)o] = 0;
                }
            }
        } else {
            auto kiv = [&](uint32_t k) { IV c; c.a = e.movk(k); c.b = c.a; return c; };
            switch (op) {
                case MPR_OP_SQUARE_LHS: O = e.x_square(A); break;
                case MPR_OP_SQRT_LHS: O = e.x_call(TG_RT_SQRT, A, nullptr); break;
                case MPR_OP_NEG_LHS: O = e.x_neg(A); break;
                case MPR_OP_SIN_LHS:
                case MPR_OP_COS_LHS: O.a = e.movk(0xbf800000u); O.b = e.movk(0x3f800000u); break;
                case MPR_OP_ASIN_LHS: O = e.x_call(TG_RT_ASIN, A, nullptr); break;
                case MPR_OP_ACOS_LHS: O = e.x_call(TG_RT_ACOS, A, nullptr); break;
                case MPR_OP_ATAN_LHS: O = e.x_call(TG_RT_ATAN, A, nullptr); break;
                case MPR_OP_EXP_LHS: O = e.x_call(TG_RT_EXP, A, nullptr); break;
                case MPR_OP_ABS_LHS: O = e.x_abs(A); break;
                case MPR_OP_LOG_LHS: O = e.x_call(TG_RT_LOG, A, nullptr); break;
                case MPR_OP_ADD_LHS_IMM: O = e.x_add_imm(A, K); break;
                case MPR_OP_ADD_LHS_RHS: O = e.x_add(A, B); break;
                case MPR_OP_MUL_LHS_IMM: O = e.x_mul_imm(A, K); break;
                case MPR_OP_MUL_LHS_RHS: O = e.x_mul(A, B); break;
                case MPR_OP_MIN_LHS_IMM: O = e.x_minmax(true, A, kiv(K), choice); break;
                case MPR_OP_MIN_LHS_RHS: O = e.x_minmax(true, A, B, choice); break;
                case MPR_OP_MAX_LHS_IMM: O = e.x_minmax(false, A, kiv(K), choice); break;
                case MPR_OP_MAX_LHS_RHS: O = e.x_minmax(false, A, B, choice); break;
                case MPR_OP_SUB_LHS_IMM: O = e.x_sub_lhs_imm(A, K); break;
                case MPR_OP_SUB_IMM_RHS: O = e.x_sub_imm_rhs(B, K); break;
                case MPR_OP_SUB_LHS_RHS: O = e.x_sub(A, B); break;
                case MPR_OP_DIV_LHS_IMM: { IV c = kiv(K); O = e.x_call(TG_RT_DIVI, A, &c); break; }
                case MPR_OP_DIV_IMM_RHS: O = e.x_call(TG_RT_DIV, kiv(K), &B); break;
                case MPR_OP_DIV_LHS_RHS: O = e.x_call(TG_RT_DIV, A, &B); break;
                case MPR_OP_COPY_IMM: O = kiv(K); break;
                case MPR_OP_COPY_LHS: O = A; break;
                case MPR_OP_COPY_RHS: O = B; break;
                default: okc = false; break;
            }
        }
        if (!okc) return g;
        if (mpr_op_is_minmax(op)) ++choice;
        slot[(size_t)o] = O;
        defined[(size_t)o] = 1;
    }
    /* epilogue */
    e.clause = end;
    if (is_target[(size_t)end]) { Inst& l = e.e(Op::LABEL, NONE()); l.imm = end; }
    const int rs = (int)mpr_cl_out(cl[end]);
    if (!defined[(size_t)rs]) return g;
    const IV R = slot[(size_t)rs];
    const int redo_label = end + 1;
    if (kind == IW_FIRST_MASKS && (choice & 63) != 0) e.flush_choices(choice >> 6);
    if (loose) {
        Opnd m1 = e.op2(Op::V_ADD_F32, PV(R_MAG), PV(R_MAG + 1));
        Opnd m2 = e.op2(Op::V_ADD_F32, PV(R_MAG + 2), PV(R_MAG + 3));
        Opnd m3 = e.op2(Op::V_ADD_F32, m1, m2);
        Opnd big = e.cmp(Op::C_U, m3, m3);                              /* a NaN somewhere */
        e.e(Op::S_OR_B64, PS(S_BAD), PS(S_BAD), big);
        if (!report_only) {                 /* (report_only: the caller reads the lanes that ask for the exact walk from s[40:41]: tests) */
            e.e(Op::S_CMP_LG_U64, NONE(), PS(S_BAD), INT(0));
            Inst& br = e.e(Op::S_CBRANCH_SCC1, NONE());
            br.imm = -redo_label - 1;
        }
        ++e.clause;
        e.e(Op::V_XOR, PV(R_OUT_LO), Gen::const_src(SIGN), R.a, NONE(), 0, 0, SIGN);
        e.e(Op::V_MOV, PV(R_OUT_HI), R.b);
        if (tight) {
            /* the second result: the result's tight value; [-inf, inf] in the lanes whose tight values met a NaN or left a domain */
            const IV T = has_t[(size_t)rs] ? tslot[(size_t)rs] : R;
            Opnd tn = e.cmp(Op::C_U, PV(R_TNAN), PV(R_TNAN));
            Opnd kill = e.sop(Op::S_OR_B64, tn, PS(S_TBAD));
            Opnd inf = e.movk(0x7f800000u);
            Opnd ta = e.sel(T.a, inf, kill), tb = e.sel(T.b, inf, kill);
            e.e(Op::V_XOR, PV(R_TOUT_LO), Gen::const_src(SIGN), ta, NONE(), 0, 0, SIGN);
            e.e(Op::V_MOV, PV(R_TOUT_HI), tb);
        }
        e.e(Op::S_SETPC, NONE(), PS(S_RET_CODE));
        if (!report_only) {
            Inst& l = e.e(Op::LABEL, NONE());
            l.imm = redo_label;
            e.e(Op::S_SETPC, NONE(), PS(S_REDO));
        }
    } else {
        e.e(Op::V_MOV, PV(R_OUT_LO), R.a);
        e.e(Op::V_MOV, PV(R_OUT_HI), R.b);
        e.e(Op::S_SETPC, NONE(), PS(S_RET_CODE));
    }
    /* the constants kept in registers go in front of the first clause */
    e.code.insert(e.code.begin() + (long)consts_at, e.prologue_consts.begin(), e.prologue_consts.end());

    /* schedule region by region, allocate; a window the registers do not suffice for is halved */
    const std::vector<Inst> unscheduled = e.code;
    /* (default: four clauses.  Measured on the chip, scripts/walk_cycles.py: a lone wavefront issues a dependent instruction as soon
     * as an independent one — 12.2 k cycles for bear's 2500 instructions with a window of 8, 12.8 k in the tape's own order — so the
     * schedule buys the wait states it makes unnecessary and little else; a small window keeps the register pressure low) */
    int w = window > 0 ? window : 4;
    for (;; w = w > 1 ? w / 2 : 0) {
        if (w == 0) return g;
        std::vector<Inst> code = unscheduled;
        int cycles = 0;
        int b = 0;
        const int n = (int)code.size();
        for (int j = 0; j <= n; ++j) {
            if (j == n || (code[j].flags() & F_BARRIER)) {
                if (w > 1) cycles += schedule_region(code, b, j, w);
                else for (int q = b; q < j; ++q) cycles += issue_cycles(code[q]) + 4;
                /* the barrier's own inputs (a branch reads scc: its definition stays the last scalar compare) are ordered by the
                 * region's dependencies on scc below: the definition is in the region that ends here */
                b = j + 1;
            }
        }
        int mv = 0, ms = 0;
        if (!allocate(code, e.nv, e.ns, loose, vgpr_limit, kind == IW_FIRST_MASKS, tight, &mv, &ms)) {
            if (window > 0 && w == window && w == 1) return g;
            continue;
        }
        int nops = 0;
        insert_wait_states(code, &nops);
        /* labels -> offsets */
        std::map<int, int> label_at;
        int pos = 0;
        for (const Inst& in : code) {
            if (in.op == Op::LABEL) label_at[in.imm] = pos;
            pos += size_dw(in);
        }
        pos = 0;
        bool fits = true;
        for (Inst& in : code) {
            const int sz = size_dw(in);
            if (in.is_branch() && in.imm < 0) {
                auto it = label_at.find(-in.imm - 1);
                if (it == label_at.end()) return g;
                const long d = (long)it->second - (pos + 1);
                if (d < -32768 || d > 32767) fits = false;
                in.imm = (int32_t)(d & 0xFFFF);
                in.resolved = true;
            }
            pos += sz;
        }
        if (!fits) return g;
        g.words.clear();
        g.text.clear();
        g.instructions = 0;
        for (const Inst& in : code) {
            if (in.op == Op::LABEL) continue;
            if (!encode(in, g.words)) { g.words.clear(); return g; }
            if (keep_text) g.text.push_back(text(in));
            ++g.instructions;
        }
        g.nops = nops;
        g.window = w;
        g.max_vgprs = mv;
        g.max_sgpr_pairs = ms;
        g.est_cycles = cycles;
        break;
    }
    g.tight = tight;
    g.nchoices = nch;
    g.walk_words = end;
    g.result_slot = rs;
    g.ok = true;
    return g;
}

}  // namespace mpr

#ifdef MPR_TEST_HOOKS
/* test entry: the code's words and, line by line, its assembler text (lines separated by '\n' in text_out) */
extern "C" int mpr_test_interval_gen(const uint64_t* clauses, int32_t len, int32_t kind, int32_t loose, int32_t window, int32_t min_run,
                                     uint32_t* out, int32_t cap, char* text_out, int32_t text_cap, int32_t* info)
{
    /* (loose & 2: the code for the harness with 64 vector registers, as the tile stages run it; & 4: report_only; & 8: tight) */
    const mpr::IntervalCode g = mpr::interval_gen_build(clauses, len, kind, (loose & 1) != 0, window, min_run, text_out != nullptr, (loose & 2) ? ((loose & 8) ? mpr::IGEN_TIGHT_VGPRS : mpr::IGEN_LEAN_VGPRS) : 0, (loose & 4) != 0, (loose & 8) != 0);
    if (!g.ok) return -1;
    if (info) {
        info[0] = g.instructions; info[1] = g.nops; info[2] = g.window; info[3] = g.max_vgprs; info[4] = g.max_sgpr_pairs;
        info[5] = g.nchoices; info[6] = g.est_cycles;
    }
    if (out && (int)g.words.size() <= cap)
        for (size_t i = 0; i < g.words.size(); ++i) out[i] = g.words[i];
    if (text_out) {
        std::string all;
        for (const std::string& l : g.text) { all += l; all += '\n'; }
        if ((int)all.size() + 1 <= text_cap) std::memcpy(text_out, all.c_str(), all.size() + 1);
        else if (text_cap > 0) text_out[0] = 0;
    }
    return (int)g.words.size();
}
#endif  /* MPR_TEST_HOOKS */
