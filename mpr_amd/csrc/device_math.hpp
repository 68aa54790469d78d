/*
 * device_math.hpp — per-lane arithmetic of the three evaluators on gfx950:
 *   Interval (outward rounded)   reference: inc/gpu_interval.hpp:18-391
 *   float                        reference: src/context.cu:882-921
 *   Deriv (forward-mode AD)      reference: inc/gpu_deriv.hpp:18-204
 *
 * Directed rounding on CDNA4.  CUDA's __fadd_rd/_ru etc. have no fast HIP equivalent, so
 * the interval kernels run with MODE.fp_round[1:0] (the f32 field) set to round-toward-
 * +infinity for their whole interval section and obtain every lower bound through
 * negation:  RD(a+b) = -RU(-a-b),  RD(a*b) = -RU((-a)*b)  (exact IEEE symmetry, signed
 * zeros included).  All rounded f32 operations in that section are inline-asm VALU
 * instructions, so the compiler can neither fold, re-associate nor move them across the
 * s_setreg; no compiler-generated f32 arithmetic is allowed in that section.
 * Division and square root have no directed hardware form: their correctly rounded
 * round-to-nearest results are computed inside a short RN "sandwich" (operands and results
 * are threaded through the two s_setreg statements, which pins the code between them) and
 * moved by one ulp according to the sign of the exact FMA residual.  The f64 field of
 * fp_round is never touched: the double-precision libm calls inside acos/asin/atan/exp/log
 * (OCML, f64 instructions only) always run in round-to-nearest, like ::acos etc. on CUDA.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mpr_clause.h"
#include "../../include/mpr_fmath.h"

#define DEV __device__ __forceinline__

namespace mprk {

struct ival { float lo, hi; };
DEV ival iv(float lo, float hi) { ival r; r.lo = lo; r.hi = hi; return r; }

/* ---- rounding-mode control ---------------------------------------------------------- */
/* enter the round-up section; values computed in RN before it are threaded through */
DEV void round_up_begin(float& a, float& b, float& c, float& d, float& e, float& f)
{
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 1"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
}
DEV void round_up_begin() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 1"); }
DEV void round_nearest_begin() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"); }

/* ---- RU primitives (round-up mode in effect) ---------------------------------------- */
DEV float ru_add(float a, float b) { float r; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
DEV float ru_sub(float a, float b) { float r; asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
DEV float ru_mul(float a, float b) { float r; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
/* -RU(-a-b), -RU(b-a), -RU((-a)*b): round-down results */
DEV float rd_add(float a, float b) { float r; asm volatile("v_add_f32_e64 %0, -%1, -%2" : "=v"(r) : "v"(a), "v"(b)); return -r; }
DEV float rd_sub(float a, float b) { float r; asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(b), "v"(a)); return -r; }
DEV float rd_mul(float a, float b) { float r; asm volatile("v_mul_f32_e64 %0, -%1, %2" : "=v"(r) : "v"(a), "v"(b)); return -r; }

DEV float next_up(float x)
{   /* smallest float > x; NaN and +inf unchanged */
    const uint32_t u = mpr_f2u(x);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u || u == 0x7F800000u) return x;
    if ((u & 0x7FFFFFFFu) == 0) return mpr_u2f(1u);
    return mpr_u2f((u & 0x80000000u) ? u - 1 : u + 1);
}
DEV float next_down(float x)
{
    const uint32_t u = mpr_f2u(x);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u || u == 0xFF800000u) return x;
    if ((u & 0x7FFFFFFFu) == 0) return mpr_u2f(0x80000001u);
    return mpr_u2f((u & 0x80000000u) ? u + 1 : u - 1);
}
DEV bool is_finite(float x) { return (mpr_f2u(x) & 0x7F800000u) != 0x7F800000u; }

/* round-to-nearest quotient + exact residual, evaluated in an RN sandwich */
DEV void rn_div2(float a1, float b1, float a2, float b2, float& q1, float& r1, float& q2, float& r2)
{
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0" : "+v"(a1), "+v"(b1), "+v"(a2), "+v"(b2));
    q1 = a1 / b1;
    r1 = __builtin_fmaf(-q1, b1, a1);
    q2 = a2 / b2;
    r2 = __builtin_fmaf(-q2, b2, a2);
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 1" : "+v"(q1), "+v"(r1), "+v"(q2), "+v"(r2));
}
/* move a correctly rounded quotient q = RN(a/b) (residual r = a - q*b) down / up */
DEV float div_fix_down(float q, float r, float a, float b)
{
    if (is_finite(q)) {
        /* exact quotient = q + r/b;  below q  <=>  r/b < 0 */
        const bool below = (r < 0.0f && b > 0.0f) || (r > 0.0f && b < 0.0f);
        return below ? next_down(q) : q;
    }
    /* RN overflowed although the exact quotient is finite: RD of a positive overflow is FLT_MAX */
    if (q > 0.0f && is_finite(a) && is_finite(b) && b != 0.0f) return mpr_u2f(0x7F7FFFFFu);
    return q;
}
DEV float div_fix_up(float q, float r, float a, float b)
{
    if (is_finite(q)) {
        const bool above = (r > 0.0f && b > 0.0f) || (r < 0.0f && b < 0.0f);
        return above ? next_up(q) : q;
    }
    if (q < 0.0f && is_finite(a) && is_finite(b) && b != 0.0f) return mpr_u2f(0xFF7FFFFFu);
    return q;
}
/* double -> float, directed; independent of the current f32 rounding mode */
DEV float d2f_rd(double d)
{
    const float f = (float)d;
    return ((double)f > d) ? next_down(f) : f;
}
DEV float d2f_ru(double d)
{
    const float f = (float)d;
    return ((double)f < d) ? next_up(f) : f;
}
/* The residual test above needs a - q b (a - s s) to be representable: it is a multiple of
 * ulp(q) ulp(b), which drops below the smallest subnormal once |a| < 2^-103, and a residual that
 * rounds to zero says "exact".  Those operands go through double precision instead: a quotient
 * (root) of 24-bit numbers that is not itself a 24-bit number is at least 2^-48 (2^-49) away from
 * one in relative terms, far more than the 2^-53 the double result is off by, so directed rounding
 * of the double gives the directed rounding of the exact value.  MODE.fp_round's double-precision
 * field stays round-to-nearest throughout. */
DEV bool tiny_nonzero(float a) { const uint32_t m = mpr_f2u(a) & 0x7FFFFFFFu; return m != 0 && m < 0x12800000u; }   /* < 2^-90 */
DEV ival div_dir_exact(float a1, float b1, float a2, float b2)
{
    return iv(d2f_rd((double)a1 / (double)b1), d2f_ru((double)a2 / (double)b2));
}
DEV ival sqrt_dir_exact(float a, float b)
{
    return iv(d2f_rd(__builtin_sqrt((double)a)), d2f_ru(__builtin_sqrt((double)b)));
}
/* {RD(a1/b1), RU(a2/b2)} */
DEV ival div_dir(float a1, float b1, float a2, float b2)
{
    if (tiny_nonzero(a1) || tiny_nonzero(a2)) return div_dir_exact(a1, b1, a2, b2);
    float q1, r1, q2, r2;
    rn_div2(a1, b1, a2, b2, q1, r1, q2, r2);
    return iv(div_fix_down(q1, r1, a1, b1), div_fix_up(q2, r2, a2, b2));
}
/* {RD(sqrt(a)), RU(sqrt(b))} */
DEV ival sqrt_dir(float a, float b)
{
    if (tiny_nonzero(a) || tiny_nonzero(b)) return sqrt_dir_exact(a, b);
    float s1, r1, s2, r2;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0" : "+v"(a), "+v"(b));
    s1 = __builtin_sqrtf(a);
    r1 = __builtin_fmaf(-s1, s1, a);
    s2 = __builtin_sqrtf(b);
    r2 = __builtin_fmaf(-s2, s2, b);
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 1" : "+v"(s1), "+v"(r1), "+v"(s2), "+v"(r2));
    const float lo = (is_finite(s1) && r1 < 0.0f) ? next_down(s1) : s1;
    const float hi = (is_finite(s2) && r2 > 0.0f) ? next_up(s2) : s2;
    return iv(lo, hi);
}
/* ---- Interval operations (same structure as inc/gpu_interval.hpp) -------------------- */
DEV ival i_neg(ival x) { return iv(-x.hi, -x.lo); }                                      /* :66-68 */
DEV ival i_add(ival x, ival y) { return iv(rd_add(x.lo, y.lo), ru_add(x.hi, y.hi)); }    /* :72-74 */
DEV ival i_add_f(ival x, float y) { return iv(rd_add(x.lo, y), ru_add(x.hi, y)); }       /* :76-78 */
/* Sign-case table of inc/gpu_interval.hpp:86-146, evaluated without branches (lanes of a
 * wave are 64 different tiles and would diverge over the nine cases).  With the classes
 *   M: lo < 0 && hi > 0   N: lo < 0 && !(hi > 0)   P: !(lo < 0) && hi > 0   Z: neither
 * the table's lower bound is RD(p*q), its upper bound RU(r*s) with
 *   p = y:P ? x.lo : y:N ? x.hi : (x:N ? x.lo : x.hi)      q = (x:N || (x:M && y:P)) ? y.hi : y.lo
 *   r = (y:P || (x:P && y:M)) ? x.hi : x.lo                s = (x:P || (x:M && y:P)) ? y.hi : y.lo
 * M*M additionally takes min / max with RD(x.lo*y.hi) and RU(x.hi*y.hi); any Z operand gives
 * [0, 0].  (Comparisons with NaN are false exactly as in the nested ifs.) */
DEV ival i_mul(ival x, ival y)                                                           /* :86-146 */
{
    const bool xn = x.lo < 0.0f, xp = x.hi > 0.0f, yn = y.lo < 0.0f, yp = y.hi > 0.0f;
    const bool xM = xn && xp, xN = xn && !xp, xP = !xn && xp;
    const bool yM = yn && yp, yN = yn && !yp, yP = !yn && yp;
    const float p = yP ? x.lo : (yN ? x.hi : (xN ? x.lo : x.hi));
    const float q = (xN || (xM && yP)) ? y.hi : y.lo;
    const float r = (yP || (xP && yM)) ? x.hi : x.lo;
    const float s = (xP || (xM && yP)) ? y.hi : y.lo;
    float lo = rd_mul(p, q);
    float hi = ru_mul(r, s);
    const float lo2 = rd_mul(x.lo, y.hi);
    const float hi2 = ru_mul(x.hi, y.hi);
    const bool mm = xM && yM;
    lo = mm ? mpr_fminf(lo2, lo) : lo;
    hi = mm ? mpr_fmaxf(hi, hi2) : hi;
    const bool zero = !(xn || xp) || !(yn || yp);
    return iv(zero ? 0.0f : lo, zero ? 0.0f : hi);
}
DEV ival i_mul_f(ival x, float y)                                                        /* :148-154 */
{
    if (y < 0.0f) return iv(rd_mul(x.hi, y), ru_mul(x.lo, y));
    return iv(rd_mul(x.lo, y), ru_mul(x.hi, y));
}
/* The six sign cases of inc/gpu_interval.hpp:162-190 pick the operands of the two directed
 * quotients; picking them with selects (same conditions, same nesting) keeps the 64 tiles of a
 * wave on one path instead of up to six:
 *        x.hi < 0          x.lo < 0 (else)      otherwise
 *   lo:  (yn?x.hi:x.lo)/y.lo   (yn?x.hi:x.lo)/(yn?y.hi:y.lo)   (yn?x.hi:x.lo)/y.hi        yn: y.hi < 0
 *   hi:  (yn?x.lo:x.hi)/y.hi   (yn?x.lo:x.hi)/(yn?y.hi:y.lo)   (yn?x.lo:x.hi)/y.lo                  */
DEV ival i_div(ival x, ival y)                                                           /* :162-190 */
{
    const float inf = mpr_u2f(0x7F800000u);
    const bool yz = y.lo <= 0.0f && y.hi >= 0.0f;
    const bool xn = x.hi < 0.0f, xm = !xn && x.lo < 0.0f, yn = y.hi < 0.0f;
    const float a1 = yn ? x.hi : x.lo, a2 = yn ? x.lo : x.hi;
    const float ym = yn ? y.hi : y.lo;
    const float b1 = xn ? y.lo : (xm ? ym : y.hi);
    const float b2 = xn ? y.hi : (xm ? ym : y.lo);
    const ival r = div_dir(a1, b1, a2, b2);
    return iv(yz ? -inf : r.lo, yz ? inf : r.hi);
}
DEV ival i_div_f(ival x, float y)                                                        /* :192-200 */
{
    const float inf = mpr_u2f(0x7F800000u);
    const bool neg = y < 0.0f, pos = y > 0.0f;
    const ival r = div_dir(neg ? x.hi : x.lo, y, neg ? x.lo : x.hi, y);
    return iv((neg || pos) ? r.lo : -inf, (neg || pos) ? r.hi : inf);
}
DEV ival i_fdiv(float x, ival y) { return i_div(iv(x, x), y); }                          /* :202-204 */
DEV ival i_min(ival x, ival y, int& choice)                                              /* :208-216 */
{
    const bool c1 = x.hi < y.lo;
    const bool c2 = !c1 && (y.hi < x.lo);
    choice = c1 ? 1 : (c2 ? 2 : choice);
    const float lo = mpr_fminf(x.lo, y.lo), hi = mpr_fminf(x.hi, y.hi);
    return iv(c1 ? x.lo : (c2 ? y.lo : lo), c1 ? x.hi : (c2 ? y.hi : hi));
}
DEV ival i_min_f(ival x, float y, int& choice) { return i_min(x, iv(y, y), choice); }    /* :218-228 */
DEV ival i_max(ival x, ival y, int& choice)                                              /* :232-240 */
{
    const bool c1 = x.lo > y.hi;
    const bool c2 = !c1 && (y.lo > x.hi);
    choice = c1 ? 1 : (c2 ? 2 : choice);
    const float lo = mpr_fmaxf(x.lo, y.lo), hi = mpr_fmaxf(x.hi, y.hi);
    return iv(c1 ? x.lo : (c2 ? y.lo : lo), c1 ? x.hi : (c2 ? y.hi : hi));
}
DEV ival i_max_f(ival x, float y, int& choice) { return i_max(x, iv(y, y), choice); }    /* :242-252 */
DEV ival i_square(ival x)                                                                /* :256-266 */
{
    const float a = ru_mul(x.lo, x.lo), b = ru_mul(x.hi, x.hi);
    const float c = rd_mul(x.lo, x.lo), d = rd_mul(x.hi, x.hi);
    const bool neg = x.hi < 0.0f, pos = x.lo > 0.0f, big = -x.lo > x.hi;
    return iv(neg ? d : (pos ? c : 0.0f), neg ? a : (pos ? b : (big ? a : b)));
}
DEV ival i_abs(ival x)                                                                   /* :268-276 */
{
    const bool nonneg = x.lo >= 0.0f, neg = x.hi < 0.0f;
    const float m = mpr_fmaxf(-x.lo, x.hi);
    return iv(nonneg ? x.lo : (neg ? -x.hi : 0.0f), nonneg ? x.hi : (neg ? -x.lo : m));
}
DEV ival i_sub(ival x, ival y) { return iv(rd_sub(x.lo, y.hi), ru_sub(x.hi, y.lo)); }    /* :284-286 */
DEV ival i_sub_f(ival x, float y) { return iv(rd_sub(x.lo, y), ru_sub(x.hi, y)); }       /* :288-290 */
DEV ival i_fsub(float x, ival y) { return iv(rd_sub(x, y.hi), ru_sub(x, y.lo)); }        /* :292-294 */
DEV ival i_sqrt(ival x)                                                                  /* :296-304 */
{
    const float nan = mpr_u2f(0x7FC00000u);
    /* x.hi < 0: NaN; x.lo <= 0: [0, RU(sqrt(x.hi))] (RD(sqrt(+0)) is +0); else both bounds */
    const ival r = sqrt_dir((x.lo <= 0.0f) ? 0.0f : x.lo, x.hi);
    const bool bad = x.hi < 0.0f;
    return iv(bad ? nan : r.lo, bad ? nan : r.hi);
}
DEV ival i_acos(ival x)                                                                  /* :306-314 */
{
    const float nan = mpr_u2f(0x7FC00000u);
    if (x.hi < -1.0f || x.lo > 1.0f) return iv(nan, nan);
    return iv(d2f_rd(::acos((double)x.hi)), d2f_ru(::acos((double)x.lo)));
}
DEV ival i_asin(ival x)                                                                  /* :316-324 */
{
    const float nan = mpr_u2f(0x7FC00000u);
    if (x.hi < -1.0f || x.lo > 1.0f) return iv(nan, nan);
    return iv(d2f_rd(::asin((double)x.lo)), d2f_ru(::asin((double)x.hi)));
}
DEV ival i_atan(ival x) { return iv(d2f_rd(::atan((double)x.lo)), d2f_ru(::atan((double)x.hi))); }   /* :326-330 */
DEV ival i_exp(ival x) { return iv(d2f_rd(::exp((double)x.lo)), d2f_ru(::exp((double)x.hi))); }      /* :332-336 */
DEV ival i_cos(ival) { return iv(-1.0f, 1.0f); }                                         /* :346-353 */
DEV ival i_sin(ival) { return iv(-1.0f, 1.0f); }                                         /* :378-380 */
DEV ival i_log(ival x)                                                                   /* :382-390 */
{
    const float nan = mpr_u2f(0x7FC00000u);
    if (x.hi < 0.0f) return iv(nan, nan);
    else if (x.lo <= 0.0f) return iv(0.0f, d2f_ru(::log((double)x.hi)));
    else return iv(d2f_rd(::log((double)x.lo)), d2f_ru(::log((double)x.hi)));
}

/* one interval clause (op wave-uniform); src/context.cu:236-279 */
DEV ival interval_clause(uint32_t op, ival lhs, ival rhs, float imm, int& choice)
{
    switch (op) {
        case MPR_OP_SQUARE_LHS: return i_square(lhs);
        case MPR_OP_SQRT_LHS: return i_sqrt(lhs);
        case MPR_OP_NEG_LHS: return i_neg(lhs);
        case MPR_OP_SIN_LHS: return i_sin(lhs);
        case MPR_OP_COS_LHS: return i_cos(lhs);
        case MPR_OP_ASIN_LHS: return i_asin(lhs);
        case MPR_OP_ACOS_LHS: return i_acos(lhs);
        case MPR_OP_ATAN_LHS: return i_atan(lhs);
        case MPR_OP_EXP_LHS: return i_exp(lhs);
        case MPR_OP_ABS_LHS: return i_abs(lhs);
        case MPR_OP_LOG_LHS: return i_log(lhs);
        case MPR_OP_ADD_LHS_IMM: return i_add_f(lhs, imm);
        case MPR_OP_ADD_LHS_RHS: return i_add(lhs, rhs);
        case MPR_OP_MUL_LHS_IMM: return i_mul_f(lhs, imm);
        case MPR_OP_MUL_LHS_RHS: return i_mul(lhs, rhs);
        case MPR_OP_MIN_LHS_IMM: return i_min_f(lhs, imm, choice);
        case MPR_OP_MIN_LHS_RHS: return i_min(lhs, rhs, choice);
        case MPR_OP_MAX_LHS_IMM: return i_max_f(lhs, imm, choice);
        case MPR_OP_MAX_LHS_RHS: return i_max(lhs, rhs, choice);
        case MPR_OP_SUB_LHS_IMM: return i_sub_f(lhs, imm);
        case MPR_OP_SUB_IMM_RHS: return i_fsub(imm, rhs);
        case MPR_OP_SUB_LHS_RHS: return i_sub(lhs, rhs);
        case MPR_OP_DIV_LHS_IMM: return i_div_f(lhs, imm);
        case MPR_OP_DIV_IMM_RHS: return i_fdiv(imm, rhs);
        case MPR_OP_DIV_LHS_RHS: return i_div(lhs, rhs);
        case MPR_OP_COPY_IMM: return iv(imm, imm);
        case MPR_OP_COPY_LHS: return lhs;
        case MPR_OP_COPY_RHS: return rhs;
        default: { const float nan = mpr_u2f(0x7FC00000u); return iv(nan, nan); }
    }
}

/* ---- float clause; round-to-nearest.  src/context.cu:882-921 ------------------------- */
DEV float float_clause(uint32_t op, float lhs, float rhs, float imm)
{
    switch (op) {
        case MPR_OP_SQUARE_LHS: return lhs * lhs;
        case MPR_OP_SQRT_LHS: return __builtin_sqrtf(lhs);
        case MPR_OP_NEG_LHS: return -lhs;
        case MPR_OP_SIN_LHS: return mpr_sinf(lhs);
        case MPR_OP_COS_LHS: return mpr_cosf(lhs);
        case MPR_OP_ASIN_LHS: return mpr_asinf(lhs);
        case MPR_OP_ACOS_LHS: return mpr_acosf(lhs);
        case MPR_OP_ATAN_LHS: return mpr_atanf(lhs);
        case MPR_OP_EXP_LHS: return mpr_expf(lhs);
        case MPR_OP_ABS_LHS: return __builtin_fabsf(lhs);
        case MPR_OP_LOG_LHS: return mpr_logf(lhs);
        case MPR_OP_ADD_LHS_IMM: return lhs + imm;
        case MPR_OP_ADD_LHS_RHS: return lhs + rhs;
        case MPR_OP_MUL_LHS_IMM: return lhs * imm;
        case MPR_OP_MUL_LHS_RHS: return lhs * rhs;
        case MPR_OP_MIN_LHS_IMM: return mpr_fminf(lhs, imm);
        case MPR_OP_MIN_LHS_RHS: return mpr_fminf(lhs, rhs);
        case MPR_OP_MAX_LHS_IMM: return mpr_fmaxf(lhs, imm);
        case MPR_OP_MAX_LHS_RHS: return mpr_fmaxf(lhs, rhs);
        case MPR_OP_SUB_LHS_IMM: return lhs - imm;
        case MPR_OP_SUB_IMM_RHS: return imm - rhs;
        case MPR_OP_SUB_LHS_RHS: return lhs - rhs;
        case MPR_OP_DIV_LHS_IMM: return lhs / imm;
        case MPR_OP_DIV_IMM_RHS: return imm / rhs;
        case MPR_OP_DIV_LHS_RHS: return lhs / rhs;
        case MPR_OP_COPY_IMM: return imm;
        case MPR_OP_COPY_LHS: return lhs;
        case MPR_OP_COPY_RHS: return rhs;
        default: return mpr_u2f(0x7FC00000u);
    }
}

/* ---- Deriv; inc/gpu_deriv.hpp ------------------------------------------------------- */
struct deriv { float dx, dy, dz, v; };
DEV deriv dv(float v, float dx, float dy, float dz) { deriv r; r.dx = dx; r.dy = dy; r.dz = dz; r.v = v; return r; }
DEV deriv d_const(float f) { return dv(f, 0.0f, 0.0f, 0.0f); }
DEV deriv d_neg(deriv a) { return dv(-a.v, -a.dx, -a.dy, -a.dz); }
DEV deriv d_add(deriv a, deriv b) { return dv(a.v + b.v, a.dx + b.dx, a.dy + b.dy, a.dz + b.dz); }
DEV deriv d_add_f(deriv a, float b) { return dv(a.v + b, a.dx, a.dy, a.dz); }
DEV deriv d_mul(deriv a, deriv b)
{
    return dv(a.v * b.v, a.dx * b.v + b.dx * a.v, a.dy * b.v + b.dy * a.v, a.dz * b.v + b.dz * a.v);
}
DEV deriv d_mul_f(deriv a, float b) { return dv(a.v * b, a.dx * b, a.dy * b, a.dz * b); }
DEV deriv d_div(deriv a, deriv b)
{
    const float d = b.v * b.v;
    return dv(a.v / b.v, (b.v * a.dx - a.v * b.dx) / d, (b.v * a.dy - a.v * b.dy) / d,
              (b.v * a.dz - a.v * b.dz) / d);
}
DEV deriv d_div_f(deriv a, float b) { return dv(a.v / b, a.dx / b, a.dy / b, a.dz / b); }
DEV deriv d_fdiv(float a, deriv b)
{
    const float d = b.v * b.v;
    return dv(a / b.v, -a * b.dx / d, -a * b.dy / d, -a * b.dz / d);
}
DEV deriv d_min(deriv a, deriv b) { return (a.v < b.v) ? a : b; }
DEV deriv d_min_f(deriv a, float b) { return (a.v < b) ? a : d_const(b); }
DEV deriv d_max(deriv a, deriv b) { return (a.v >= b.v) ? a : b; }
DEV deriv d_max_f(deriv a, float b) { return (a.v >= b) ? a : d_const(b); }
DEV deriv d_abs(deriv a) { return (a.v < 0.0f) ? d_neg(a) : a; }
DEV deriv d_sub(deriv a, deriv b) { return dv(a.v - b.v, a.dx - b.dx, a.dy - b.dy, a.dz - b.dz); }
DEV deriv d_sub_f(deriv a, float b) { return dv(a.v - b, a.dx, a.dy, a.dz); }
DEV deriv d_fsub(float a, deriv b) { return dv(a - b.v, -b.dx, -b.dy, -b.dz); }
DEV deriv d_sqrt(deriv a)
{
    const float s = __builtin_sqrtf(a.v);
    const float d = 2 * s;
    return dv(s, a.dx / d, a.dy / d, a.dz / d);
}
DEV deriv d_atan(deriv a)
{
    const float d = a.v * a.v + 1;
    return dv(mpr_atanf(a.v), a.dx / d, a.dy / d, a.dz / d);
}
DEV deriv d_acos(deriv a)
{
    const float d = -__builtin_sqrtf(1 - a.v * a.v);
    return dv(mpr_acosf(a.v), a.dx / d, a.dy / d, a.dz / d);
}
DEV deriv d_asin(deriv a)
{
    const float d = __builtin_sqrtf(1 - a.v * a.v);
    return dv(mpr_asinf(a.v), a.dx / d, a.dy / d, a.dz / d);
}
DEV deriv d_exp(deriv a)
{
    const float v = mpr_expf(a.v);
    return dv(v, v * a.dx, v * a.dy, v * a.dz);
}
DEV deriv d_cos(deriv a)
{
    const float s = -mpr_sinf(a.v);
    return dv(mpr_cosf(a.v), s * a.dx, s * a.dy, s * a.dz);
}
DEV deriv d_sin(deriv a)
{
    const float c = mpr_cosf(a.v);
    return dv(mpr_sinf(a.v), c * a.dx, c * a.dy, c * a.dz);
}
DEV deriv d_log(deriv a)
{
    const float v = a.v;
    return dv(mpr_logf(v), a.dx / v, a.dy / v, a.dz / v);
}
/* src/context.cu:1081-1118 (SQUARE is lhs * lhs, :1081) */
DEV deriv deriv_clause(uint32_t op, deriv lhs, deriv rhs, float imm)
{
    switch (op) {
        case MPR_OP_SQUARE_LHS: return d_mul(lhs, lhs);
        case MPR_OP_SQRT_LHS: return d_sqrt(lhs);
        case MPR_OP_NEG_LHS: return d_neg(lhs);
        case MPR_OP_SIN_LHS: return d_sin(lhs);
        case MPR_OP_COS_LHS: return d_cos(lhs);
        case MPR_OP_ASIN_LHS: return d_asin(lhs);
        case MPR_OP_ACOS_LHS: return d_acos(lhs);
        case MPR_OP_ATAN_LHS: return d_atan(lhs);
        case MPR_OP_EXP_LHS: return d_exp(lhs);
        case MPR_OP_ABS_LHS: return d_abs(lhs);
        case MPR_OP_LOG_LHS: return d_log(lhs);
        case MPR_OP_ADD_LHS_IMM: return d_add_f(lhs, imm);
        case MPR_OP_ADD_LHS_RHS: return d_add(lhs, rhs);
        case MPR_OP_MUL_LHS_IMM: return d_mul_f(lhs, imm);
        case MPR_OP_MUL_LHS_RHS: return d_mul(lhs, rhs);
        case MPR_OP_MIN_LHS_IMM: return d_min_f(lhs, imm);
        case MPR_OP_MIN_LHS_RHS: return d_min(lhs, rhs);
        case MPR_OP_MAX_LHS_IMM: return d_max_f(lhs, imm);
        case MPR_OP_MAX_LHS_RHS: return d_max(lhs, rhs);
        case MPR_OP_SUB_LHS_IMM: return d_sub_f(lhs, imm);
        case MPR_OP_SUB_IMM_RHS: return d_fsub(imm, rhs);
        case MPR_OP_SUB_LHS_RHS: return d_sub(lhs, rhs);
        case MPR_OP_DIV_LHS_IMM: return d_div_f(lhs, imm);
        case MPR_OP_DIV_IMM_RHS: return d_fdiv(imm, rhs);
        case MPR_OP_DIV_LHS_RHS: return d_div(lhs, rhs);
        case MPR_OP_COPY_IMM: return d_const(imm);
        case MPR_OP_COPY_LHS: return lhs;
        case MPR_OP_COPY_RHS: return rhs;
        default: { const float nan = mpr_u2f(0x7FC00000u); return dv(nan, nan, nan, nan); }
    }
}

DEV uint32_t f2u8(float v)
{   /* truncate toward zero, saturate, NaN -> 0 */
    if (!(v == v)) return 0;
    if (v <= 0.0f) return 0;
    if (v >= 255.0f) return 255;
    return (uint32_t)(int)v;
}

}  // namespace mprk
