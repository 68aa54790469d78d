/*
 * mpr_oracle.c — CPU restatement of mpr's hierarchical tape-evaluation renderer.
 *
 * *** TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
 * *** cpu_baseline leg may load this library.  The product (libmpr_amd.so) never does.
 *
 * What it restates (semantics per tile / voxel / pixel, sequential loops; NOT the CUDA
 * scheduling).  All file:line citations are into the reference tree (mkeeter/mpr):
 *   inc/gpu_interval.hpp:18-391   Interval and its outward-rounded operations
 *   inc/gpu_deriv.hpp:18-204      Deriv (forward-mode AD)
 *   src/context.cu:23-1132        kernel semantics (preload, calculate_intervals, eval_tiles_i
 *                                 incl. tape pushing, mask, assign_next, subdivide, copy_filled,
 *                                 calculate_voxels/pixels, eval_voxels_f, eval_pixels_d)
 *   src/context.cu:1136-1508      stage order of render2D / render3D / render2D_brute
 *                                 (+ a brute force in 3-D, which the reference does not have: every 4^3 tile to the float
 *                                 pass with the root tape — tests/test_oracle_kat.py holds the hierarchy against it)
 *   src/context.cpp:17-49         buffer sizes
 *
 * Pinning status: the reference ships no tests, golden images or stored vectors for this
 * path, its CUDA sources cannot be compiled here (no nvcc, libfive and Eigen
 * absent; oracle/_ref is therefore not buildable), so there is no reference OUTPUT to pin
 * against: **parity unpinned** in that sense.  What pins the oracle instead are the known
 * answers derivable from the reference's own source text (tests/test_oracle_kat.py):
 * the compiled two-sphere kernel of benchmark/brute.cu:39-61, the hierarchy == brute-force
 * invariant that benchmark/brute.cu relies on, the tile-occupancy semantics of
 * benchmark/circle.cpp:42-103 and the clause table of benchmark/print_tape_table.cpp:29-51 —
 * and two routes that share none of this file's code: an independent numpy evaluator's frames of
 * the benchmark models and of random shapes (tests/golden/make_independent.py, tests/test_independent.py)
 * and exact rational arithmetic for every interval opcode (tests/test_soundness.py).
 *
 * Deliberate, documented choices where the CUDA toolchain's behaviour cannot be known:
 *   - no FMA contraction anywhere (nvcc may contract; irrelevant for the benchmark views,
 *     SURVEY.md §8(c));
 *   - float transcendentals come from oracle/orc_fmath.h — the oracle's own implementation of the
 *     algorithm the product defines them by (nothing is shared with include/mpr_fmath.h) — instead
 *     of CUDA libm; double transcendentals inside Interval ops come from glibc;
 *   - powf(x, 2) is x*x (inc/gpu_deriv.hpp:86,100; src/context.cu:1125-1127);
 *   - float -> uint8 conversion of normals saturates and maps NaN to 0 (CUDA cvt.rzi.u8).
 *
 * Directed rounding: every interval operation runs with the FPU in round-up mode and gets
 * its lower bounds through negation, RD(a op b) = -RU((-a) op' b); this is bit-identical to
 * switching to round-down (IEEE symmetry) and orc_selftest_rounding() checks that claim
 * against fesetround(FE_DOWNWARD).  The kernels use the same formulation.
 *
 * Build: gcc -O2 -std=gnu11 -frounding-math -ffp-contract=off -fno-fast-math -mfma -fopenmp
 */
#define _GNU_SOURCE
#include "mpr_oracle.h"

#include <fenv.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "orc_fmath.h"

/* ===================================================================================== */
/* directed-rounding primitives; FE_UPWARD must be in effect                             */
/* ===================================================================================== */
static inline float ru_add(float a, float b) { volatile float x = a, y = b; volatile float r = x + y; return r; }
static inline float ru_mul(float a, float b) { volatile float x = a, y = b; volatile float r = x * y; return r; }
static inline float ru_div(float a, float b) { volatile float x = a, y = b; volatile float r = x / y; return r; }
static inline float rd_add(float a, float b) { return -ru_add(-a, -b); }
static inline float ru_sub(float a, float b) { return ru_add(a, -b); }
static inline float rd_sub(float a, float b) { return -ru_add(-a, b); }
static inline float rd_mul(float a, float b) { return -ru_mul(-a, b); }
static inline float rd_div(float a, float b) { return -ru_div(-a, b); }
static inline float ru_sqrt(float a) { volatile float x = a; volatile float r = sqrtf(x); return r; }
static inline float rd_sqrt(float a)
{
    fesetround(FE_DOWNWARD);
    volatile float x = a;
    volatile float r = sqrtf(x);
    fesetround(FE_UPWARD);
    return r;
}
static inline float d2f_ru(double d) { volatile double x = d; volatile float r = (float)x; return r; }
static inline float d2f_rd(double d) { volatile double x = -d; volatile float r = (float)x; return -r; }
/* double libm evaluated in round-to-nearest (what ::acos etc. do on the device) */
#define RN_DOUBLE(expr) ({ fesetround(FE_TONEAREST); volatile double _r = (expr); fesetround(FE_UPWARD); _r; })

typedef struct { float lo, hi; } ival;
static inline ival iv(float lo, float hi) { ival r = {lo, hi}; return r; }

/* inc/gpu_interval.hpp:66-68 */
static inline ival i_neg(ival x) { return iv(-x.hi, -x.lo); }
/* :72-82 */
static inline ival i_add(ival x, ival y) { return iv(rd_add(x.lo, y.lo), ru_add(x.hi, y.hi)); }
static inline ival i_add_f(ival x, float y) { return iv(rd_add(x.lo, y), ru_add(x.hi, y)); }
/* :86-146 — sign-case table */
static ival i_mul(ival x, ival y)
{
    if (x.lo < 0.0f) {
        if (x.hi > 0.0f) {
            if (y.lo < 0.0f) {
                if (y.hi > 0.0f) { /* M * M */
                    return iv(orc_fminf(rd_mul(x.lo, y.hi), rd_mul(x.hi, y.lo)),
                              orc_fmaxf(ru_mul(x.lo, y.lo), ru_mul(x.hi, y.hi)));
                } else { /* M * N */
                    return iv(rd_mul(x.hi, y.lo), ru_mul(x.lo, y.lo));
                }
            } else {
                if (y.hi > 0.0f) { /* M * P */
                    return iv(rd_mul(x.lo, y.hi), ru_mul(x.hi, y.hi));
                } else { /* M * Z */
                    return iv(0.0f, 0.0f);
                }
            }
        } else {
            if (y.lo < 0.0f) {
                if (y.hi > 0.0f) { /* N * M */
                    return iv(rd_mul(x.lo, y.hi), ru_mul(x.lo, y.lo));
                } else { /* N * N */
                    return iv(rd_mul(x.hi, y.hi), ru_mul(x.lo, y.lo));
                }
            } else {
                if (y.hi > 0.0f) { /* N * P */
                    return iv(rd_mul(x.lo, y.hi), ru_mul(x.hi, y.lo));
                } else { /* N * Z */
                    return iv(0.0f, 0.0f);
                }
            }
        }
    } else {
        if (x.hi > 0.0f) {
            if (y.lo < 0.0f) {
                if (y.hi > 0.0f) { /* P * M */
                    return iv(rd_mul(x.hi, y.lo), ru_mul(x.hi, y.hi));
                } else { /* P * N */
                    return iv(rd_mul(x.hi, y.lo), ru_mul(x.lo, y.hi));
                }
            } else {
                if (y.hi > 0.0f) { /* P * P */
                    return iv(rd_mul(x.lo, y.lo), ru_mul(x.hi, y.hi));
                } else { /* P * Z */
                    return iv(0.0f, 0.0f);
                }
            }
        } else { /* Z * ? */
            return iv(0.0f, 0.0f);
        }
    }
}
/* :148-154 */
static inline ival i_mul_f(ival x, float y)
{
    if (y < 0.0f) return iv(rd_mul(x.hi, y), ru_mul(x.lo, y));
    return iv(rd_mul(x.lo, y), ru_mul(x.hi, y));
}
/* :162-190 */
static ival i_div(ival x, ival y)
{
    if (y.lo <= 0.0f && y.hi >= 0.0f) {
        return iv(-INFINITY, INFINITY);
    } else if (x.hi < 0.0f) {
        if (y.hi < 0.0f) return iv(rd_div(x.hi, y.lo), ru_div(x.lo, y.hi));
        else return iv(rd_div(x.lo, y.lo), ru_div(x.hi, y.hi));
    } else if (x.lo < 0.0f) {
        if (y.hi < 0.0f) return iv(rd_div(x.hi, y.hi), ru_div(x.lo, y.hi));
        else return iv(rd_div(x.lo, y.lo), ru_div(x.hi, y.lo));
    } else {
        if (y.hi < 0.0f) return iv(rd_div(x.hi, y.hi), ru_div(x.lo, y.lo));
        else return iv(rd_div(x.lo, y.hi), ru_div(x.hi, y.lo));
    }
}
/* :192-200 */
static inline ival i_div_f(ival x, float y)
{
    if (y < 0.0f) return iv(rd_div(x.hi, y), ru_div(x.lo, y));
    else if (y > 0.0f) return iv(rd_div(x.lo, y), ru_div(x.hi, y));
    else return iv(-INFINITY, INFINITY);
}
/* :202-204 */
static inline ival i_fdiv(float x, ival y) { return i_div(iv(x, x), y); }
/* :208-228 */
static inline ival i_min(ival x, ival y, int* choice)
{
    if (x.hi < y.lo) { *choice = 1; return x; }
    else if (y.hi < x.lo) { *choice = 2; return y; }
    return iv(orc_fminf(x.lo, y.lo), orc_fminf(x.hi, y.hi));
}
static inline ival i_min_f(ival x, float y, int* choice)
{
    if (x.hi < y) { *choice = 1; return x; }
    else if (y < x.lo) { *choice = 2; return iv(y, y); }
    return iv(orc_fminf(x.lo, y), orc_fminf(x.hi, y));
}
/* :232-252 */
static inline ival i_max(ival x, ival y, int* choice)
{
    if (x.lo > y.hi) { *choice = 1; return x; }
    else if (y.lo > x.hi) { *choice = 2; return y; }
    return iv(orc_fmaxf(x.lo, y.lo), orc_fmaxf(x.hi, y.hi));
}
static inline ival i_max_f(ival x, float y, int* choice)
{
    if (x.lo > y) { *choice = 1; return x; }
    else if (y > x.hi) { *choice = 2; return iv(y, y); }
    return iv(orc_fmaxf(x.lo, y), orc_fmaxf(x.hi, y));
}
/* :256-266 */
static inline ival i_square(ival x)
{
    if (x.hi < 0.0f) return iv(rd_mul(x.hi, x.hi), ru_mul(x.lo, x.lo));
    else if (x.lo > 0.0f) return iv(rd_mul(x.lo, x.lo), ru_mul(x.hi, x.hi));
    else if (-x.lo > x.hi) return iv(0.0f, ru_mul(x.lo, x.lo));
    else return iv(0.0f, ru_mul(x.hi, x.hi));
}
/* :268-276 */
static inline ival i_abs(ival x)
{
    if (x.lo >= 0.0f) return x;
    else if (x.hi < 0.0f) return i_neg(x);
    else return iv(0.0f, orc_fmaxf(-x.lo, x.hi));
}
/* :284-294 */
static inline ival i_sub(ival x, ival y) { return iv(rd_sub(x.lo, y.hi), ru_sub(x.hi, y.lo)); }
static inline ival i_sub_f(ival x, float y) { return iv(rd_sub(x.lo, y), ru_sub(x.hi, y)); }
static inline ival i_fsub(float x, ival y) { return iv(rd_sub(x, y.hi), ru_sub(x, y.lo)); }
/* :296-304 */
static inline ival i_sqrt(ival x)
{
    if (x.hi < 0.0f) return iv(NAN, NAN);
    else if (x.lo <= 0.0f) return iv(0.0f, ru_sqrt(x.hi));
    else return iv(rd_sqrt(x.lo), ru_sqrt(x.hi));
}
/* :306-314 */
static inline ival i_acos(ival x)
{
    if (x.hi < -1.0f || x.lo > 1.0f) return iv(NAN, NAN);
    const double a = RN_DOUBLE(acos((double)x.hi)), b = RN_DOUBLE(acos((double)x.lo));
    return iv(d2f_rd(a), d2f_ru(b));
}
/* :316-324 */
static inline ival i_asin(ival x)
{
    if (x.hi < -1.0f || x.lo > 1.0f) return iv(NAN, NAN);
    const double a = RN_DOUBLE(asin((double)x.lo)), b = RN_DOUBLE(asin((double)x.hi));
    return iv(d2f_rd(a), d2f_ru(b));
}
/* :326-330 */
static inline ival i_atan(ival x)
{
    const double a = RN_DOUBLE(atan((double)x.lo)), b = RN_DOUBLE(atan((double)x.hi));
    return iv(d2f_rd(a), d2f_ru(b));
}
/* :332-336 */
static inline ival i_exp(ival x)
{
    const double a = RN_DOUBLE(exp((double)x.lo)), b = RN_DOUBLE(exp((double)x.hi));
    return iv(d2f_rd(a), d2f_ru(b));
}
/* :346-353 — cos() returns [-1, 1] unconditionally (everything after :353 is dead code) */
static inline ival i_cos(ival x) { (void)x; return iv(-1.0f, 1.0f); }
/* :378-380 — sin(x) = cos(x - pi/2), hence also [-1, 1] */
static inline ival i_sin(ival x) { (void)x; return iv(-1.0f, 1.0f); }
/* :382-390 — NB the lower bound 0 (not -inf) when the argument touches 0 */
static inline ival i_log(ival x)
{
    if (x.hi < 0.0f) return iv(NAN, NAN);
    else if (x.lo <= 0.0f) {
        const double b = RN_DOUBLE(log((double)x.hi));
        return iv(0.0f, d2f_ru(b));
    } else {
        const double a = RN_DOUBLE(log((double)x.lo)), b = RN_DOUBLE(log((double)x.hi));
        return iv(d2f_rd(a), d2f_ru(b));
    }
}

/* one interval clause; FE_UPWARD in effect.  src/context.cu:236-279 */
static inline ival interval_clause(uint32_t op, ival lhs, ival rhs, float imm, int* choice)
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
        default: return iv(NAN, NAN);   /* src/context.cu:281 asserts */
    }
}

/* one float clause; round-to-nearest.  src/context.cu:882-921 */
static inline float float_clause(uint32_t op, float lhs, float rhs, float imm)
{
    switch (op) {
        case MPR_OP_SQUARE_LHS: return lhs * lhs;
        case MPR_OP_SQRT_LHS: return sqrtf(lhs);
        case MPR_OP_NEG_LHS: return -lhs;
        case MPR_OP_SIN_LHS: return orc_sinf(lhs);
        case MPR_OP_COS_LHS: return orc_cosf(lhs);
        case MPR_OP_ASIN_LHS: return orc_asinf(lhs);
        case MPR_OP_ACOS_LHS: return orc_acosf(lhs);
        case MPR_OP_ATAN_LHS: return orc_atanf(lhs);
        case MPR_OP_EXP_LHS: return orc_expf(lhs);
        case MPR_OP_ABS_LHS: return fabsf(lhs);
        case MPR_OP_LOG_LHS: return orc_logf(lhs);
        case MPR_OP_ADD_LHS_IMM: return lhs + imm;
        case MPR_OP_ADD_LHS_RHS: return lhs + rhs;
        case MPR_OP_MUL_LHS_IMM: return lhs * imm;
        case MPR_OP_MUL_LHS_RHS: return lhs * rhs;
        case MPR_OP_MIN_LHS_IMM: return orc_fminf(lhs, imm);
        case MPR_OP_MIN_LHS_RHS: return orc_fminf(lhs, rhs);
        case MPR_OP_MAX_LHS_IMM: return orc_fmaxf(lhs, imm);
        case MPR_OP_MAX_LHS_RHS: return orc_fmaxf(lhs, rhs);
        case MPR_OP_SUB_LHS_IMM: return lhs - imm;
        case MPR_OP_SUB_IMM_RHS: return imm - rhs;
        case MPR_OP_SUB_LHS_RHS: return lhs - rhs;
        case MPR_OP_DIV_LHS_IMM: return lhs / imm;
        case MPR_OP_DIV_IMM_RHS: return imm / rhs;
        case MPR_OP_DIV_LHS_RHS: return lhs / rhs;
        case MPR_OP_COPY_IMM: return imm;
        case MPR_OP_COPY_LHS: return lhs;
        case MPR_OP_COPY_RHS: return rhs;
        default: return NAN;
    }
}

/* ===================================================================================== */
/* Deriv — inc/gpu_deriv.hpp:18-204.  (dx, dy, dz, v), round-to-nearest                  */
/* ===================================================================================== */
typedef struct { float dx, dy, dz, v; } deriv;
static inline deriv dv(float v, float dx, float dy, float dz) { deriv r = {dx, dy, dz, v}; return r; }
static inline deriv d_const(float f) { return dv(f, 0.0f, 0.0f, 0.0f); }                     /* :20 */
static inline deriv d_neg(deriv a) { return dv(-a.v, -a.dx, -a.dy, -a.dz); }                 /* :44-46 */
static inline deriv d_add(deriv a, deriv b) { return dv(a.v + b.v, a.dx + b.dx, a.dy + b.dy, a.dz + b.dz); } /* :50-55 */
static inline deriv d_add_f(deriv a, float b) { return dv(a.v + b, a.dx, a.dy, a.dz); }      /* :57-59 */
static inline deriv d_mul(deriv a, deriv b)                                                  /* :67-72 */
{
    return dv(a.v * b.v, a.dx * b.v + b.dx * a.v, a.dy * b.v + b.dy * a.v, a.dz * b.v + b.dz * a.v);
}
static inline deriv d_mul_f(deriv a, float b) { return dv(a.v * b, a.dx * b, a.dy * b, a.dz * b); } /* :74-79 */
static inline deriv d_div(deriv a, deriv b)                                                  /* :87-93 */
{
    const float d = b.v * b.v;
    return dv(a.v / b.v, (b.v * a.dx - a.v * b.dx) / d, (b.v * a.dy - a.v * b.dy) / d,
              (b.v * a.dz - a.v * b.dz) / d);
}
static inline deriv d_div_f(deriv a, float b) { return dv(a.v / b, a.dx / b, a.dy / b, a.dz / b); } /* :95-97 */
static inline deriv d_fdiv(float a, deriv b)                                                 /* :99-105 */
{
    const float d = b.v * b.v;
    return dv(a / b.v, -a * b.dx / d, -a * b.dy / d, -a * b.dz / d);
}
static inline deriv d_min(deriv a, deriv b) { return (a.v < b.v) ? a : b; }                  /* :109-111 */
static inline deriv d_min_f(deriv a, float b) { return (a.v < b) ? a : d_const(b); }         /* :113-115 */
static inline deriv d_max(deriv a, deriv b) { return (a.v >= b.v) ? a : b; }                 /* :123-125 */
static inline deriv d_max_f(deriv a, float b) { return (a.v >= b) ? a : d_const(b); }        /* :127-129 */
static inline deriv d_abs(deriv a) { return (a.v < 0.0f) ? d_neg(a) : a; }                   /* :146-152 */
static inline deriv d_sub(deriv a, deriv b) { return dv(a.v - b.v, a.dx - b.dx, a.dy - b.dy, a.dz - b.dz); } /* :156-161 */
static inline deriv d_sub_f(deriv a, float b) { return dv(a.v - b, a.dx, a.dy, a.dz); }      /* :163-165 */
static inline deriv d_fsub(float a, deriv b) { return dv(a - b.v, -b.dx, -b.dy, -b.dz); }    /* :167-169 */
static inline deriv d_sqrt(deriv a)                                                          /* :171-174 */
{
    const float d = 2 * sqrtf(a.v);
    return dv(sqrtf(a.v), a.dx / d, a.dy / d, a.dz / d);
}
static inline deriv d_atan(deriv a)                                                          /* :176-179 */
{
    const float d = a.v * a.v + 1;
    return dv(orc_atanf(a.v), a.dx / d, a.dy / d, a.dz / d);
}
static inline deriv d_acos(deriv a)                                                          /* :181-184 */
{
    const float d = -sqrtf(1 - a.v * a.v);
    return dv(orc_acosf(a.v), a.dx / d, a.dy / d, a.dz / d);
}
static inline deriv d_asin(deriv a)                                                          /* :186-189 */
{
    const float d = sqrtf(1 - a.v * a.v);
    return dv(orc_asinf(a.v), a.dx / d, a.dy / d, a.dz / d);
}
static inline deriv d_exp(deriv a)                                                           /* :191-194 */
{
    const float v = orc_expf(a.v);
    return dv(v, v * a.dx, v * a.dy, v * a.dz);
}
static inline deriv d_cos(deriv a)                                                           /* :196-199 */
{
    const float s = -orc_sinf(a.v);
    return dv(orc_cosf(a.v), s * a.dx, s * a.dy, s * a.dz);
}
static inline deriv d_sin(deriv a)                                                           /* :201-204 (upstream :196-199) */
{
    const float c = orc_cosf(a.v);
    return dv(orc_sinf(a.v), c * a.dx, c * a.dy, c * a.dz);
}
static inline deriv d_log(deriv a)                                                           /* :201-204 */
{
    const float v = a.v;
    return dv(orc_logf(v), a.dx / v, a.dy / v, a.dz / v);
}

/* one Deriv clause.  src/context.cu:1081-1118; NB SQUARE is evaluated as lhs * lhs (:1081) */
static inline deriv deriv_clause(uint32_t op, deriv lhs, deriv rhs, float imm)
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
        default: return dv(NAN, NAN, NAN, NAN);
    }
}

/* ===================================================================================== */
/* frame state                                                                           */
/* ===================================================================================== */
struct orc_frame {
    int32_t S, dim;
    int32_t* filled[4];
    size_t filled_n[4];
    uint32_t* normals;
    mpr_tile_node* tiles[4];
    size_t tiles_n[4];
    uint64_t* pool;
    int64_t pool_cap;
    int32_t tape_index;
    orc_counters c;
    float* heat;          /* heatmap frames (flags bit2): S x S work per pixel, else NULL */
};

static inline float imm_of(uint64_t d) { return orc_float(mpr_cl_immbits(d)); }

/* src/context.cu:23-30 */
typedef struct { int32_t x, y, z, w; } int4_;
static inline int4_ unpack(int32_t pos, int32_t tps)
{
    int4_ r = {pos % tps, (pos / tps) % tps, (pos / tps) / tps, pos % (tps * tps)};
    return r;
}

static inline void atomic_max_i32(int32_t* p, int32_t v)
{
    int32_t old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}

/* Interval triple for a tile — src/context.cu:78-159.  Corner coordinates in
 * round-to-nearest, transform in interval arithmetic (FE_UPWARD set by the caller AFTER the
 * corners were computed, hence the two-phase interface). */
typedef struct { ival x, y, z; } ival3;
static void tile_corners(int32_t position, int32_t tps, int dim, ival out[3])
{
    const int4_ p = unpack(position, tps);
    const float t = (float)tps;
    out[0] = iv((p.x / t - 0.5f) * 2.0f, ((p.x + 1) / t - 0.5f) * 2.0f);
    out[1] = iv((p.y / t - 0.5f) * 2.0f, ((p.y + 1) / t - 0.5f) * 2.0f);
    if (dim == 3) out[2] = iv((p.z / t - 0.5f) * 2.0f, ((p.z + 1) / t - 0.5f) * 2.0f);
    else out[2] = iv(0.0f, 0.0f);
}
#define M4(r, c) mat[(r) + (c) * 4]
#define M3(r, c) mat[(r) + (c) * 3]
static ival3 tile_intervals(const ival c[3], int dim, const float* mat, float z)
{
    ival3 o;
    if (dim == 3) {
        ival r[4];
        for (int i = 0; i < 4; ++i) {
            r[i] = i_add_f(i_add(i_add(i_mul_f(c[0], M4(i, 0)), i_mul_f(c[1], M4(i, 1))),
                                 i_mul_f(c[2], M4(i, 2))), M4(i, 3));
        }
        o.x = i_div(r[0], r[3]);
        o.y = i_div(r[1], r[3]);
        o.z = i_div(r[2], r[3]);
    } else {
        ival r[3];
        for (int i = 0; i < 3; ++i) {
            r[i] = i_add_f(i_add(i_mul_f(c[0], M3(i, 0)), i_mul_f(c[1], M3(i, 1))), M3(i, 2));
        }
        o.x = i_div(r[0], r[2]);
        o.y = i_div(r[1], r[2]);
        o.z = iv(z, z);
    }
    return o;
}

/* ---- eval_tiles_i for ONE tile — src/context.cu:188-459 --------------------------------
 * returns: 0 empty, 1 masked, 2 filled, 3 ambiguous (no push), 4 ambiguous + pushed,
 *          5 ambiguous, push abandoned (pool full)
 * fwd/bwd: words fetched by the forward / backward walk.  */
enum { T_EMPTY = 0, T_MASKED = 1, T_FILLED = 2, T_AMBIG = 3, T_PUSHED = 4, T_OVERFLOW = 5 };

static int eval_tile_i(orc_frame* f, int dim, int32_t* image, int32_t tps, mpr_tile_node* tile,
                       const float* mat, float z, int64_t* fwd, int64_t* bwd, int64_t* written,
                       int64_t* nclauses, int* slots_exceeded, int* would_push)
{
    uint64_t* const tape_data = f->pool;
    const int64_t POOL = f->pool_cap;

    ival corners[3];
    tile_corners(tile->position, tps, dim, corners);

    fesetround(FE_UPWARD);
    const ival3 in = tile_intervals(corners, dim, mat, z);

    ival slots[256];
    const uint64_t head0 = tape_data[0];
    slots[mpr_cl_out(head0)] = in.x;      /* byte 1 */
    slots[mpr_cl_lhs(head0)] = in.y;      /* byte 2 */
    slots[mpr_cl_rhs(head0)] = in.z;      /* byte 3 */

    const uint64_t* data = &tape_data[tile->tape];
    uint32_t choices[256];
    memset(choices, 0, sizeof(choices));
    int choice_index = 0;
    int has_any_choice = 0;

    for (;;) {
        const uint64_t d = *++data;
        ++*fwd;
        const uint32_t op = mpr_cl_op(d);
        if (!op) break;
        if (op == MPR_OP_JUMP) { data += mpr_cl_jump(d); continue; }
        const uint32_t o = mpr_cl_out(d), l = mpr_cl_lhs(d), r = mpr_cl_rhs(d);
        if (o >= MPR_KERNEL_SLOTS || l >= MPR_KERNEL_SLOTS || r >= MPR_KERNEL_SLOTS) *slots_exceeded = 1;
        int c = 0;
        slots[o] = interval_clause(op, slots[l], slots[r], imm_of(d), &c);
        ++*nclauses;
        if (mpr_op_is_minmax(op)) {
            if (choice_index < MPR_MAX_CHOICES) {
                choices[choice_index / 16] |= ((uint32_t)c << ((choice_index % 16) * 2));
            }
            choice_index++;
            has_any_choice |= (c != 0);
        }
    }
    fesetround(FE_TONEAREST);

    const uint32_t i_out = mpr_cl_out(*data);
    const ival result = slots[i_out];

    /* Empty — :293-296 */
    if (result.lo > 0.0f) { tile->position = -1; return T_EMPTY; }
    /* Masked — :299-305 */
    if (dim == 3) {
        const int4_ pos = unpack(tile->position, tps);
        if (__atomic_load_n(&image[pos.w], __ATOMIC_RELAXED) > pos.z) {
            /* culled by a tile that filled this column during the stage: with other timing it would
             * have gone on to push a tape (heatmap bounds only) */
            *would_push = !(result.hi < 0.0f) && has_any_choice;
            tile->position = -1;
            return T_MASKED;
        }
    }
    /* Filled — :308-317 */
    if (result.hi < 0.0f) {
        const int4_ pos = unpack(tile->position, tps);
        tile->position = -1;
        if (dim == 3) atomic_max_i32(&image[pos.w], pos.z);
        else image[pos.w] = 1;
        return T_FILLED;
    }
    if (!has_any_choice) return T_AMBIG;   /* :319-321 */

    /* ---- tape pushing — :323-458 ---- */
    uint8_t active[256];
    memset(active, 0, sizeof(active));
    active[i_out] = 1;

    if (__atomic_load_n(&f->tape_index, __ATOMIC_RELAXED) >= POOL) return T_OVERFLOW;     /* :336-338 */
    int32_t out_index = __atomic_fetch_add(&f->tape_index, MPR_SUBTAPE_CHUNK, __ATOMIC_RELAXED);
    int32_t out_offset = MPR_SUBTAPE_CHUNK;
    if ((int64_t)out_index + out_offset >= POOL) return T_OVERFLOW;                        /* :345-347 */

    out_offset--;
    tape_data[out_index + out_offset] = *data;     /* end marker */
    ++*written;

    for (;;) {
        uint64_t d = *--data;
        ++*bwd;
        const uint32_t op = mpr_cl_op(d);
        if (!op) break;
        if (op == MPR_OP_JUMP) { data += mpr_cl_jump(d); continue; }

        const int has_choice = mpr_op_is_minmax(op);
        choice_index -= has_choice;

        const uint32_t o = mpr_cl_out(d);
        if (!active[o]) continue;

        const int choice = (has_choice && choice_index < MPR_MAX_CHOICES)
                               ? ((choices[choice_index / 16] >> ((choice_index % 16) * 2)) & 3)
                               : 0;

        --out_offset;
        if (out_offset == 0) {
            const int32_t prev_index = out_index;
            if (__atomic_load_n(&f->tape_index, __ATOMIC_RELAXED) >= POOL) return T_OVERFLOW;   /* :389-391 */
            out_index = __atomic_fetch_add(&f->tape_index, MPR_SUBTAPE_CHUNK, __ATOMIC_RELAXED);
            out_offset = MPR_SUBTAPE_CHUNK;
            if ((int64_t)out_index + out_offset >= POOL) return T_OVERFLOW;                      /* :396-398 */
            --out_offset;
            const int32_t delta = prev_index - (out_index + out_offset);
            tape_data[out_index + out_offset] = mpr_cl_make(MPR_OP_JUMP, 0, 0, 0, (uint32_t)delta);
            /* the reference only rewrites the op byte and the jump word of the old slot 0;
             * chunks come from fresh pool memory, digest functions ignore JUMP payload bytes */
            tape_data[prev_index] = mpr_cl_make(MPR_OP_JUMP, 0, 0, 0, (uint32_t)(-delta));
            *written += 2;
            --out_offset;
        }

        active[o] = 0;
        if (choice == 0) {
            const uint32_t l = mpr_cl_lhs(d), r = mpr_cl_rhs(d);
            if (l) active[l] = 1;
            if (r) active[r] = 1;
        } else if (choice == 1) {
            const uint32_t l = mpr_cl_lhs(d);
            active[l] = 1;
            if (l == o) { ++out_offset; continue; }
            d = (d & ~0xFFull) | MPR_OP_COPY_LHS;
        } else if (choice == 2) {
            const uint32_t r = mpr_cl_rhs(d);
            if (r) {
                active[r] = 1;
                if (r == o) { ++out_offset; continue; }
                d = (d & ~0xFFull) | MPR_OP_COPY_RHS;
            } else {
                d = (d & ~0xFFull) | MPR_OP_COPY_IMM;
            }
        }
        tape_data[out_index + out_offset] = d;
        ++*written;
    }

    out_offset--;
    tape_data[out_index + out_offset] = *data;     /* head: copy of the parent's head */
    ++*written;
    tile->tape = out_index + out_offset;
    return T_PUSHED;
}

/* Heatmap frames — eval_tiles_i_heatmap, src/context.cu:1622-1632 and :1817-1826: a tile's walk is
 * spread over the pixels of its xy footprint.  In 3-D several tiles of one pixel column add
 * concurrently (atomicAdd upstream), so the order of the float additions is timing dependent. */
static void heat_splat(float* heat, int32_t S, int32_t position, int32_t tps, int32_t tile_px, int64_t work)
{
    const int4_ pos = unpack(position, tps);
    const float v = (float)(uint32_t)work / (float)(tile_px * tile_px);
    for (int32_t y = 0; y < tile_px; ++y) {
        for (int32_t x = 0; x < tile_px; ++x) {
            float* const h = &heat[(pos.x * tile_px + x) + (size_t)(pos.y * tile_px + y) * S];
#pragma omp atomic
            *h += v;
        }
    }
}

/* ---- float walk of one tape for one point — src/context.cu:866-921 ---- */
static float eval_point_f(const uint64_t* tape_data, int32_t tape, float x, float y, float z,
                          int64_t* words)
{
    float slots[256];
    const uint64_t head0 = tape_data[0];
    slots[mpr_cl_out(head0)] = x;
    slots[mpr_cl_lhs(head0)] = y;
    slots[mpr_cl_rhs(head0)] = z;
    const uint64_t* data = &tape_data[tape];
    for (;;) {
        const uint64_t d = *++data;
        ++*words;
        const uint32_t op = mpr_cl_op(d);
        if (!op) break;
        if (op == MPR_OP_JUMP) { data += mpr_cl_jump(d); continue; }
        slots[mpr_cl_out(d)] = float_clause(op, slots[mpr_cl_lhs(d)], slots[mpr_cl_rhs(d)], imm_of(d));
    }
    return slots[mpr_cl_out(*data)];
}

static deriv eval_point_d(const uint64_t* tape_data, int32_t tape, float x, float y, float z,
                          int64_t* words)
{
    deriv slots[256];
    const uint64_t head0 = tape_data[0];
    /* src/context.cu:1021-1031: value first, then the unit partials (an unused axis is slot 0) */
    slots[mpr_cl_out(head0)] = d_const(x);
    slots[mpr_cl_lhs(head0)] = d_const(y);
    slots[mpr_cl_rhs(head0)] = d_const(z);
    slots[mpr_cl_out(head0)].dx = 1.0f;
    slots[mpr_cl_lhs(head0)].dy = 1.0f;
    slots[mpr_cl_rhs(head0)].dz = 1.0f;
    const uint64_t* data = &tape_data[tape];
    for (;;) {
        const uint64_t d = *++data;
        ++*words;
        const uint32_t op = mpr_cl_op(d);
        if (!op) break;
        if (op == MPR_OP_JUMP) { data += mpr_cl_jump(d); continue; }
        slots[mpr_cl_out(d)] = deriv_clause(op, slots[mpr_cl_lhs(d)], slots[mpr_cl_rhs(d)], imm_of(d));
    }
    return slots[mpr_cl_out(*data)];
}

static inline uint8_t f2u8(float v)
{   /* CUDA float -> unsigned char: truncate toward zero, saturate, NaN -> 0 */
    if (!(v == v)) return 0;
    if (v <= 0.0f) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)(int)v;
}

/* ===================================================================================== */
/* the frame                                                                             */
/* ===================================================================================== */
static void grow_tiles(orc_frame* f, int stage, size_t n)
{
    free(f->tiles[stage]);
    f->tiles[stage] = (mpr_tile_node*)malloc(sizeof(mpr_tile_node) * (n ? n : 1));
    f->tiles_n[stage] = n;
}

orc_frame* orc_render(const uint64_t* tape, int32_t length, int32_t dim, int32_t S, const float* mat,
                      float z, int64_t pool_clauses, int32_t threads, const int32_t* owner,
                      int32_t rank, int32_t flags)
{
    if (!tape || length < 2 || (dim != 2 && dim != 3) || S < 64 || S % 64) return NULL;
    /* brute force: 2-D the reference's render2D_brute (:1461-1508); 3-D — the reference has none — the same idea: every 4^3 tile of
     * the volume straight to the float pass with the root tape, normals from the root tape: what a renderer without a hierarchy
     * draws (tests: where the reference's hierarchy draws the same, and where — a NaN end in its intervals — it does not) */
    const int brute = (flags & 1) != 0;
    const int skip_normals = (flags & 2) != 0;
    /* heatmap: 1 = as executed here, 2 = lower bound, 3 = upper bound over the timing-dependent
     * parts of a 3-D frame (a tile culled mid-stage by a neighbour's fill does not push; a voxel
     * pair is skipped when the heightmap it reads is already above it) */
    const int heat_mode = brute ? 0 : (flags & 4) ? 1 : (flags & 8) ? 2 : (flags & 16) ? 3 : 0;
    const int want_heat = heat_mode != 0;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    threads = 1;
#endif
    orc_frame* f = (orc_frame*)calloc(1, sizeof(orc_frame));
    f->S = S;
    f->dim = dim;
    f->c.threads = threads;
    if (pool_clauses <= 0) pool_clauses = (int64_t)MPR_NUM_SUBTAPES_BIG * MPR_SUBTAPE_CHUNK;
    if (pool_clauses > 0x7FFFFFFF) pool_clauses = 0x7FFFFFFF;   /* tape indices are int32 */
    f->pool_cap = pool_clauses;
    f->pool = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(pool_clauses > length ? pool_clauses : length));
    memcpy(f->pool, tape, sizeof(uint64_t) * (size_t)length);   /* src/context.cu:1139-1142 */
    f->tape_index = length;
    for (int i = 0; i < 4; ++i) {                               /* src/context.cpp:21-27 */
        const int ts = 64 >> (2 * i);
        f->filled_n[i] = (size_t)(S / ts) * (S / ts);
        f->filled[i] = (int32_t*)calloc(f->filled_n[i], sizeof(int32_t));
    }
    f->normals = (uint32_t*)calloc((size_t)S * S, sizeof(uint32_t));
    if (want_heat) f->heat = (float*)calloc((size_t)S * S, sizeof(float));

    int64_t F_tiles = 0, F_vox = 0, F_norm = 0, R = 0, W = 0, LC = 0, W_surv = 0, F_vox_min = 0, F_vox_max = 0;
    int slots_exceeded = 0, overflowed = 0;

    /* which stages run: 3-D 0,1,2 (64/16/4 px, x4); 2-D 0,2 (64/8 px, x8) — :1164-1165, :1310 */
    int stage_list[3], nstages;
    if (brute) { nstages = 0; }
    else if (dim == 3) { stage_list[0] = 0; stage_list[1] = 1; stage_list[2] = 2; nstages = 3; }
    else { stage_list[0] = 0; stage_list[1] = 2; nstages = 2; }

    /* preload_tiles — :45-57 (+ optional column ownership for the multi-GPU mode) */
    size_t count;
    if (!brute) {
        const int32_t t0 = S / 64;
        count = (size_t)t0 * t0 * (dim == 3 ? t0 : 1);
        grow_tiles(f, 0, count);
        for (size_t i = 0; i < count; ++i) {
            f->tiles[0][i].position = (int32_t)i;
            f->tiles[0][i].tape = 0;
            f->tiles[0][i].next = -1;
            if (owner && owner[i % ((size_t)t0 * t0)] != rank) f->tiles[0][i].position = -1;
        }
    } else {
        /* render2D_brute — :1461-1508: every 8x8 tile goes straight to the pixel pass (3-D: every 4^3 tile) */
        const int32_t t8 = S / (dim == 3 ? 4 : 8);
        count = (size_t)t8 * t8 * (dim == 3 ? t8 : 1);
        grow_tiles(f, 3, count);
        for (size_t i = 0; i < count; ++i) {
            f->tiles[3][i].position = (int32_t)i;
            f->tiles[3][i].tape = 0;
            f->tiles[3][i].next = -1;
        }
    }

    for (int si = 0; si < nstages; ++si) {
        const int i = stage_list[si];
        const int last = (si == nstages - 1);
        const int next = (dim == 3) ? i + 1 : (i ? 3 : 2);
        const int32_t tile_size_px = (dim == 3) ? (64 >> (2 * i)) : (i ? 8 : 64);
        const int32_t tps = S / tile_size_px;
        mpr_tile_node* tiles = f->tiles[i];
        int32_t* image = f->filled[i];
        const int cs = (dim == 3) ? i : si;   /* counter slot */
        f->c.tiles_in[cs] = (int64_t)count;

        /* mask_filled_tiles before evaluation (3-D) — :1335 */
        int64_t masked = 0;
        if (dim == 3) {
            for (size_t t = 0; t < count; ++t) {
                if (tiles[t].position == -1) continue;
                const int4_ pos = unpack(tiles[t].position, tps);
                if (image[pos.w] > pos.z) { tiles[t].position = -1; masked++; }
            }
        }

        /* heatmap lower bound: backward work only of tiles that also survive the stage */
        int32_t* pushed_words = (heat_mode == 2) ? (int32_t*)calloc(count ? count : 1, sizeof(int32_t)) : NULL;
        int32_t* wr_words = (int32_t*)calloc(count ? count : 1, sizeof(int32_t));      /* words each tile's push wrote */

        /* eval_tiles_i — :1185 / :1342.  Groups of 64 consecutive tiles share a tape. */
        int64_t n_empty = 0, n_filled = 0, n_pushed = 0, n_ambig = 0;
        const size_t ngroups = (count + 63) / 64;
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads) \
    reduction(+ : n_empty, n_filled, n_pushed, n_ambig, masked, F_tiles, R, W, LC) \
    reduction(| : slots_exceeded, overflowed)
        for (size_t g = 0; g < ngroups; ++g) {
            int64_t gfwd = 0, gbwd = 0;
            const size_t hi = (g * 64 + 64 < count) ? g * 64 + 64 : count;
            for (size_t t = g * 64; t < hi; ++t) {
                if (tiles[t].position == -1) continue;   /* :204-206 */
                int64_t fwd = 0, bwd = 0, wr = 0, nc = 0;
                int se = 0, would_push = 0;
                const int32_t position0 = tiles[t].position;
                const int r = eval_tile_i(f, dim, image, tps, &tiles[t], mat, z, &fwd, &bwd, &wr, &nc, &se, &would_push);
                if (f->heat) {
                    /* forward words without the terminator (:1551-1556), then, for a tile that pushed a
                     * tape, the words of the backward walk (:1700-1704), which visits the same words */
                    heat_splat(f->heat, S, position0, tps, tile_size_px, fwd - 1);
                    if (heat_mode == 2) {
                        if (r == T_PUSHED) pushed_words[t] = (int32_t)(bwd - 1);
                    } else if (r == T_PUSHED) {
                        heat_splat(f->heat, S, position0, tps, tile_size_px, bwd - 1);
                    } else if (heat_mode == 3 && r == T_MASKED && would_push) {
                        heat_splat(f->heat, S, position0, tps, tile_size_px, fwd - 1);
                    }
                }
                slots_exceeded |= se;
                LC += fwd - 1;          /* every word but the terminator, per tile */
                W += wr;
                if (r == T_PUSHED) wr_words[t] = (int32_t)wr;
                if (fwd > gfwd) gfwd = fwd;
                if (bwd > gbwd) gbwd = bwd;
                switch (r) {
                    case T_EMPTY: n_empty++; break;
                    case T_MASKED: masked++; break;
                    case T_FILLED: n_filled++; break;
                    case T_PUSHED: n_pushed++; n_ambig++; break;
                    case T_OVERFLOW: overflowed |= 1; n_ambig++; break;
                    default: n_ambig++; break;
                }
            }
            F_tiles += gfwd;
            R += gbwd;
        }

        /* mask_filled_tiles after evaluation (3-D) — :1359 */
        if (dim == 3) {
            for (size_t t = 0; t < count; ++t) {
                if (tiles[t].position == -1) continue;
                const int4_ pos = unpack(tiles[t].position, tps);
                if (image[pos.w] > pos.z) { tiles[t].position = -1; masked++; n_ambig--; }
            }
        }

        for (size_t t = 0; t < count; ++t)
            if (tiles[t].position != -1) W_surv += wr_words[t];
        free(wr_words);
        if (pushed_words) {
            for (size_t t = 0; t < count; ++t) {
                if (pushed_words[t] && tiles[t].position != -1)
                    heat_splat(f->heat, S, tiles[t].position, tps, tile_size_px, pushed_words[t]);
            }
            free(pushed_words);
        }

        /* assign_next_nodes — :512-551 (here: in list order; the reference's order is
         * timing dependent, only the SET of active tiles is comparable) */
        int32_t active = 0;
        for (size_t t = 0; t < count; ++t) tiles[t].next = (tiles[t].position != -1) ? active++ : -1;

        f->c.tiles_empty[cs] = n_empty;
        f->c.tiles_filled[cs] = n_filled;
        f->c.tiles_masked[cs] = masked;
        f->c.tiles_active[cs] = active;
        f->c.tiles_pushed[cs] = n_pushed;

        /* subdivide_active_tiles / copy_active_tiles — :564-651 */
        const size_t next_count = last ? (size_t)active : (size_t)active * 64;
        grow_tiles(f, next, next_count);
        mpr_tile_node* out = f->tiles[next];
        const int sub = (dim == 3) ? 4 : 8;
        for (size_t t = 0; t < count; ++t) {
            if (tiles[t].next == -1) continue;
            if (!last) {
                const int4_ pos = unpack(tiles[t].position, tps);
                const int32_t sps = tps * sub;
                for (int s = 0; s < 64; ++s) {
                    const int4_ sp = unpack(s, sub);
                    int32_t nt;
                    if (dim == 3) {
                        nt = (pos.x * 4 + sp.x) + (pos.y * 4 + sp.y) * sps + (pos.z * 4 + sp.z) * sps * sps;
                    } else {
                        nt = (pos.x * 8 + sp.x) + (pos.y * 8 + sp.y) * sps;
                    }
                    mpr_tile_node* o = &out[(size_t)tiles[t].next * 64 + s];
                    o->position = nt;
                    o->tape = tiles[t].tape;
                    o->next = -1;
                }
            } else {
                mpr_tile_node* o = &out[tiles[t].next];
                o->position = tiles[t].position;
                o->tape = tiles[t].tape;
                o->next = -1;
                tiles[t].next = -1;   /* :650 */
            }
        }

        /* copy_filled — :664-692 */
        {
            const int32_t nts = tile_size_px / sub;
            const int32_t ns = S / nts;
            const int32_t* prev = image;
            int32_t* img = f->filled[next];
            for (int32_t y = 0; y < ns; ++y) {
                for (int32_t x = 0; x < ns; ++x) {
                    const int32_t t = prev[x / sub + (y / sub) * (ns / sub)];
                    if (t) img[x + y * ns] = (dim == 3) ? t * 4 + 3 : 1;
                }
            }
        }
        count = next_count;
    }

    /* ---- per-voxel / per-pixel float pass — :707-964 ---- */
    f->c.voxel_tiles = (int64_t)count;
    {
        const mpr_tile_node* vt = f->tiles[3];
        int32_t* image = f->filled[3];
        const int sub = (dim == 3) ? 4 : 8;
        const int32_t tps = S / sub;
        const float size_recip = 1.0f / (float)(tps * sub);
        /* heatmap bounds (3-D): the walk of every smallest tile, for the second pass below */
        int32_t* tile_words = (dim == 3 && heat_mode >= 2) ? (int32_t*)calloc(count ? count : 1, sizeof(int32_t)) : NULL;
        int32_t* walk_words = (dim == 3) ? (int32_t*)calloc(count ? count : 1, sizeof(int32_t)) : NULL;   /* F bounds */
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads) reduction(+ : F_vox, LC)
        for (size_t t = 0; t < count; ++t) {
            const int4_ pos = unpack(vt[t].position, tps);
            int64_t words_max = 0;
            if (tile_words || walk_words) {
                int64_t w = 0;
                (void)eval_point_f(f->pool, vt[t].tape, 0.0f, 0.0f, 0.0f, &w);
                if (tile_words) tile_words[t] = (int32_t)(w - 1);
                if (walk_words) walk_words[t] = (int32_t)w;
            }
            for (int s = 0; s < 64; ++s) {
                const int4_ sp = unpack(s, sub);
                int64_t words = 0;
                if (dim == 3) {
                    const int32_t px = pos.x * 4 + sp.x, py = pos.y * 4 + sp.y, pz = pos.z * 4 + sp.z;
                    /* :852-864 — the pair (pz, pz+2) is skipped when image >= pz_low + 2 */
                    const int32_t pz_low = pos.z * 4 + (sp.z & 1);
                    if (__atomic_load_n(&image[px + py * S], __ATOMIC_RELAXED) >= pz_low + 2) continue;
                    const float fx = ((px + 0.5f) * size_recip - 0.5f) * 2.0f;
                    const float fy = ((py + 0.5f) * size_recip - 0.5f) * 2.0f;
                    const float fz = ((pz + 0.5f) * size_recip - 0.5f) * 2.0f;
                    const float fw = M4(3, 0) * fx + M4(3, 1) * fy + M4(3, 2) * fz + M4(3, 3);
                    float v[3];
                    for (int k = 0; k < 3; ++k)
                        v[k] = (M4(k, 0) * fx + M4(k, 1) * fy + M4(k, 2) * fz + M4(k, 3)) / fw;
                    const float r = eval_point_f(f->pool, vt[t].tape, v[0], v[1], v[2], &words);
                    if (r < 0.0f) atomic_max_i32(&image[px + py * S], pz);
                    /* eval_voxels_f_heatmap, :1960-1962: one add per reference thread (voxels z and
                     * z + 2 share a thread), i.e. by the voxels with sp.z < 2 */
                    if (heat_mode == 1 && sp.z < 2) {
                        float* const h = &f->heat[px + (size_t)py * S];
                        const float w = (float)(uint32_t)(words - 1);
#pragma omp atomic
                        *h += w;
                    }
                } else {
                    const int32_t px = pos.x * 8 + sp.x, py = pos.y * 8 + sp.y;
                    const float fx = ((px + 0.5f) * size_recip - 0.5f) * 2.0f;
                    const float fy = ((py + 0.5f) * size_recip - 0.5f) * 2.0f;
                    const float fw = M3(2, 0) * fx + M3(2, 1) * fy + M3(2, 2);
                    float v[2];
                    for (int k = 0; k < 2; ++k) v[k] = (M3(k, 0) * fx + M3(k, 1) * fy + M3(k, 2)) / fw;
                    const float r = eval_point_f(f->pool, vt[t].tape, v[0], v[1], z, &words);
                    if (r < 0.0f) image[px + py * S] = 1;
                    /* :1977-1980: half of the thread's walk to each of its two pixels */
                    if (f->heat) f->heat[px + (size_t)py * S] += (float)(uint32_t)(words - 1) / 2.0f;
                }
                LC += words - 1;
                if (words > words_max) words_max = words;
            }
            F_vox += words_max;
        }
        if (walk_words) {
            /* the float pass's F over all execution orders: a tile is walked unless all 32 voxel pairs are skipped, and a pair
             * that the FINAL heightmap leaves visible is visible whenever the tile runs */
            for (size_t t = 0; t < count; ++t) {
                const int4_ pos = unpack(vt[t].position, tps);
                int visible = 0;
                for (int s = 0; s < 32 && !visible; ++s) {
                    const int4_ sp = unpack(s, sub);
                    const int32_t px = pos.x * 4 + sp.x, py = pos.y * 4 + sp.y;
                    const int32_t pz_low = pos.z * 4 + (sp.z & 1);
                    if (image[px + py * S] < pz_low + 2) visible = 1;
                }
                F_vox_max += walk_words[t];
                if (visible) F_vox_min += walk_words[t];
            }
            free(walk_words);
        } else {
            F_vox_min = F_vox_max = F_vox;
        }
        if (tile_words) {
            /* lower bound: only the voxel pairs that no order of execution can skip (the final
             * heightmap is still below them); upper bound: nothing is skipped */
            for (size_t t = 0; t < count; ++t) {
                const int4_ pos = unpack(vt[t].position, tps);
                for (int s = 0; s < 32; ++s) {
                    const int4_ sp = unpack(s, sub);
                    const int32_t px = pos.x * 4 + sp.x, py = pos.y * 4 + sp.y;
                    const int32_t pz_low = pos.z * 4 + (sp.z & 1);
                    if (heat_mode == 2 && image[px + py * S] >= pz_low + 2) continue;
                    f->heat[px + (size_t)py * S] += (float)(uint32_t)tile_words[t];
                }
            }
            free(tile_words);
        }
    }

    /* ---- normals — eval_pixels_d, :978-1132 ---- */
    if (dim == 3 && !skip_normals) {
        const int32_t* image = f->filled[3];
        const mpr_tile_node *tiles = f->tiles[0], *subtiles = f->tiles[1], *microtiles = f->tiles[2];
        int64_t npix = 0;
        const int32_t patches = S / 8;
#pragma omp parallel for schedule(dynamic, 8) num_threads(threads) reduction(+ : F_norm, LC, npix)
        for (int32_t patch = 0; patch < patches * patches; ++patch) {
            /* F accounting: distinct tapes inside one 8x8 patch */
            int32_t seen_tape[64];
            int64_t seen_words[64];
            int nseen = 0;
            for (int s = 0; s < 64; ++s) {
                const int32_t px = (patch % patches) * 8 + (s % 8), py = (patch / patches) * 8 + (s / 8);
                const int32_t pxy = px + py * S;
                int32_t pz = image[pxy];
                if (pz == 0) continue;
                if (pz < S - 1) pz += 1;
                const float size_recip = 1.0f / (float)S;
                const float fx = ((px + 0.5f) * size_recip - 0.5f) * 2.0f;
                const float fy = ((py + 0.5f) * size_recip - 0.5f) * 2.0f;
                const float fz = ((pz + 0.5f) * size_recip - 0.5f) * 2.0f;
                const float fw = M4(3, 0) * fx + M4(3, 1) * fy + M4(3, 2) * fz + M4(3, 3);
                float v[3];
                for (int k = 0; k < 3; ++k)
                    v[k] = (M4(k, 0) * fx + M4(k, 1) * fy + M4(k, 2) * fz + M4(k, 3)) / fw;

                /* :1034-1066 — deepest tile's tape */
                int32_t tape_at;
                const int32_t t64 = S / 64;
                const int32_t tile = px / 64 + (py / 64) * t64 + (pz / 64) * t64 * t64;
                if (brute) {
                    tape_at = 0;
                } else if (tiles[tile].next == -1) {
                    tape_at = tiles[tile].tape;
                } else {
                    const int32_t subtile = tiles[tile].next * 64 + (px % 64) / 16 + ((py % 64) / 16) * 4 +
                                            ((pz % 64) / 16) * 16;
                    if (subtiles[subtile].next == -1) {
                        tape_at = subtiles[subtile].tape;
                    } else {
                        const int32_t micro = subtiles[subtile].next * 64 + (px % 16) / 4 +
                                              ((py % 16) / 4) * 4 + ((pz % 16) / 4) * 16;
                        tape_at = microtiles[micro].tape;
                    }
                }
                int64_t words = 0;
                const deriv r = eval_point_d(f->pool, tape_at, v[0], v[1], v[2], &words);
                LC += words - 1;
                npix++;
                int k;
                for (k = 0; k < nseen; ++k) if (seen_tape[k] == tape_at) break;
                if (k == nseen) { seen_tape[nseen] = tape_at; seen_words[nseen] = words; nseen++; }
                /* :1123-1131 */
                const float norm = sqrtf(r.dx * r.dx + r.dy * r.dy + r.dz * r.dz);
                const uint8_t dx = f2u8((r.dx / norm) * 127 + 128);
                const uint8_t dy = f2u8((r.dy / norm) * 127 + 128);
                const uint8_t dz = f2u8((r.dz / norm) * 127 + 128);
                f->normals[pxy] = (0xFFu << 24) | ((uint32_t)dz << 16) | ((uint32_t)dy << 8) | dx;
            }
            for (int k = 0; k < nseen; ++k) F_norm += seen_words[k];
        }
        f->c.normal_pixels = npix;
    }

    f->c.clauses_fwd_tiles = F_tiles;
    f->c.clauses_fwd_voxels = F_vox;
    f->c.clauses_fwd_normals = F_norm;
    f->c.clauses_fwd = F_tiles + F_vox + F_norm;
    f->c.clauses_bwd = R;
    f->c.clauses_written = W;
    f->c.clauses_written_survivors = W_surv;
    f->c.clauses_fwd_voxels_min = F_vox_min;
    f->c.clauses_fwd_voxels_max = F_vox_max;
    f->c.lane_clauses = LC;
    f->c.tape_index = f->tape_index;
    f->c.pool_overflowed = overflowed;
    f->c.slots_exceeded = slots_exceeded;
    if (f->heat) {
        /* src/context.cu:2140-2144, :2334-2338 */
        const float clauses = (float)(length - 2);
        for (size_t i = 0; i < (size_t)S * S; ++i) f->heat[i] /= clauses;
    }
    return f;
}

void orc_frame_free(orc_frame* f)
{
    if (!f) return;
    for (int i = 0; i < 4; ++i) { free(f->filled[i]); free(f->tiles[i]); }
    free(f->normals);
    free(f->heat);
    free(f->pool);
    free(f);
}
const int32_t* orc_filled(const orc_frame* f, int32_t stage, size_t* n)
{
    if (!f || stage < 0 || stage > 3) return NULL;
    if (n) *n = f->filled_n[stage];
    return f->filled[stage];
}
const uint32_t* orc_normals(const orc_frame* f, size_t* n)
{
    if (!f) return NULL;
    if (n) *n = (size_t)f->S * f->S;
    return f->normals;
}
const float* orc_heatmap(const orc_frame* f, size_t* n)
{
    if (!f || !f->heat) return NULL;
    if (n) *n = (size_t)f->S * f->S;
    return f->heat;
}
const mpr_tile_node* orc_tiles(const orc_frame* f, int32_t stage, size_t* n)
{
    if (!f || stage < 0 || stage > 3) return NULL;
    if (n) *n = f->tiles_n[stage];
    return f->tiles[stage];
}
const uint64_t* orc_tape_pool(const orc_frame* f, int32_t* tape_index)
{
    if (!f) return NULL;
    if (tape_index) *tape_index = f->tape_index;
    return f->pool;
}
void orc_get_counters(const orc_frame* f, orc_counters* out) { if (f && out) *out = f->c; }

/* ---- digests ---- */
int32_t orc_tape_digest(const uint64_t* pool, int64_t pool_len, int32_t head, uint64_t* hash)
{
    uint64_t h = 0xcbf29ce484222325ull;
    int32_t n = 0;
    int64_t p = head;
    if (p < 0 || p >= pool_len) { if (hash) *hash = 0; return -1; }
    for (int64_t guard = 0; guard < pool_len + 8; ++guard) {
        ++p;
        if (p < 0 || p >= pool_len) { if (hash) *hash = 0; return -1; }
        const uint64_t d = pool[p];
        const uint32_t op = mpr_cl_op(d);
        if (op == MPR_OP_JUMP) { p += mpr_cl_jump(d); continue; }
        for (int b = 0; b < 8; ++b) { h ^= (d >> (8 * b)) & 0xFF; h *= 0x100000001b3ull; }
        if (!op) break;
        n++;
    }
    if (hash) *hash = h;
    return n;
}
void orc_tiles_digest(const uint64_t* pool, int64_t pool_len, const mpr_tile_node* tiles, size_t n,
                      int32_t* len, uint64_t* hash)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        uint64_t h = 0;
        const int32_t l = orc_tape_digest(pool, pool_len, tiles[i].tape, &h);
        if (len) len[i] = l;
        if (hash) hash[i] = h;
    }
}

/* ---- primitives ---- */
int32_t orc_interval_op(int32_t op, float a_lo, float a_hi, float b_lo, float b_hi, float imm,
                        float* out_lo, float* out_hi)
{
    int c = 0;
    fesetround(FE_UPWARD);
    const ival r = interval_clause((uint32_t)op, iv(a_lo, a_hi), iv(b_lo, b_hi), imm, &c);
    fesetround(FE_TONEAREST);
    *out_lo = r.lo;
    *out_hi = r.hi;
    return c;
}
void orc_interval_op_n(int32_t op, int32_t n, const float* a_lo, const float* a_hi, const float* b_lo,
                       const float* b_hi, float imm, float* out_lo, float* out_hi, int32_t* choice)
{
    for (int32_t i = 0; i < n; ++i) {
        const int32_t c = orc_interval_op(op, a_lo[i], a_hi[i], b_lo ? b_lo[i] : 0.0f,
                                          b_hi ? b_hi[i] : 0.0f, imm, &out_lo[i], &out_hi[i]);
        if (choice) choice[i] = c;
    }
}
float orc_float_op(int32_t op, float a, float b, float imm) { return float_clause((uint32_t)op, a, b, imm); }
void orc_float_op_n(int32_t op, int32_t n, const float* a, const float* b, float imm, float* out)
{
    for (int32_t i = 0; i < n; ++i) out[i] = float_clause((uint32_t)op, a[i], b ? b[i] : 0.0f, imm);
}
void orc_deriv_op_n(int32_t op, int32_t n, const float* a4, const float* b4, float imm, float* out4)
{
    for (int32_t i = 0; i < n; ++i) {
        deriv a = {a4[4 * i], a4[4 * i + 1], a4[4 * i + 2], a4[4 * i + 3]};
        deriv b = {0, 0, 0, 0};
        if (b4) { b.dx = b4[4 * i]; b.dy = b4[4 * i + 1]; b.dz = b4[4 * i + 2]; b.v = b4[4 * i + 3]; }
        const deriv r = deriv_clause((uint32_t)op, a, b, imm);
        out4[4 * i] = r.dx; out4[4 * i + 1] = r.dy; out4[4 * i + 2] = r.dz; out4[4 * i + 3] = r.v;
    }
}

static inline uint64_t splitmix(uint64_t* s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline int same_bits(float a, float b)
{
    if (a != a && b != b) return 1;
    return orc_bits(a) == orc_bits(b);
}
int64_t orc_selftest_rounding(int64_t n, uint64_t seed)
{
    int64_t bad = 0;
    for (int64_t i = 0; i < n; ++i) {
        /* random bit patterns: covers subnormals, infinities, NaN, huge exponent gaps */
        uint64_t r = splitmix(&seed);
        float a = orc_float((uint32_t)r), b = orc_float((uint32_t)(r >> 32));
        if (i & 1) {   /* nearby magnitudes, where cancellation and ties happen */
            const uint64_t q = splitmix(&seed);
            b = orc_float((orc_bits(a) & 0xFF800000u) ^ (uint32_t)(q & 0x80FFFFFFu));
        }
        volatile float x = a, y = b;
        fesetround(FE_DOWNWARD);
        volatile float add_d = x + y, sub_d = x - y, mul_d = x * y, div_d = x / y, sq_d = sqrtf(x);
        volatile double dd = (double)x * 1.0000001; volatile float cv_d = (float)dd;
        fesetround(FE_UPWARD);
        volatile float sub_u = x - y;
        const int ok = same_bits(add_d, rd_add(a, b)) && same_bits(sub_d, rd_sub(a, b)) &&
                       same_bits(mul_d, rd_mul(a, b)) && same_bits(div_d, rd_div(a, b)) &&
                       same_bits(sq_d, rd_sqrt(a)) && same_bits(sub_u, ru_sub(a, b)) &&
                       same_bits(cv_d, d2f_rd(dd));
        fesetround(FE_TONEAREST);
        if (!ok) bad++;
    }
    return bad;
}

void orc_fmath_n(int32_t which, int32_t n, const float* x, float* out)
{
    for (int32_t i = 0; i < n; ++i) {
        switch (which) {
            case 0: out[i] = orc_sinf(x[i]); break;
            case 1: out[i] = orc_cosf(x[i]); break;
            case 2: out[i] = orc_asinf(x[i]); break;
            case 3: out[i] = orc_acosf(x[i]); break;
            case 4: out[i] = orc_atanf(x[i]); break;
            case 5: out[i] = orc_expf(x[i]); break;
            case 6: out[i] = orc_logf(x[i]); break;
            default: out[i] = NAN;
        }
    }
}


/* ====================================================================================== */
/* mpr::Effects — reference src/effects.cu:17-286, restated in oracle/orc_effects.h (the     */
/* oracle's own text; the product's include/mpr_effects_*.h are not used here).              */
/* ====================================================================================== */
#include "orc_effects.h"

void orc_effects_tables(float* kernel, float* rvecs)
{
    const int old = fegetround();
    fesetround(FE_TONEAREST);
    orc_v3 k[64], r[256];
    orc_fx_tables(k, r);
    if (kernel) memcpy(kernel, k, sizeof(k));
    if (rvecs) memcpy(rvecs, r, sizeof(r));
    fesetround(old);
}
/* the C library's own rand() sequence for `seed` (private state; rand() == random() in glibc) */
void orc_glibc_rand(uint32_t seed, int32_t n, int32_t* out)
{
    struct random_data rd;
    char state[128];
    memset(&rd, 0, sizeof rd);
    memset(state, 0, sizeof state);
    initstate_r(seed, state, sizeof state, &rd);
    for (int32_t i = 0; i < n; ++i) random_r(&rd, &out[i]);
}
/* drawSSAO (which = 0, :246-263): tmp = raw occlusion, image = blurred.
 * drawShaded (which = 1, :265-286): image = raw occlusion, tmp = blurred, then image = shading. */
void orc_effects(int32_t which, int32_t size, const int32_t* depth, const uint32_t* normals, int32_t* image, int32_t* tmp)
{
    const int old = fegetround();
    fesetround(FE_TONEAREST);
    orc_v3 kernel[64], rvecs[256];
    orc_fx_tables(kernel, rvecs);
    const size_t n = (size_t)size * size;
    memset(image, 0, n * sizeof(int32_t));
    memset(tmp, 0, n * sizeof(int32_t));
    if (which == 0) {
        orc_fx_draw_ssao(depth, normals, kernel, rvecs, size, tmp);
        orc_fx_blur_ssao(depth, tmp, size, image);
    } else {
        orc_fx_draw_ssao(depth, normals, kernel, rvecs, size, image);
        orc_fx_blur_ssao(depth, image, size, tmp);
        orc_fx_draw_shaded(depth, normals, tmp, size, image);
    }
    fesetround(old);
}
