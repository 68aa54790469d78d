/*
 * mpr_fmath.h — single-precision elementary functions with ONE bit-exact definition for
 * both the gfx950 kernels and the host (C / C++).
 *
 * Why this exists: the reference's float pass calls the CUDA math library
 * (sinf, cosf, asinf, acosf, atanf, expf, logf, fminf, fmaxf — src/context.cu:888-912 and
 * inc/gpu_deriv.hpp:166-204).  CUDA libm, AMD OCML and glibc are three different
 * implementations that agree only to ~1-2 ulp, so a GPU result could never be compared
 * bit-for-bit with a CPU restatement.  These functions are built only from IEEE-754
 * operations that are correctly rounded on both machines (+ - * / sqrt fma, int<->float
 * conversions, bit casts); compiled with -ffp-contract=off they return identical bits on
 * x86-64 and on gfx950.  Algorithms: Cephes single-precision routines (S. Moshier, public
 * domain, netlib cephes/single), re-expressed with explicit fmaf(); sin/cos use a double
 * precision Cody-Waite reduction.  Measured error vs. a double reference: <= 2 ulp
 * (tests/test_fmath.py).  Rounding mode must be round-to-nearest when these are called.
 *
 * This header is product code only.  The oracle has its own, separately written implementation of the
 * same definition (oracle/orc_fmath.h: necessarily the same operation sequence, since the bits must
 * agree); the independent evidence for these functions is the ulp test of the GPU's outputs against
 * mpmath (tests/test_soundness.py::test_gpu_float_functions_against_mpmath).
 */
#ifndef MPR_FMATH_H
#define MPR_FMATH_H

#include <stdint.h>
#include <string.h>
#include <math.h>

#if defined(__HIPCC__)
#define MPR_HD __host__ __device__ static inline
#else
#define MPR_HD static inline
#endif

MPR_HD uint32_t mpr_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
MPR_HD float mpr_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
MPR_HD int mpr_isnanf(float f) { return (mpr_f2u(f) & 0x7FFFFFFFu) > 0x7F800000u; }

/* fminf / fmaxf with fully specified semantics (IEEE-754-2019 minimumNumber /
 * maximumNumber: a quiet-NaN operand is ignored, -0 orders below +0).  On gfx950 this is
 * exactly v_min_f32 / v_max_f32 (CDNA ISA: NaN operand -> the other one, -0 < +0), so the
 * device build uses the instruction; tests/test_gpu_primitives.py checks the two definitions
 * against each other (signalling NaNs, which no arithmetic produces, excepted). */
#if defined(__HIP_DEVICE_COMPILE__)
MPR_HD float mpr_fminf(float a, float b) { return __builtin_fminf(a, b); }
MPR_HD float mpr_fmaxf(float a, float b) { return __builtin_fmaxf(a, b); }
#else
MPR_HD float mpr_fminf(float a, float b)
{
    if (mpr_isnanf(a)) return b;
    if (mpr_isnanf(b)) return a;
    if (a == b) return (mpr_f2u(a) & 0x80000000u) ? a : b;
    return a < b ? a : b;
}
MPR_HD float mpr_fmaxf(float a, float b)
{
    if (mpr_isnanf(a)) return b;
    if (mpr_isnanf(b)) return a;
    if (a == b) return (mpr_f2u(a) & 0x80000000u) ? b : a;
    return a > b ? a : b;
}
#endif

/* ---- expf ------------------------------------------------------------------------- */
MPR_HD float mpr_expf(float x)
{
    if (mpr_isnanf(x)) return x;
    if (x > 88.72284f) return mpr_u2f(0x7F800000u);
    if (x < -103.98f) return 0.0f;
    /* k = round(x / ln2) via the 1.5*2^23 trick (|x/ln2| < 151) */
    const float t = x * 1.44269504088896341f;
    const float kf = (t + 12582912.0f) - 12582912.0f;
    float r = fmaf(kf, -0.693359375f, x);
    r = fmaf(kf, 2.12194440e-4f, r);
    const float z = r * r;
    float p = 1.9875691500E-4f;
    p = fmaf(p, r, 1.3981999507E-3f);
    p = fmaf(p, r, 8.3334519073E-3f);
    p = fmaf(p, r, 4.1665795894E-2f);
    p = fmaf(p, r, 1.6666665459E-1f);
    p = fmaf(p, r, 5.0000001201E-1f);
    p = fmaf(p, z, r);
    p = p + 1.0f;
    /* scale by 2^k in two exact-or-single-rounding steps (k in [-150, 128]) */
    const int k = (int)kf;
    const int k1 = k / 2, k2 = k - k1;
    const float s1 = mpr_u2f((uint32_t)(k1 + 127) << 23);
    const float s2 = mpr_u2f((uint32_t)(k2 + 127) << 23);
    return (p * s1) * s2;
}

/* ---- logf ------------------------------------------------------------------------- */
MPR_HD float mpr_logf(float x)
{
    uint32_t u = mpr_f2u(x);
    if (mpr_isnanf(x)) return x;
    if (u & 0x80000000u) {
        if ((u & 0x7FFFFFFFu) == 0) return mpr_u2f(0xFF800000u);  /* log(-0) = -inf */
        return mpr_u2f(0x7FC00000u);                               /* log(<0) = NaN */
    }
    if (u == 0) return mpr_u2f(0xFF800000u);
    if (u == 0x7F800000u) return x;
    int e = 0;
    if (u < 0x00800000u) {          /* subnormal: scale by 2^23 (exact) */
        x = x * 8388608.0f;
        u = mpr_f2u(x);
        e = -23;
    }
    /* x = m * 2^e with m in [0.5, 1) */
    e += (int)(u >> 23) - 126;
    float m = mpr_u2f((u & 0x007FFFFFu) | 0x3F000000u);
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = (m + m) - 1.0f;
    } else {
        m = m - 1.0f;
    }
    const float z = m * m;
    float y = 7.0376836292E-2f;
    y = fmaf(y, m, -1.1514610310E-1f);
    y = fmaf(y, m, 1.1676998740E-1f);
    y = fmaf(y, m, -1.2420140846E-1f);
    y = fmaf(y, m, 1.4249322787E-1f);
    y = fmaf(y, m, -1.6668057665E-1f);
    y = fmaf(y, m, 2.0000714765E-1f);
    y = fmaf(y, m, -2.4999993993E-1f);
    y = fmaf(y, m, 3.3333331174E-1f);
    y = (y * m) * z;
    const float fe = (float)e;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(-0.5f, z, y);
    float r = m + y;
    r = fmaf(fe, 0.693359375f, r);
    return r;
}

/* ---- sinf / cosf ------------------------------------------------------------------ */
/* reduce x to r in [-pi/4, pi/4] and quadrant q; double-precision Cody-Waite */
MPR_HD float mpr_trig_reduce(float x, int* q)
{
    const double xd = (double)x;
    const double t = xd * 0.63661977236758134308;            /* 2/pi */
    const double j = (t + 6755399441055744.0) - 6755399441055744.0;
    double r = fma(j, -1.57079632679489655800e+00, xd);       /* pi/2 hi */
    r = fma(j, -6.12323399573676603587e-17, r);               /* pi/2 lo */
    *q = (int)((long long)j & 3);
    return (float)r;
}
MPR_HD float mpr_sin_poly(float r)
{
    const float z = r * r;
    float y = -1.9515295891E-4f;
    y = fmaf(y, z, 8.3321608736E-3f);
    y = fmaf(y, z, -1.6666654611E-1f);
    y = (y * z) * r;
    return y + r;
}
MPR_HD float mpr_cos_poly(float r)
{
    const float z = r * r;
    float y = 2.443315711809948E-005f;
    y = fmaf(y, z, -1.388731625493765E-003f);
    y = fmaf(y, z, 4.166664568298827E-002f);
    y = (y * z) * z;
    y = fmaf(-0.5f, z, y);
    return y + 1.0f;
}
MPR_HD float mpr_sinf(float x)
{
    const uint32_t a = mpr_f2u(x) & 0x7FFFFFFFu;
    if (a >= 0x7F800000u) return mpr_u2f(0x7FC00000u);        /* inf / NaN */
    if (a >= 0x4F000000u) return 0.0f;                         /* |x| >= 2^31: not reduced */
    int q;
    const float r = mpr_trig_reduce(x, &q);
    switch (q) {
        case 0: return mpr_sin_poly(r);
        case 1: return mpr_cos_poly(r);
        case 2: return -mpr_sin_poly(r);
        default: return -mpr_cos_poly(r);
    }
}
MPR_HD float mpr_cosf(float x)
{
    const uint32_t a = mpr_f2u(x) & 0x7FFFFFFFu;
    if (a >= 0x7F800000u) return mpr_u2f(0x7FC00000u);
    if (a >= 0x4F000000u) return 1.0f;
    int q;
    const float r = mpr_trig_reduce(x, &q);
    switch (q) {
        case 0: return mpr_cos_poly(r);
        case 1: return -mpr_sin_poly(r);
        case 2: return -mpr_cos_poly(r);
        default: return mpr_sin_poly(r);
    }
}

/* ---- atanf ------------------------------------------------------------------------ */
MPR_HD float mpr_atanf(float x)
{
    if (mpr_isnanf(x)) return x;
    const uint32_t sign = mpr_f2u(x) & 0x80000000u;
    float a = mpr_u2f(mpr_f2u(x) & 0x7FFFFFFFu);
    float yhi, ylo;                          /* pi/2 or pi/4 as hi + lo */
    if (a > 2.414213562373095f) {           /* tan(3pi/8) */
        yhi = 1.5707963705062866f;
        ylo = -4.371139000186243e-8f;
        a = -(1.0f / a);
    } else if (a > 0.4142135623730950f) {   /* tan(pi/8) */
        yhi = 0.7853981852531433f;
        ylo = -2.1855695000931215e-8f;
        a = (a - 1.0f) / (a + 1.0f);
    } else {
        yhi = 0.0f;
        ylo = 0.0f;
    }
    const float z = a * a;
    float p = 8.05374449538e-2f;
    p = fmaf(p, z, -1.38776856032E-1f);
    p = fmaf(p, z, 1.99777106478E-1f);
    p = fmaf(p, z, -3.33329491539E-1f);
    p = (p * z) * a;
    p = p + ylo;
    p = p + a;
    const float y = yhi + p;
    return mpr_u2f(mpr_f2u(y) ^ sign);
}

/* ---- asinf / acosf ---------------------------------------------------------------- */
MPR_HD float mpr_asin_poly(float x, float z)
{
    float p = 4.2163199048E-2f;
    p = fmaf(p, z, 2.4181311049E-2f);
    p = fmaf(p, z, 4.5470025998E-2f);
    p = fmaf(p, z, 7.4953002686E-2f);
    p = fmaf(p, z, 1.6666752422E-1f);
    p = (p * z) * x;
    return p + x;
}
MPR_HD float mpr_asinf(float x)
{
    if (mpr_isnanf(x)) return x;
    const uint32_t sign = mpr_f2u(x) & 0x80000000u;
    const float a = mpr_u2f(mpr_f2u(x) & 0x7FFFFFFFu);
    if (a > 1.0f) return mpr_u2f(0x7FC00000u);
    if (a < 1.0e-4f) return x;
    float r;
    if (a > 0.5f) {
        const float z = 0.5f * (1.0f - a);
        const float s = sqrtf(z);
        r = mpr_asin_poly(s, z);
        r = r + r;
        r = (1.5707963705062866f - r) + -4.371139000186243e-8f;   /* pi/2 = hi + lo */
    } else {
        r = mpr_asin_poly(a, a * a);
    }
    return mpr_u2f(mpr_f2u(r) ^ sign);
}
MPR_HD float mpr_acosf(float x)
{
    if (mpr_isnanf(x)) return x;
    if (x < -1.0f || x > 1.0f) return mpr_u2f(0x7FC00000u);
    if (x < -0.5f) {
        const float z = 0.5f * (1.0f + x);
        const float s = sqrtf(z);
        const float r = mpr_asin_poly(s, z);
        return (3.1415927410125732f - (r + r)) + -8.742278000372485e-8f;   /* pi = hi + lo */
    }
    if (x > 0.5f) {
        const float z = 0.5f * (1.0f - x);
        const float s = sqrtf(z);
        const float r = mpr_asin_poly(s, z);
        return r + r;
    }
    return (1.5707963705062866f - mpr_asinf(x)) + -4.371139000186243e-8f;
}

#endif
