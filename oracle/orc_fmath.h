/*
 * orc_fmath.h — the ORACLE's own single-precision elementary functions (TEST INFRASTRUCTURE).
 *
 * The reference's float pass and its Deriv type call CUDA's sinf, cosf, asinf, acosf, atanf, expf,
 * logf, fminf, fmaxf (src/context.cu:888-912, inc/gpu_deriv.hpp:166-204).  CUDA's libm is not
 * available anywhere outside a CUDA device, so the product DEFINES these seven functions by a
 * published algorithm (Cephes single precision, S. Moshier, netlib cephes/single: expf.c, logf.c,
 * sinf.c, atanf.c, asinf.c; every multiply-add a fused one, sin/cos reduced in double precision) and
 * implements that definition in include/mpr_fmath.h and in the assembly interpreters.  This file is
 * a second, separately written implementation of the same definition: nothing here is included
 * from, or generated out of, the product's sources.  Where the two disagree in a single bit the
 * GPU-vs-oracle tests fail; how close the definition is to the true functions — the only thing that
 * can be said about CUDA's libm — is tested on the GPU's OUTPUTS against mpmath
 * (tests/test_soundness.py::test_gpu_float_functions_against_mpmath; tests/test_fmath.py for this file).
 *
 * Written as coefficient tables + Horner loops over explicit fmaf(); round-to-nearest must be in
 * effect; compile with -ffp-contract=off.
 */
#ifndef ORC_FMATH_H
#define ORC_FMATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

static inline uint32_t orc_bits(float f) { uint32_t u; memcpy(&u, &f, sizeof u); return u; }
static inline float orc_float(uint32_t u) { float f; memcpy(&f, &u, sizeof f); return f; }
static inline int orc_isnan(float f) { return (orc_bits(f) << 1) > 0xFF000000u; }
#define ORC_QNAN orc_float(0x7FC00000u)
#define ORC_INF orc_float(0x7F800000u)

/* fminf / fmaxf as the hardware the reference ran on and IEEE-754-2019 minimumNumber define them:
 * a NaN operand loses, and of two zeros the negative one is the smaller. */
static inline float orc_fminf(float a, float b)
{
    if (orc_isnan(a)) return b;
    if (orc_isnan(b)) return a;
    if (a < b) return a;
    if (b < a) return b;
    return (orc_bits(a) >> 31) ? a : b;          /* equal: prefer -0 */
}
static inline float orc_fmaxf(float a, float b)
{
    if (orc_isnan(a)) return b;
    if (orc_isnan(b)) return a;
    if (a > b) return a;
    if (b > a) return b;
    return (orc_bits(a) >> 31) ? b : a;          /* equal: prefer +0 */
}

/* Horner evaluation, highest coefficient first: ((c0 * x + c1) * x + c2) ... with fused steps */
static inline float orc_horner(const float* c, int n, float x)
{
    float acc = c[0];
    for (int i = 1; i < n; ++i) acc = fmaf(acc, x, c[i]);
    return acc;
}

/* ---- exp (cephes expf.c): x = k ln2 + r, e^r by a degree-5 polynomial in r, times 2^k -------- */
static inline float orc_expf(float x)
{
    static const float P[6] = {1.9875691500E-4f, 1.3981999507E-3f, 8.3334519073E-3f,
                               4.1665795894E-2f, 1.6666665459E-1f, 5.0000001201E-1f};
    if (orc_isnan(x)) return x;
    if (x > 88.72284f) return ORC_INF;
    if (x < -103.98f) return 0.0f;
    const float magic = 12582912.0f;                               /* 1.5 * 2^23: rounds to an integer */
    const float kf = (x * 1.44269504088896341f + magic) - magic;
    const float r = fmaf(kf, 2.12194440e-4f, fmaf(kf, -0.693359375f, x));   /* ln2 = C1 + C2 */
    const float er = fmaf(orc_horner(P, 6, r), r * r, r) + 1.0f;
    /* 2^k as two factors so that neither over- nor underflows on its own; the first product is exact */
    const int k = (int)kf, ka = k / 2, kb = k - ka;
    return (er * orc_float((uint32_t)(ka + 127) << 23)) * orc_float((uint32_t)(kb + 127) << 23);
}

/* ---- log (cephes logf.c): x = m 2^e, m in [sqrt(1/2), sqrt(2)), log(1 + t) by a degree-8 polynomial */
static inline float orc_logf(float x)
{
    static const float P[9] = {7.0376836292E-2f, -1.1514610310E-1f, 1.1676998740E-1f, -1.2420140846E-1f,
                               1.4249322787E-1f, -1.6668057665E-1f, 2.0000714765E-1f, -2.4999993993E-1f,
                               3.3333331174E-1f};
    if (orc_isnan(x)) return x;
    uint32_t u = orc_bits(x);
    if ((u << 1) == 0) return -ORC_INF;                             /* +-0 */
    if (u >> 31) return ORC_QNAN;                                   /* negative */
    if (u == 0x7F800000u) return x;
    int e = -126;
    if (u < 0x00800000u) {                                          /* subnormal: exact scaling by 2^23 */
        u = orc_bits(x * 8388608.0f);
        e -= 23;
    }
    e += (int)(u >> 23);
    float t = orc_float((u & 0x007FFFFFu) | 0x3F000000u);           /* mantissa in [0.5, 1) */
    if (t < 0.707106781186547524f) {
        e -= 1;
        t = (t + t) - 1.0f;
    } else {
        t = t - 1.0f;
    }
    const float tt = t * t, fe = (float)e;
    float y = (orc_horner(P, 9, t) * t) * tt;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(-0.5f, tt, y);
    return fmaf(fe, 0.693359375f, t + y);
}

/* ---- sin / cos (cephes sinf.c polynomials; argument reduction by pi/2 carried out in double) ---- */
static inline float orc_sin_kernel(float r)
{
    static const float S[3] = {-1.9515295891E-4f, 8.3321608736E-3f, -1.6666654611E-1f};
    const float rr = r * r;
    return (orc_horner(S, 3, rr) * rr) * r + r;
}
static inline float orc_cos_kernel(float r)
{
    static const float C[3] = {2.443315711809948E-005f, -1.388731625493765E-003f, 4.166664568298827E-002f};
    const float rr = r * r;
    return fmaf(-0.5f, rr, (orc_horner(C, 3, rr) * rr) * rr) + 1.0f;
}
/* which = 0: sin, 1: cos */
static inline float orc_sincos(float x, int which)
{
    const uint32_t mag = orc_bits(x) & 0x7FFFFFFFu;
    if (mag >= 0x7F800000u) return ORC_QNAN;                        /* inf, NaN */
    if (mag >= 0x4F000000u) return which ? 1.0f : 0.0f;             /* |x| >= 2^31: left unreduced */
    const double big = 6755399441055744.0;                         /* 1.5 * 2^52 */
    const double xd = (double)x;
    const double n = (xd * 0.63661977236758134308 + big) - big;     /* nearest multiple of pi/2 */
    const double rd = fma(n, -6.12323399573676603587e-17, fma(n, -1.57079632679489655800e+00, xd));
    const float r = (float)rd;
    const int quadrant = (int)(((long long)n + which) & 3);        /* cos(x) = sin(x + pi/2) */
    const float v = (quadrant & 1) ? orc_cos_kernel(r) : orc_sin_kernel(r);
    return (quadrant & 2) ? -v : v;
}
static inline float orc_sinf(float x) { return orc_sincos(x, 0); }
static inline float orc_cosf(float x) { return orc_sincos(x, 1); }

/* ---- atan (cephes atanf.c): two range reductions at tan(pi/8), tan(3pi/8) ----------------------- */
static inline float orc_atanf(float x)
{
    static const float P[4] = {8.05374449538e-2f, -1.38776856032E-1f, 1.99777106478E-1f, -3.33329491539E-1f};
    if (orc_isnan(x)) return x;
    const uint32_t sign = orc_bits(x) & 0x80000000u;
    float a = orc_float(orc_bits(x) ^ sign);
    float hi = 0.0f, lo = 0.0f;                                     /* the constant added back, as hi + lo */
    if (a > 2.414213562373095f) {
        hi = 1.5707963705062866f; lo = -4.371139000186243e-8f;     /* pi/2 */
        a = -(1.0f / a);
    } else if (a > 0.4142135623730950f) {
        hi = 0.7853981852531433f; lo = -2.1855695000931215e-8f;    /* pi/4 */
        a = (a - 1.0f) / (a + 1.0f);
    }
    const float aa = a * a;
    const float y = hi + (((orc_horner(P, 4, aa) * aa) * a + lo) + a);
    return orc_float(orc_bits(y) ^ sign);
}

/* ---- asin / acos (cephes asinf.c) ---------------------------------------------------------------- */
static inline float orc_asin_kernel(float x, float xx)
{
    static const float P[5] = {4.2163199048E-2f, 2.4181311049E-2f, 4.5470025998E-2f, 7.4953002686E-2f,
                               1.6666752422E-1f};
    return (orc_horner(P, 5, xx) * xx) * x + x;
}
#define ORC_PIO2_HI 1.5707963705062866f
#define ORC_PIO2_LO -4.371139000186243e-8f
static inline float orc_asinf(float x)
{
    if (orc_isnan(x)) return x;
    const uint32_t sign = orc_bits(x) & 0x80000000u;
    const float a = orc_float(orc_bits(x) ^ sign);
    if (a > 1.0f) return ORC_QNAN;
    if (a < 1.0e-4f) return x;
    float r;
    if (a > 0.5f) {
        const float h = 0.5f * (1.0f - a);
        r = orc_asin_kernel(sqrtf(h), h);
        r = (ORC_PIO2_HI - (r + r)) + ORC_PIO2_LO;
    } else {
        r = orc_asin_kernel(a, a * a);
    }
    return orc_float(orc_bits(r) ^ sign);
}
static inline float orc_acosf(float x)
{
    if (orc_isnan(x)) return x;
    if (x < -1.0f || x > 1.0f) return ORC_QNAN;
    if (x > 0.5f) {
        const float h = 0.5f * (1.0f - x);
        const float r = orc_asin_kernel(sqrtf(h), h);
        return r + r;
    }
    if (x < -0.5f) {
        const float h = 0.5f * (1.0f + x);
        const float r = orc_asin_kernel(sqrtf(h), h);
        return (3.1415927410125732f - (r + r)) + -8.742278000372485e-8f;      /* pi = hi + lo */
    }
    return (ORC_PIO2_HI - orc_asinf(x)) + ORC_PIO2_LO;
}

#endif
