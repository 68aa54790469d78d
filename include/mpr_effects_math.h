/*
 * mpr_effects_math.h — per-pixel arithmetic of mpr::Effects (reference src/effects.cu:17-209:
 * draw_ssao, blur_ssao, draw_shaded), compiled (without FMA contraction) into the HIP kernels of
 * libmpr_amd.  Product code only: the oracle checks it with a restatement of its own
 * (oracle/orc_effects.h).
 *
 * Restated behaviour, including what looks accidental in the reference:
 *   - Eigen unrolls fixed-size reductions (dot, squaredNorm, matrix * vector) as first half + second
 *     half, i.e. t0 + (t1 + t2) for three terms (Eigen/src/Core/Redux.h, redux_novec_unroller);
 *     normalized() is v / sqrt(squaredNorm) and leaves an all-zero vector alone (Eigen >= 3.3);
 *   - float -> unsigned / uint8 / int32 conversions saturate and map NaN to 0 (CUDA cvt.rzi);
 *   - `occlusion = 1.0 - (occlusion / rows)` is evaluated in double and stored back into the float
 *     variable; `occlusion * 255` is then a float product (src/effects.cu:84-85);
 *   - powf(t, 2.0f) is t * t;
 *   - blur_ssao's second pass indexes the window at (xmin + i, ymin + j) WITHOUT the pixel's own
 *     (x, y) (src/effects.cu:124-126), and divides by count - 1 (NaN or inf for windows with
 *     fewer than two covered pixels, which then never win the `stdev < best` comparison);
 *   - the bounds tests of draw_ssao / draw_shaded use && (src/effects.cu:32,171); the kernels
 *     here are launched over exactly S x S pixels, where that makes no difference.
 */
#ifndef MPR_EFFECTS_MATH_H
#define MPR_EFFECTS_MATH_H

#include "mpr_fmath.h"

MPR_HD uint32_t mpr_fx_f2u(float f)
{
    if (!(f > 0.0f)) return 0u;                 /* NaN, negatives, zero */
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}
MPR_HD int32_t mpr_fx_f2i(float f)
{
    if (mpr_isnanf(f)) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (int32_t)(-2147483647 - 1);
    return (int32_t)f;
}
MPR_HD uint32_t mpr_fx_d2u8(double d)
{
    if (!(d > 0.0)) return 0u;
    if (d >= 255.0) return 255u;
    return (uint32_t)d;
}
MPR_HD float mpr_fx_sum3(float t0, float t1, float t2) { return t0 + (t1 + t2); }
MPR_HD void mpr_fx_normalize(float v[3])
{
    const float zz = mpr_fx_sum3(v[0] * v[0], v[1] * v[1], v[2] * v[2]);
    if (!(zz > 0.0f)) return;
    const float n = sqrtf(zz);
    v[0] = v[0] / n;
    v[1] = v[1] / n;
    v[2] = v[2] / n;
}
MPR_HD void mpr_fx_normal_of(uint32_t n, float out[3])
{
    out[0] = (float)(n & 0xFF) - 128.0f;
    out[1] = (float)((n >> 8) & 0xFF) - 128.0f;
    out[2] = (float)((n >> 16) & 0xFF) - 128.0f;
    mpr_fx_normalize(out);
}

/* draw_ssao, src/effects.cu:17-87; returns -1 for "pixel left untouched" */
MPR_HD int32_t mpr_fx_ssao_pixel(const int32_t* depth, const uint32_t* norm, const float* kernel /* [64][3] */,
                                 const float* rvecs /* [256][3] */, int S, int x, int y)
{
    const float RADIUS = 0.1f;
    const int h = depth[x + y * S];
    if (!h) return -1;
    const float pos[3] = {2.0f * ((x + 0.5f) / S - 0.5f), 2.0f * ((y + 0.5f) / S - 0.5f), 2.0f * ((h + 0.5f) / S - 0.5f)};
    float normal[3];
    mpr_fx_normal_of(norm[x + y * S], normal);
    const float* rvec = rvecs + 3 * ((x % 16) * 16 + (y % 16));
    const float rn = mpr_fx_sum3(rvec[0] * normal[0], rvec[1] * normal[1], rvec[2] * normal[2]);
    float tangent[3] = {rvec[0] - normal[0] * rn, rvec[1] - normal[1] * rn, rvec[2] - normal[2] * rn};
    mpr_fx_normalize(tangent);
    const float bitangent[3] = {normal[1] * tangent[2] - normal[2] * tangent[1],
                                normal[2] * tangent[0] - normal[0] * tangent[2],
                                normal[0] * tangent[1] - normal[1] * tangent[0]};
    float occlusion = 0.0f;
    for (int i = 0; i < 64; ++i) {
        const float* k = kernel + 3 * i;
        float sp[3];
        for (int c = 0; c < 3; ++c)
            sp[c] = mpr_fx_sum3(tangent[c] * k[0], bitangent[c] * k[1], normal[c] * k[2]) * RADIUS + pos[c];
        const uint32_t px = mpr_fx_f2u((sp[0] / 2.0f + 0.5f) * S);
        const uint32_t py = mpr_fx_f2u((sp[1] / 2.0f + 0.5f) * S);
        const uint32_t actual_h = (px < (uint32_t)S && py < (uint32_t)S) ? (uint32_t)depth[px + py * (uint32_t)S] : 0u;
        const float actual_z = 2.0f * ((actual_h + 0.5f) / S - 0.5f);
        const float dz = fabsf(sp[2] - actual_z);
        if (dz < RADIUS) {
            occlusion += (sp[2] <= actual_z) ? 1.0f : 0.0f;
        } else if (dz < RADIUS * 2.0f) {
            if (sp[2] <= actual_z) {
                const float t = (RADIUS - (dz - RADIUS)) / RADIUS;
                occlusion += t * t;
            }
        }
    }
    const float occ = (float)(1.0 - (double)(occlusion / 64));
    return (int32_t)mpr_fx_d2u8((double)(occ * 255));
}

/* blur_ssao, src/effects.cu:91-152 */
MPR_HD int32_t mpr_fx_blur_pixel(const int32_t* image, const int32_t* ssao, int S, int x, int y)
{
    const int R = 2;
    float best = 1000000.0f, value = 0.0f;
    for (unsigned q = 0; q < 4; ++q) {
        const int xmin = (q & 1) ? 0 : -R, ymin = (q & 2) ? 0 : -R;
        float sum = 0.0f, count = 0.0f;
        for (int i = 0; i <= R; ++i)
            for (int j = 0; j <= R; ++j) {
                const int tx = x + xmin + i, ty = y + ymin + j;
                if (tx >= 0 && tx < S && ty >= 0 && ty < S && image[tx + ty * S]) {
                    sum += (float)ssao[tx + ty * S];
                    count += 1.0f;
                }
            }
        const float mean = sum / count;
        float stdev = 0.0f;
        for (int i = 0; i <= R; ++i)
            for (int j = 0; j <= R; ++j) {
                const int tx = xmin + i, ty = ymin + j;          /* sic: not offset by (x, y) */
                if (tx >= 0 && tx < S && ty >= 0 && ty < S && image[tx + ty * S]) {
                    const float d = mean - (float)ssao[tx + ty * S];
                    stdev += d * d;
                }
            }
        stdev = stdev / (count - 1.0f);
        stdev = sqrtf(stdev);
        if (stdev < best) {
            best = stdev;
            value = mean;
        }
    }
    return mpr_fx_f2i(value);
}

/* draw_shaded, src/effects.cu:156-209; returns 0 for "pixel left untouched" (alpha is never 0 otherwise) */
MPR_HD uint32_t mpr_fx_shade_pixel(const int32_t* depth, const uint32_t* norm, const int32_t* ssao, int S, int x, int y)
{
    const int h = depth[x + y * S];
    if (!h) return 0u;
    const uint32_t s = (uint32_t)ssao[x + y * S] & 0xFFu;          /* const uint8_t s = ssao[...] */
    float normal[3];
    mpr_fx_normal_of(norm[x + y * S], normal);
    const float pos[3] = {2.0f * ((x + 0.5f) / S - 0.5f), 2.0f * ((y + 0.5f) / S - 0.5f), 2.0f * ((h + 0.5f) / S - 0.5f)};
    float ld[3] = {5.0f - pos[0], 5.0f - pos[1], 10.0f - pos[2]};
    mpr_fx_normalize(ld);
    float light = mpr_fmaxf(0.0f, mpr_fx_sum3(ld[0] * normal[0], ld[1] * normal[1], ld[2] * normal[2])) * 0.8f;
    light *= (float)s / 255.0f;
    light += 0.2f;
    if (light < 0.0f) light = 0.0f;
    else if (light > 1.0f) light = 1.0f;
    const uint32_t color = mpr_fx_d2u8((double)(light * 255.0f));
    return (0xFFu << 24) | (color << 16) | (color << 8) | color;
}

#endif
