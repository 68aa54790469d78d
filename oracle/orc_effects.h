/*
 * orc_effects.h — the ORACLE's own restatement of mpr::Effects (TEST INFRASTRUCTURE).
 *
 * Follows the reference's src/effects.cu kernel by kernel — draw_ssao :17-87, blur_ssao :91-152,
 * draw_shaded :156-209, the table construction of Effects::Effects :213-236 and the pass order of
 * drawSSAO :246-263 / drawShaded :265-286 — written independently of the product's
 * include/mpr_effects_math.h and include/mpr_effects_tables.h (nothing is included from there).
 *
 * What had to be decided because Eigen and nvcc are not available to ask:
 *   - Eigen's fixed-size reductions (dot, squaredNorm, matrix * vector) are unrolled as
 *     redux(first half) + redux(second half); for three terms that is  t0 + (t1 + t2);
 *   - normalized() divides by sqrt(squaredNorm) and leaves an all-zero vector alone (Eigen >= 3.3);
 *   - cross() and every other expression: no fused multiply-add;
 *   - float -> unsigned / uint8_t / int32_t conversions as CUDA's cvt.rzi: truncate, saturate, NaN -> 0;
 *   - `1.0 - (occlusion / rows)` and `occlusion * 255` are double arithmetic (the literal 1.0);
 *   - powf(t, 2.0f) = t * t;
 *   - rand() without srand() is glibc's srand(1) sequence; here it comes from the C library itself
 *     (random_r over private state: the generator behind rand(), without touching the process's).
 */
#ifndef ORC_EFFECTS_H
#define ORC_EFFECTS_H

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z; } orc_v3;

static inline float orc_sum3(float t0, float t1, float t2) { return t0 + (t1 + t2); }
static inline float orc_dot(orc_v3 a, orc_v3 b) { return orc_sum3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline orc_v3 orc_scale(orc_v3 a, float s) { orc_v3 r = {a.x * s, a.y * s, a.z * s}; return r; }
static inline orc_v3 orc_sub(orc_v3 a, orc_v3 b) { orc_v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static inline orc_v3 orc_add(orc_v3 a, orc_v3 b) { orc_v3 r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }
static inline orc_v3 orc_cross(orc_v3 a, orc_v3 b)
{
    orc_v3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
    return r;
}
static inline orc_v3 orc_normalized(orc_v3 a)
{
    const float zz = orc_dot(a, a);
    if (!(zz > 0.0f)) return a;
    const float n = sqrtf(zz);
    orc_v3 r = {a.x / n, a.y / n, a.z / n};
    return r;
}
static inline uint32_t orc_cvt_u32(float f)
{
    if (f != f || f <= 0.0f) return 0;
    return f >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)f;
}
static inline uint32_t orc_cvt_u8(double d)
{
    if (d != d || d <= 0.0) return 0;
    return d >= 255.0 ? 255u : (uint32_t)d;
}
static inline int32_t orc_cvt_i32(float f)
{
    if (f != f) return 0;
    if (f >= 2147483648.0f) return INT32_MAX;
    if (f <= -2147483648.0f) return INT32_MIN;
    return (int32_t)f;
}
static inline orc_v3 orc_unpack_normal(uint32_t n)       /* src/effects.cu:49-53, :180-184 */
{
    orc_v3 v = {(float)(n & 0xFF) - 128.0f, (float)((n >> 8) & 0xFF) - 128.0f, (float)((n >> 16) & 0xFF) - 128.0f};
    return orc_normalized(v);
}
static inline orc_v3 orc_pixel_pos(int x, int y, int h, int S)   /* :41-44, :187-190 */
{
    orc_v3 p = {2.0f * ((x + 0.5f) / S - 0.5f), 2.0f * ((y + 0.5f) / S - 0.5f), 2.0f * ((h + 0.5f) / S - 0.5f)};
    return p;
}

/* Effects::Effects(), src/effects.cu:213-236.  kernel: 64 rows, rvecs: 256 rows of (x, y, z). */
static void orc_fx_tables(orc_v3* kernel, orc_v3* rvecs)
{
    struct random_data rd;
    char state[128];
    int32_t r;
    memset(&rd, 0, sizeof rd);
    memset(state, 0, sizeof state);
    initstate_r(1u, state, sizeof state, &rd);        /* == srand(1): what an unseeded rand() starts from */
#define ORC_RAND01() (random_r(&rd, &r), (float)r / (float)RAND_MAX)
    for (unsigned i = 0; i < 64; ++i) {
        orc_v3 v;
        v.x = 2.0f * (ORC_RAND01() - 0.5f);
        v.y = 2.0f * (ORC_RAND01() - 0.5f);
        v.z = ORC_RAND01();
        const float n = sqrtf(orc_dot(v, v));
        v.x /= n; v.y /= n; v.z /= n;
        float scale = (float)i / (float)(64 - 1);
        scale = (scale * scale) * 0.9f + 0.1f;
        kernel[i] = orc_scale(v, scale);
    }
    for (unsigned i = 0; i < 256; ++i) {
        orc_v3 v;
        v.x = 2.0f * (ORC_RAND01() - 0.5f);
        v.y = 2.0f * (ORC_RAND01() - 0.5f);
        v.z = 0.0f;
        const float n = sqrtf(orc_dot(v, v));
        v.x /= n; v.y /= n; v.z /= n;
        rvecs[i] = v;
    }
#undef ORC_RAND01
}

/* draw_ssao over the whole image; pixels without a surface are left as they are */
static void orc_fx_draw_ssao(const int32_t* depth, const uint32_t* norm, const orc_v3* kernel, const orc_v3* rvecs,
                             int S, int32_t* out)
{
    const float RADIUS = 0.1f;
    for (int y = 0; y < S; ++y) {
        for (int x = 0; x < S; ++x) {
            const int h = depth[x + y * S];
            if (!h) continue;
            const orc_v3 pos = orc_pixel_pos(x, y, h, S);
            const orc_v3 normal = orc_unpack_normal(norm[x + y * S]);
            const orc_v3 rvec = rvecs[(x % 16) * 16 + (y % 16)];      /* threadIdx = pixel mod 16 (16x16 blocks) */
            const orc_v3 tangent = orc_normalized(orc_sub(rvec, orc_scale(normal, orc_dot(rvec, normal))));
            const orc_v3 bitangent = orc_cross(normal, tangent);
            /* rows of tbn = [tangent | bitangent | normal] */
            const orc_v3 row0 = {tangent.x, bitangent.x, normal.x};
            const orc_v3 row1 = {tangent.y, bitangent.y, normal.y};
            const orc_v3 row2 = {tangent.z, bitangent.z, normal.z};
            float occlusion = 0.0f;
            for (unsigned i = 0; i < 64; ++i) {
                const orc_v3 k = kernel[i];
                const orc_v3 rotated = {orc_dot(row0, k), orc_dot(row1, k), orc_dot(row2, k)};
                const orc_v3 sp = orc_add(orc_scale(rotated, RADIUS), pos);
                const uint32_t px = orc_cvt_u32((sp.x / 2.0f + 0.5f) * S);
                const uint32_t py = orc_cvt_u32((sp.y / 2.0f + 0.5f) * S);
                const uint32_t actual_h = (px < (uint32_t)S && py < (uint32_t)S) ? (uint32_t)depth[px + py * (uint32_t)S] : 0u;
                const float actual_z = 2.0f * ((actual_h + 0.5f) / S - 0.5f);
                const float dz = fabsf(sp.z - actual_z);
                if (dz < RADIUS) {
                    occlusion += (sp.z <= actual_z);
                } else if (dz < RADIUS * 2.0f) {
                    if (sp.z <= actual_z) {
                        const float t = (RADIUS - (dz - RADIUS)) / RADIUS;
                        occlusion += t * t;
                    }
                }
            }
            const float o = (float)(1.0 - (double)(occlusion / 64));     /* float occlusion = 1.0 - (...) */
            out[x + y * S] = (int32_t)orc_cvt_u8((double)(o * 255));      /* const uint8_t o = occlusion * 255 */
        }
    }
}

/* blur_ssao: of the four 3x3 windows touching the pixel, the mean of the one with the smallest deviation */
static void orc_fx_blur_ssao(const int32_t* image, const int32_t* ssao, int S, int32_t* out)
{
    const int R = 2;
    for (int y = 0; y < S; ++y) {
        for (int x = 0; x < S; ++x) {
            float best = 1000000.0f, value = 0.0f;
            for (unsigned w = 0; w < 4; ++w) {
                const int xmin = (w & 1) ? 0 : -R, ymin = (w & 2) ? 0 : -R;
                float sum = 0.0f, count = 0.0f;
                for (int i = 0; i <= R; ++i) {
                    for (int j = 0; j <= R; ++j) {
                        const int tx = x + xmin + i, ty = y + ymin + j;
                        if (tx < 0 || tx >= S || ty < 0 || ty >= S || !image[tx + ty * S]) continue;
                        sum += ssao[tx + ty * S];
                        count++;
                    }
                }
                const float mean = sum / count;
                float stdev = 0.0f;
                for (int i = 0; i <= R; ++i) {
                    for (int j = 0; j <= R; ++j) {
                        const int tx = xmin + i, ty = ymin + j;             /* src/effects.cu:124-125: no x, y here */
                        if (tx < 0 || tx >= S || ty < 0 || ty >= S || !image[tx + ty * S]) continue;
                        const float d = mean - ssao[tx + ty * S];
                        stdev += d * d;
                    }
                }
                stdev /= count - 1.0f;
                stdev = sqrtf(stdev);
                if (stdev < best) {
                    best = stdev;
                    value = mean;
                }
            }
            out[x + y * S] = orc_cvt_i32(value);
        }
    }
}

static void orc_fx_draw_shaded(const int32_t* depth, const uint32_t* norm, const int32_t* ssao, int S, int32_t* out)
{
    for (int y = 0; y < S; ++y) {
        for (int x = 0; x < S; ++x) {
            const int h = depth[x + y * S];
            if (!h) continue;
            const uint8_t s = (uint8_t)ssao[x + y * S];
            const orc_v3 normal = orc_unpack_normal(norm[x + y * S]);
            const orc_v3 pos = orc_pixel_pos(x, y, h, S);
            const orc_v3 light_pos = {5.0f, 5.0f, 10.0f};
            const orc_v3 light_dir = orc_normalized(orc_sub(light_pos, pos));
            float light = fmaxf(0.0f, orc_dot(light_dir, normal)) * 0.8f;
            light *= s / 255.0f;
            light += 0.2f;
            if (light < 0.0f) light = 0.0f;
            else if (light > 1.0f) light = 1.0f;
            const uint32_t color = orc_cvt_u8((double)(light * 255.0f));
            out[x + y * S] = (int32_t)((0xFFu << 24) | (color << 16) | (color << 8) | color);
        }
    }
}

#endif
