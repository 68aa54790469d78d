/*
 * mpr_effects_tables.h — the SSAO sample kernel and noise vectors of mpr::Effects
 * (reference src/effects.cu:213-236), shared by libmpr_amd and the oracle.
 *
 * The reference fills both tables from the C library's rand() without seeding it, i.e. with the
 * sequence of srand(1).  That sequence is glibc's TYPE_3 additive feedback generator
 * (r[i] = r[i-3] + r[i-31], 34 warm-up words seeded by the Lehmer step 16807 x mod 2^31 - 1,
 * 310 outputs discarded); it is restated here so that the tables do not depend on the state of the
 * host process's rand() — and are the reference's own on any glibc system.
 */
#ifndef MPR_EFFECTS_TABLES_H
#define MPR_EFFECTS_TABLES_H

#include <math.h>
#include <stdint.h>

typedef struct mpr_glibc_rand {
    int32_t r[34];
    int f, b;       /* front / back indices into r[3..33] (the 31-word state) */
} mpr_glibc_rand;

static inline void mpr_glibc_srand(mpr_glibc_rand* g, uint32_t seed)
{
    int32_t* const st = g->r + 3;        /* 31 words */
    int32_t word = seed ? (int32_t)seed : 1;
    st[0] = word;
    for (int i = 1; i < 31; ++i) {
        /* word = 16807 * word % 2147483647 without overflow (Schrage) */
        const long hi = word / 127773, lo = word % 127773;
        long w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        word = (int32_t)w;
        st[i] = word;
    }
    g->f = 3;
    g->b = 0;
    for (int i = 0; i < 310; ++i) {
        st[g->f] = (int32_t)((uint32_t)st[g->f] + (uint32_t)st[g->b]);
        g->f = (g->f + 1) % 31;
        g->b = (g->b + 1) % 31;
    }
}
static inline int32_t mpr_glibc_rand_next(mpr_glibc_rand* g)
{
    int32_t* const st = g->r + 3;
    st[g->f] = (int32_t)((uint32_t)st[g->f] + (uint32_t)st[g->b]);
    const int32_t result = (int32_t)(((uint32_t)st[g->f]) >> 1);
    g->f = (g->f + 1) % 31;
    g->b = (g->b + 1) % 31;
    return result;
}

/* kernel[64][3], rvecs[256][3]; row-major (row i = sample i).  src/effects.cu:213-236 */
static inline void mpr_effects_tables(float kernel[64 * 3], float rvecs[256 * 3])
{
    mpr_glibc_rand g;
    mpr_glibc_srand(&g, 1);
    const float rmax = (float)2147483647;        /* RAND_MAX */
    for (unsigned i = 0; i < 64; ++i) {
        float v[3];
        v[0] = 2.0f * ((float)mpr_glibc_rand_next(&g) / rmax - 0.5f);
        v[1] = 2.0f * ((float)mpr_glibc_rand_next(&g) / rmax - 0.5f);
        v[2] = (float)mpr_glibc_rand_next(&g) / rmax;
        const float n = sqrtf(v[0] * v[0] + (v[1] * v[1] + v[2] * v[2]));   /* Eigen's reduction order */
        float scale = (float)i / (float)(64 - 1);
        scale = (scale * scale) * 0.9f + 0.1f;
        for (int k = 0; k < 3; ++k) kernel[i * 3 + k] = (v[k] / n) * scale;
    }
    for (unsigned i = 0; i < 256; ++i) {
        float v[3];
        v[0] = 2.0f * ((float)mpr_glibc_rand_next(&g) / rmax - 0.5f);
        v[1] = 2.0f * ((float)mpr_glibc_rand_next(&g) / rmax - 0.5f);
        v[2] = 0.0f;
        const float n = sqrtf(v[0] * v[0] + (v[1] * v[1] + v[2] * v[2]));   /* Eigen's reduction order */
        for (int k = 0; k < 3; ++k) rvecs[i * 3 + k] = v[k] / n;
    }
}

#endif
