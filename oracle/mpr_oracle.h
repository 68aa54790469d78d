/*
 * mpr_oracle.h — interface of the CPU oracle (TEST INFRASTRUCTURE, not product code).
 * See mpr_oracle.c for what it restates and how it is pinned.
 */
#ifndef MPR_ORACLE_H
#define MPR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#include "../include/mpr_clause.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_frame orc_frame;

typedef struct orc_counters {
    int64_t tiles_in[3];
    int64_t tiles_empty[3];
    int64_t tiles_filled[3];
    int64_t tiles_masked[3];     /* masked before, during or after evaluation (3-D) */
    int64_t tiles_active[3];     /* survivors handed to the next stage */
    int64_t tiles_pushed[3];     /* tiles that wrote a shortened tape */
    int64_t voxel_tiles;
    int64_t clauses_fwd;         /* F: words fetched forward per 64-tile group / per voxel tile /
                                       per (8x8 patch, tape) group of the normals pass */
    int64_t clauses_fwd_tiles;   /*    of which tile stages */
    int64_t clauses_fwd_voxels;  /*    of which float voxel/pixel pass */
    int64_t clauses_fwd_normals; /*    of which normals pass */
    int64_t clauses_bwd;         /* R */
    int64_t clauses_written;     /* W */
    int64_t lane_clauses;        /* words visited forward, terminator excluded, per tile / voxel / pixel */
    int64_t normal_pixels;
    int32_t tape_index;
    int32_t pool_overflowed;
    int32_t slots_exceeded;
    int32_t threads;
    /* 3-D: the float pass's F over all execution orders lies between these (the skip test of src/context.cu:852-864
     * reads a heightmap other tiles are still writing): the walks of the smallest tiles with a voxel pair the FINAL
     * heightmap still leaves visible, and the walks of all of them; 2-D: both equal clauses_fwd_voxels */
    int64_t clauses_fwd_voxels_min, clauses_fwd_voxels_max;
    /* W restricted to the tiles that survive their stage (3-D: whether a tile that is culled later pushed first is
     * timing dependent, src/context.cu:299-305 vs :312) */
    int64_t clauses_written_survivors;
} orc_counters;

/* Render a frame.  dim = 2 or 3.  mat: column-major 3x3 (dim 2) or 4x4 (dim 3).
 * pool_clauses: capacity of the tape pool (0 = MPR_NUM_SUBTAPES_BIG*64).
 * threads: OpenMP threads (<=0: all).  owner/rank: optional column partition (NULL = all).
 * flags: bit0 = brute force (render2D_brute, dim 2 only), bit1 = skip the normals pass,
 *        bit2 = heatmap frame (Context::render2D_heatmap / render3D_heatmap, src/context.cu:1984-2339):
 *        also accumulate the amortised work per pixel, read back with orc_heatmap;
 *        bit3 / bit4 instead of bit2 = the lower / upper bound of that heatmap over the parts of a
 *        3-D frame that depend on timing upstream too (mid-stage culling of tiles by a neighbour's
 *        fill, the float pass's skip test) — any execution order lands between the two.  */
orc_frame* orc_render(const uint64_t* tape, int32_t length, int32_t dim, int32_t image_size_px,
                      const float* mat, float z, int64_t pool_clauses, int32_t threads,
                      const int32_t* owner, int32_t rank, int32_t flags);
void orc_frame_free(orc_frame* f);

const int32_t* orc_filled(const orc_frame* f, int32_t stage, size_t* n);
const uint32_t* orc_normals(const orc_frame* f, size_t* n);
const float* orc_heatmap(const orc_frame* f, size_t* n);      /* NULL unless flags bit2 was set */
const mpr_tile_node* orc_tiles(const orc_frame* f, int32_t stage, size_t* n);
const uint64_t* orc_tape_pool(const orc_frame* f, int32_t* tape_index);
void orc_get_counters(const orc_frame* f, orc_counters* out);

/* Walk the tape whose head is pool[head] (forward, following JUMPs) and return its number of
 * operation clauses and a 64-bit FNV-1a hash of them (JUMP/chunk layout stripped).
 * Works on any pool with the reference's sub-tape format (src/context.cu:340-458), so the
 * same function digests the oracle's pool and a pool read back from the GPU. */
int32_t orc_tape_digest(const uint64_t* pool, int64_t pool_len, int32_t head, uint64_t* hash);
/* Digest many tiles at once: for i < n, tile i's tape -> len[i], hash[i]. */
void orc_tiles_digest(const uint64_t* pool, int64_t pool_len, const mpr_tile_node* tiles, size_t n,
                      int32_t* len, uint64_t* hash);

/* ---- primitives, exposed for unit tests and for GPU-vs-oracle fuzzing ---- */
/* interval op `op` (MPR_OP_*); returns choice (0/1/2) for min/max, else 0 */
int32_t orc_interval_op(int32_t op, float a_lo, float a_hi, float b_lo, float b_hi, float imm,
                        float* out_lo, float* out_hi);
void orc_interval_op_n(int32_t op, int32_t n, const float* a_lo, const float* a_hi,
                       const float* b_lo, const float* b_hi, float imm, float* out_lo,
                       float* out_hi, int32_t* choice);
float orc_float_op(int32_t op, float a, float b, float imm);
void orc_float_op_n(int32_t op, int32_t n, const float* a, const float* b, float imm, float* out);
/* derivative op on (dx,dy,dz,v) quadruples */
void orc_deriv_op_n(int32_t op, int32_t n, const float* a4, const float* b4, float imm, float* out4);
/* self-check of the "everything in round-up mode" formulation against direct
 * fesetround(FE_DOWNWARD / FE_UPWARD) evaluation on n random operand pairs; returns the
 * number of mismatches */
int64_t orc_selftest_rounding(int64_t n, uint64_t seed);
/* the oracle's float functions (oracle/orc_fmath.h) */
void orc_fmath_n(int32_t which, int32_t n, const float* x, float* out);

/* ---- mpr::Effects (reference src/effects.cu) over a finished frame's heightmap + normals ----
 * which: 0 = drawSSAO (image = blurred occlusion; tmp = raw occlusion), 1 = drawShaded (image =
 * shading; tmp = blurred occlusion).  image / tmp: size * size int32 each, zeroed first. */
void orc_effects(int32_t which, int32_t size, const int32_t* depth, const uint32_t* normals, int32_t* image, int32_t* tmp);
/* the SSAO tables (64 x 3 and 256 x 3 floats) and the first n draws of the restated glibc rand() */
void orc_effects_tables(float* kernel, float* rvecs);
void orc_glibc_rand(uint32_t seed, int32_t n, int32_t* out);

#ifdef __cplusplus
}
#endif
#endif
