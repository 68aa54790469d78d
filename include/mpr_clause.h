/*
 * mpr_clause.h — tape clause encoding and opcode set (data format of the hot path).
 *
 * Restates the *format* defined by the reference at
 *   inc/clause.hpp:18-23      (byte layout of a 64-bit clause)
 *   inc/gpu_opcode.hpp:18-56  (opcode numbering)
 *   inc/parameters.hpp:14-22  (chunk size / pool size)
 * Plain C so that the HIP kernels, the C++ host code and the C oracle all read the same
 * definition.  A clause is a little-endian u64:
 *
 *   bits  0..7   op        (0 terminates a tape; the *head* clause of a tape is also op 0 and
 *                           carries the X/Y/Z input slots in bytes 1..3; the *end* clause
 *                           carries the result slot in byte 1)
 *   bits  8..15  out slot
 *   bits 16..23  lhs slot  (0 = "no operand")
 *   bits 24..31  rhs slot  (0 = "no operand")
 *   bits 32..63  f32 immediate, or i32 relative jump target for MPR_OP_JUMP
 */
#ifndef MPR_CLAUSE_H
#define MPR_CLAUSE_H

#include <stdint.h>

#if defined(__HIPCC__)
#define MPR_CL_FN __host__ __device__ static inline
#else
#define MPR_CL_FN static inline
#endif

#ifdef __cplusplus
extern "C" {
#endif

enum mpr_opcode {
    MPR_OP_INVALID = 0,
    MPR_OP_JUMP = 1,

    MPR_OP_SQUARE_LHS = 2,
    MPR_OP_SQRT_LHS = 3,
    MPR_OP_NEG_LHS = 4,
    MPR_OP_SIN_LHS = 5,
    MPR_OP_COS_LHS = 6,
    MPR_OP_ASIN_LHS = 7,
    MPR_OP_ACOS_LHS = 8,
    MPR_OP_ATAN_LHS = 9,
    MPR_OP_EXP_LHS = 10,
    MPR_OP_ABS_LHS = 11,
    MPR_OP_LOG_LHS = 12,

    /* commutative: the non-constant operand is always lhs */
    MPR_OP_ADD_LHS_IMM = 13,
    MPR_OP_ADD_LHS_RHS = 14,
    MPR_OP_MUL_LHS_IMM = 15,
    MPR_OP_MUL_LHS_RHS = 16,
    MPR_OP_MIN_LHS_IMM = 17,
    MPR_OP_MIN_LHS_RHS = 18,
    MPR_OP_MAX_LHS_IMM = 19,
    MPR_OP_MAX_LHS_RHS = 20,

    /* non-commutative */
    MPR_OP_SUB_LHS_IMM = 21,
    MPR_OP_SUB_IMM_RHS = 22,
    MPR_OP_SUB_LHS_RHS = 23,
    MPR_OP_DIV_LHS_IMM = 24,
    MPR_OP_DIV_IMM_RHS = 25,
    MPR_OP_DIV_LHS_RHS = 26,

    /* produced only by tape shortening */
    MPR_OP_COPY_IMM = 27,
    MPR_OP_COPY_LHS = 28,
    MPR_OP_COPY_RHS = 29,

    MPR_OP_COUNT = 30
};

/* inc/parameters.hpp:16 — sub-tapes are linked lists of 64-clause chunks */
#define MPR_SUBTAPE_CHUNK 64
/* inc/parameters.hpp:18-22 — pool size in chunks (default build / BIG_SERVER build) */
#define MPR_NUM_SUBTAPES_DEFAULT 640000
#define MPR_NUM_SUBTAPES_BIG 6400000
/* src/context.cu:210,866,1007 — the kernels' slot files hold 128 entries */
#define MPR_KERNEL_SLOTS 128
/* src/context.cu:215,257 — at most 256*16 min/max choices are recorded per tile */
#define MPR_MAX_CHOICES 4096

MPR_CL_FN uint32_t mpr_cl_op(uint64_t c)  { return (uint32_t)(c & 0xFF); }
MPR_CL_FN uint32_t mpr_cl_out(uint64_t c) { return (uint32_t)((c >> 8) & 0xFF); }
MPR_CL_FN uint32_t mpr_cl_lhs(uint64_t c) { return (uint32_t)((c >> 16) & 0xFF); }
MPR_CL_FN uint32_t mpr_cl_rhs(uint64_t c) { return (uint32_t)((c >> 24) & 0xFF); }
MPR_CL_FN uint32_t mpr_cl_immbits(uint64_t c) { return (uint32_t)(c >> 32); }
MPR_CL_FN int32_t mpr_cl_jump(uint64_t c) { return (int32_t)(uint32_t)(c >> 32); }
MPR_CL_FN uint64_t mpr_cl_make(uint32_t op, uint32_t out, uint32_t lhs, uint32_t rhs,
                                   uint32_t immbits)
{
    return (uint64_t)(op & 0xFF) | ((uint64_t)(out & 0xFF) << 8) | ((uint64_t)(lhs & 0xFF) << 16) |
           ((uint64_t)(rhs & 0xFF) << 24) | ((uint64_t)immbits << 32);
}
MPR_CL_FN int mpr_op_is_minmax(uint32_t op)
{   /* src/context.cu:365-366: tested by range */
    return op >= MPR_OP_MIN_LHS_IMM && op <= MPR_OP_MAX_LHS_RHS;
}

/* src/gpu_opcode.cu:17-58 */
const char* mpr_op_str(uint8_t op);

/* (position, tape, next) — inc/context.hpp:23-27 */
typedef struct mpr_tile_node {
    int32_t position;   /* linear tile index at its level, or -1 when culled / filled / empty */
    int32_t tape;       /* index of the tile's head clause in the tape pool */
    int32_t next;       /* compacted id among the active tiles of its level, or -1 */
} mpr_tile_node;

#ifdef __cplusplus
}
#endif
#endif
