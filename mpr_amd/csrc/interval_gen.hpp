/*
 * interval_gen.hpp — the ROOT tape's interval forward walk (reference src/context.cu:188-321, inc/gpu_interval.hpp:71-391) as
 * scheduled gfx950 code: every clause's routine in line on the registers its operands live in, independent clauses interleaved.
 *
 * Round 4's generated walk (tile_gen.cpp) was a dependent chain that called the interpreter's routines: half of its in-line
 * instructions were v_movs into and out of the routines' fixed registers, and a lone wavefront had nothing to issue under each
 * dependent result (a tile stage runs four wavefronts per SIMD; the DAG of a tape has 5-8 independent clauses per level).  Here
 * a clause is expanded to instructions over VIRTUAL registers (gfx950_ir.hpp) — slots are renamed away: a value is a pair of
 * virtual registers, COPY and NEG cost nothing —, the instructions of neighbouring clauses are list-scheduled by their
 * dependencies (a window of clauses bounds the register pressure), a linear scan hands out physical registers, and a last pass
 * inserts the wait states the chip does not interlock (VALU-written SGPR -> VALU: 2; transcendental -> other VALU: 1).
 *
 * Two kinds of arithmetic:
 *  - EXACT: the reference's enclosures bit for bit — the instruction sequences of the interpreter's routines
 *    (tile_interp_asm.hpp: TI_BODIES_TEXT; tile_gen_asm.hpp: L_gmin / L_gmax) on renamed registers; sqrt / div / exp / log /
 *    asin / acos / atan stay calls (operands v[36:39], result v[40:41], entry points in the SGPR pairs of TileGenReg).
 *  - LOOSE (frames nobody reads; tile_gen_asm.hpp explains what such a frame owes the reference): sound, slightly wider
 *    enclosures on NEGATED lower bounds (a value is (-lo, hi), so that both ends round up and sums, differences and products by
 *    constants take two instructions), products by the four-products rule, exp / log / sqrt / reciprocals from the hardware's
 *    base-2 instructions widened by their error bound (two units in the last place assumed, checked on every float:
 *    tests/test_gpu_primitives.py: test_loose_interval_code_on_every_float), constants' reciprocals rounded on the host.
 *    Infinite ends are ordinary (an exp that overflows has the upper end +inf, a quotient by an interval that holds zero is
 *    [-inf, inf]).  The loose walk is straight-line code: instead of testing each operand where it is used it keeps two sticky
 *    flags — the lanes in which an operand left its routine's domain (s[40:41]: a negative radicand, a logarithm's lower end that
 *    is not a positive normal number, 0 x inf in a product) and a NaN accumulator (v42..v45: the sum of the widths hi - lo of the
 *    axes and of every product / square / quotient / exponential; a NaN end anywhere makes it a NaN, and inf - inf does) — and
 *    when either is raised at the end the WHOLE walk runs again on the exact code (s[60:61]: the harness's redo entry; 0.1 % of
 *    bear's wavefronts).  vgpr_limit = IGEN_LEAN_VGPRS: the code names v0..v63 only (tile_gen_asm.hpp: tile_gen_forward2_lean).
 *
 * Kinds of walk: FIRST (nobody above decided anything), BELOW (a stage below the first: the parent tile's decisions
 * s[72:73] / s[74:75] are imposed on the min / max clauses), BELOW_GUARDED (also jumps over the runs of clauses those decisions
 * leave dead — voxel_gen.hpp: tape_dead_runs; for stages that push no tapes).
 *
 * FIRST_MASKS (loose only; report_only implied: the code returns with the lanes that ask for the exact walk in s[40:41]): a first stage's
 * walk for tapes beyond 24 slots / 64 min / max clauses — architecture has 93 and 488 — in the kernel whose interpreter keeps 93
 * slots in registers (tile_interp_asm.hpp: tile_interp_asm_vgpr): values live in v0..v253 (linear scan: a tape whose live values
 * do not fit is not taken), and a choice is recorded the way that interpreter's forward walk records it, for ITS backward walk to
 * read: 16 bytes in LDS per min / max clause, the lanes that chose the lhs and the lanes that chose the rhs (v_writelane_b32 of the
 * four halves into lane k & 63 of v56..v59, one ds_write_b128 per 64 choices at v60 + 1024 (k >> 6): v60 = the lane's entry of the
 * first group); s[78:79] = the lanes that decided anything.
 *
 * Register conventions of the code (the harness: tile_gen_asm.hpp: tile_gen_forward2)
 *   in:  v0..v5 = x.lo, x.hi, y.lo, y.hi, z.lo, z.hi; s[72:73] / s[74:75] decided above; v56..v59 = 0; round-up mode, all lanes on
 *   out: v[36:37] = the end clause's interval; v56 / v57 (v58 / v59): bit k = the lane chose the lhs (rhs) at min / max clause k
 *   returns through s[38:39]; loose code leaves through s[60:61] when the walk has to be redone
 */
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace mpr {

enum IntervalWalkKind : int { IW_FIRST = 0, IW_BELOW = 1, IW_BELOW_GUARDED = 2, IW_FIRST_MASKS = 3 };

struct IntervalCode {
    bool ok = false;
    std::vector<uint32_t> words;
    std::vector<std::string> text;      /* one line per instruction (assembler syntax; kept when asked for: tests) */
    int instructions = 0, nops = 0, window = 0;
    int max_vgprs = 0, max_sgpr_pairs = 0;
    int nchoices = 0, walk_words = 0, result_slot = 0;
    int est_cycles = 0;                 /* the scheduler's own estimate for a lone wavefront */
    bool tight = false;                 /* the code also leaves the second, tight result (v[38:39]) */
};

constexpr int IGEN_MAX_CHOICES = 64;
constexpr int IGEN_MAX_CHOICES_MASKS = 4096;    /* (the store's 16-bit offset: 64 groups of 64 entries of 16 bytes) */
/* loose walks name vector registers below this only (interval_gen_build: vgpr_limit): the tile stages run them in wavefronts of 80
 * registers, six to a SIMD instead of four (tile_gen_asm.hpp: tile_gen_forward2_lean) */
constexpr int IGEN_LEAN_VGPRS = 64;
/* ... tight code below this (it carries the values that depend on a sin / cos twice): 96 registers, five to a SIMD */
constexpr int IGEN_TIGHT_VGPRS = 80;

/* clauses: head, operations, end (the host copy of a root tape).  loose: see above (false for tapes with asin / acos / atan
 * clauses or a constant divisor outside 2^-100 .. 2^100: ok == false).  window: clauses the scheduler may look ahead (0: the
 * default, shrunk until the registers suffice; 1: the tape's own order).  min_run: shortest dead run worth a guard.
 * report_only (loose, tests): no branch to the redo entry — the code returns with the lanes that ask for the exact walk in s[40:41]. */
/* tight (loose code of the kinds FIRST / BELOW / BELOW_GUARDED; ok == false for a tape without sin / cos: nothing to tighten): the
 * code also leaves, in v[38:39], a SECOND enclosure of the result, from sin / cos enclosed by their monotone pieces where the
 * reference — and everything above — has [-1, 1] whatever the argument (inc/gpu_interval.hpp:353; the range reduction behind it,
 * :355-375, is dead code).  Everything the walk leaves otherwise — v[36:37], the decisions, the lanes that ask for the exact walk —
 * is what the code without it leaves: the clauses that depend on a sin / cos are computed twice, on the wide and on the tight
 * values of their operands.  The tight values decide nothing and are recorded nowhere; what was decided ABOVE is imposed on them as
 * on the wide ones.  A lane whose tight values met a NaN or left a routine's domain gets [-inf, inf].  What the second result is for:
 * kernels.hip: k_eval_tiles<.., LEAN> — a smallest tile it proves empty or filled skips the float pass. */
IntervalCode interval_gen_build(const uint64_t* clauses, int len, int kind, bool loose, int window = 0, int min_run = 3, bool keep_text = false, int vgpr_limit = 0,
                                bool report_only = false, bool tight = false);

}  // namespace mpr
