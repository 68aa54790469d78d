/*
 * voxel_gen.hpp — the ROOT tape's float walk (reference src/context.cu:828-964, eval_voxels_f) as gfx950 machine code,
 * generated on the host when a tape is made resident (next to tile_gen.hpp's interval and Deriv walks).
 *
 * The tape a smallest tile walks in the reference is the root tape with decisions applied: those of the 16^3 tile above it
 * and its own (min / max clauses turned into copies, clauses nothing reads any more dropped).  With both sets of decisions
 * at hand as bits over the ROOT tape's min / max clauses (the tiles' records, tile_gen.hpp) every smallest tile can run the
 * one piece of code that exists once per tape — no translation on the device, no per-group code, no instruction-cache
 * invalidates — and since that code is written once, on the host, it can be written well:
 *   - a clause is its instructions on the clause's OWN slot registers: square root, exp, log and the division by a
 *     constant are emitted in line (the interpreters' instruction sequences, asm_float_bodies.hpp, with the operand and
 *     result registers renamed: same operations on the same values in the same order, hence the same bits) — no operand
 *     moves, no s_swappc / s_setpc pair; the rare operands (special values, tiny or huge arguments) leave through a stub
 *     behind the code that calls the interpreters' full routine;
 *   - a decided min / max is a scalar branch to a stub that copies the chosen operand (the decisions of the tile being
 *     evaluated are wave-uniform: one tile per wavefront at a time), an undecided one is one v_min / v_max;
 *   - DEAD CODE IS JUMPED OVER.  For every (min / max clause m, operand side) the clauses that only that operand of m
 *     reaches — its exclusive sub-DAG — die when m is decided for the other side; where they form a run of consecutive
 *     clauses in the tape (expression trees are written out depth first: they mostly do) the run starts with
 *     `s_bitcmp1_b64 <decided for the other side>, m; s_cbranch_scc1 <end of run>`.  For bear (544 clauses, 27 min / max:
 *     34 runs) a smallest tile then runs 451 clauses on average where its own shortened tape has 431
 *     (scripts/skip_study.py): what tape shortening buys, without tapes.
 *
 * Register conventions (the harness, kernels_voxel_gen.hip, and the routines are written against them)
 *   slot s                v[48 + s]                                   s <= 23
 *   inputs                v32, v33, v34 = x, y, z (copied into their slots by the code's first instructions)
 *   temporaries           v35 .. v47; v7 = 0x2ff (class mask of the inline constant division); v5 = 0x39506967, v6 = 0x3d9021bb
 *                         (the leading coefficients of the exp / log polynomials: p = c v + c' is ONE v_fmaak with c in a register,
 *                         where the routines, which may not keep registers between calls, spend a v_mov and a v_fmac)
 *   result                v37; the code returns through s[72:73]
 *   routines              argument v35 (, v36), result v37, return address s[30:31]; entry points in SGPR pairs (VoxelGenReg)
 *   decisions             s[76:77] / s[78:79]: bit k set = the tile decided min / max clause k of the root tape for the lhs / rhs
 *   constants             s80 = 87.0f (exp: |x| <= 87 has none of the special cases), s81 = 0x100 (class: positive normal)
 */
#pragma once
#include <cstdint>
#include <vector>

namespace mpr {

enum VoxelGenReg : int {
    VG_SLOT_BASE = 48, VG_MAX_SLOTS = 24, VG_MAX_CHOICES = 64,
    VG_RT_DIV = 52, VG_RT_SQRT = 54, VG_RT_EXP = 56, VG_RT_LOG = 58, VG_RT_SIN = 60, VG_RT_COS = 62,
    VG_RT_ASIN = 64, VG_RT_ACOS = 66, VG_RT_ATAN = 68,
    VG_RET = 72, VG_DEC_L = 76, VG_DEC_R = 78, VG_K_87 = 80, VG_K_POSNORMAL = 81,
    VG_K_EXP_C5 = 5, VG_K_LOG_C8 = 6,        /* VGPRs: the leading coefficients of the exp / log polynomials */
};

/* A run of consecutive clauses [first, last] of a tape that is dead when the tape's `choice`-th min / max clause is decided for
 * the lhs (by_lhs) / for the rhs: the clauses only the OTHER operand of that clause reaches (its exclusive sub-DAG).  Runs of
 * one clause pair nest or are disjoint.  end = index of the tape's end clause; runs shorter than min_run are left out. */
struct DeadRun { int first, last, choice; bool by_lhs; };
std::vector<DeadRun> tape_dead_runs(const uint64_t* clauses, int end, int min_run);

struct VoxelGen {
    bool ok = false;
    std::vector<uint32_t> code;
    int nchoices = 0;       /* min / max clauses of the tape */
    int runs = 0;           /* guarded runs of clauses */
    int stubs = 0;          /* out-of-line pieces behind the code */
};

/* clauses: head, operations, end (the host copy of a root tape).  ok == false: the tape does not fit the conventions above
 * (a slot beyond 23, more than 64 min / max clauses, a jump or an unknown opcode).  min_run: shortest run of dead clauses
 * worth a guard (0: no guards at all). */
VoxelGen voxel_gen_build(const uint64_t* clauses, int len, int min_run = 5);

}  // namespace mpr
