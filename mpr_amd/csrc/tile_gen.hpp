/*
 * tile_gen.hpp — the ROOT tape's walks as gfx950 machine code, generated on the host when a tape is made resident.
 *
 * Every tile of a frame's first stage walks the same tape (reference src/context.cu:188-321 forward, :323-458 backward), so
 * unlike the tapes pushed later this one is worth compiling: a clause becomes the handful of instructions its opcode needs,
 * with the slot registers and the immediate in the instruction words — no fetch, no decode, no dispatch, no operand moves
 * through a register index.  This file: the BACKWARD walks (tape pushing) and the Deriv walk of the normals pass; the forward
 * walk is interval_gen.hpp's (round 5: scheduled code on renamed registers; round 3 / 4's forward walk — one row per clause on
 * fixed slot registers, calling the interpreter's routines — is gone).  Counterpart on the device: tile_gen_asm.hpp.
 * The tapes pushed later are this tape with decisions applied; with a tile's decisions kept as bits over the root tape's
 * min / max clauses (its RECORD, below) the stages below the first and the normals pass (reference :978-1132; the Deriv walk,
 * run by kernels_normals_asm.hip: k_eval_normals_gen) run the same one piece of code per tape as well.
 *
 * Register conventions of the generated code (fixed: the harness and the routine bodies are written against them)
 *   (forward walk: interval_gen.hpp)
 *   decisions               v56 / v57: bit k set = this lane's tile chose the LHS at min / max clause k (k < 32 / k >= 32),
 *                           v58 / v59: the same for the RHS
 *   routine entry points    SGPR pairs (TileGenReg), return address s[36:37]; the code itself returns through s[38:39]
 * backward walk
 *   v60 active slots (bit s), v61 pool index of the last word written, v62 first index of the current chunk,
 *   v[46:47] the clause word being stored (v47 keeps the last upper half), s[76:77] pool, s[62:63] "chunk full" routine,
 *   v54 number of min / max clauses the lane's tape keeps, v41 / v42 which ones (bit k: clause k);
 *   the walk for tapes that are shortened again also: s[0..23] the clauses of the tape being shortened, s[64:65] / s[66:67] the
 *   min / max clauses that are copies on it (decided above for the lhs / rhs), v64..v87 the clauses of the tape being written
 * Deriv walk: slot s = v[50 + s], routine operands v35 (, v36), result v37, return address s[70:71]; s[98:99] the value lanes;
 *   v74 / v75 (v76 / v77): min / max clauses decided for the lhs (rhs) for the lane's pixel
 */
#pragma once
#include <cstdint>
#include <vector>

namespace mpr {

constexpr int TILE_GEN_MAX_SLOTS = 24;      /* slots 0..23 */
constexpr int TILE_GEN_MAX_CHOICES = 64;
constexpr int TILE_GEN_PRESENCE_WORDS = 24; /* tapes of up to 768 clauses can be shortened again by generated code */
/* a tile's record (TileStageArgs::gen_decisions), 64-bit words: the root tape's min / max clauses decided for the lhs, for the
 * rhs (its own decisions and everything decided above it), those its tape keeps as min / max, 0; then one bit per clause of
 * the root tape: on the tile's tape (TILE_GEN_PRESENCE_WORDS dwords) */
constexpr int TILE_GEN_RECORD_U64 = 4 + TILE_GEN_PRESENCE_WORDS / 2;

enum TileGenReg : int {
    TG_RT_SQUARE = 62, TG_RT_ABS = 64, TG_RT_MUL = 66, TG_RT_SQRT = 68, TG_RT_MIN = 70, TG_RT_MAX = 80, TG_RT_DIV = 82,
    TG_RT_DIVI = 84, TG_RT_ASIN = 86, TG_RT_ACOS = 88, TG_RT_ATAN = 90, TG_RT_EXP = 98, TG_RT_LOG = 96,
    TG_RET_ROUTINE = 36, TG_RET_CODE = 38,
    TG_RT_CHUNK = 62,
    /* the normals pass's Deriv walk: routines return through s[70:71] */
    TD_RT_DIV = 72, TD_RT_SQRT = 74, TD_RT_EXP = 76, TD_RT_LOG = 78, TD_RT_SINCOS = 68, TD_RT_ASIN = 80, TD_RT_ACOS = 82, TD_RT_ATAN = 84,
};

struct TileGen {
    bool ok = false;
    std::vector<uint32_t> bwd;      /* backward walk of tape pushing */
    std::vector<uint32_t> bwd_full; /* the same for tapes that are shortened again, and for the stages below the first (tile_gen.cpp); empty:
                                     * the tape has more clauses than a record has presence bits */
    std::vector<uint32_t> deriv;    /* the normals pass's walk (value + three partials per pixel: four lanes), decisions in v74..v77 */
    std::vector<uint32_t> deriv_guarded;   /* the same walk jumping over the runs that are dead for EVERY pixel of the wavefront: s[64:65] / s[66:67] =
                                            * min / max clauses decided for the lhs / rhs in all 64 lanes (the AND of their v74..v77) */
    int words = 0;                  /* clause words a walk visits: the operations and the end clause (or the head) */
    int nchoices = 0;               /* min / max clauses */
    int result_slot = 0;
};

/* clauses: head, operations, end (the host copy of a root tape).  ok == false: the tape does not fit the conventions
 * above (a slot beyond 23, more than 64 min / max clauses, a jump or an unknown opcode) — the interpreter walks it. */
TileGen tile_gen_build(const uint64_t* clauses, int len);

/* The backward walk (tape pushing, reference src/context.cu:323-458) of a FIRST stage for tapes those conventions do not fit but the
 * interpreter with 93 slots in registers does (slots 0..95, up to 4096 min / max clauses): the same per-lane walk as TileGen::bwd — the
 * clause's out slot active? take a word of the lane's chunk, mark the operands, store the clause, a decided min / max as the COPY it
 * becomes — with the active slots in three registers and the choices read from the 16-byte-per-clause records the forward walk left in
 * LDS (the interpreter's format: interval_gen.hpp: IW_FIRST_MASKS writes the same).  Run by tile_gen_asm.hpp: tile_gen_backward_big.
 * Empty: the tape does not fit. */
std::vector<uint32_t> tile_gen_build_big_backward(const uint64_t* clauses, int len, int* nchoices = nullptr);

}  // namespace mpr
