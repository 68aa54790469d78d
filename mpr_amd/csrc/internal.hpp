/* internal.hpp — definitions shared by the translation units of libmpr_amd.so */
#pragma once
#include <atomic>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "tape_schedule.hpp"

namespace mpr {
/* A tape's walks as gfx950 machine code (tile_gen.hpp: interval forward / backward / Deriv / backward for tapes that are
 * shortened again; voxel_gen.hpp: the float walk), generated once, on the host, when the tape is made — not in the first frame
 * that renders it — and shared by every copy of the tape.  words = the seven pieces back to back. */
struct TapeCode {
    std::vector<uint32_t> words;
    int fwd_dw = 0, bwd_dw = 0, deriv_dw = 0, full_dw = 0, vox_dw = 0, fwdg_dw = 0, derivg_dw = 0;   /* (in this order; fwdg / derivg: the guarded walks) */
    int walk_words = 0, nchoices = 0;
    /* round 5 (interval_gen.hpp), behind the seven: the scheduled interval forward walks [kind: first / below / below, guarded][exact, loose,
     * round 6: tight]: where in `words`, and how many dwords (0: none — loose: the tape has clauses the loose arithmetic does not take; tight:
     * or no sin / cos) */
    int iw_at[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, iw_dw[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    int iw_instructions[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    int vox_min_run = 0;             /* the shortest guarded run the float walk was generated with */
};
constexpr int TAPE_CODE_DEFAULT_MIN_RUN = 5;
/* null: the generators do not take this tape (more than 24 slots or 64 min / max clauses, ...) */
std::shared_ptr<const TapeCode> build_tape_code(const uint64_t* clauses, int len, int vox_min_run);
}  // namespace mpr

struct mpr_tape {
    std::vector<uint64_t> clauses;   /* head, operations, end — host copy */
    int32_t num_slots = 0;           /* highest slot index used + 1 */
    int32_t num_choices = 0;         /* min/max clauses */
    int32_t flags = 0;
    bool loose_ok = true;          /* no asin / acos, no division by a constant that is zero or outside 2^-100 .. 2^100: the tile stages of
                                      frames nobody reads may take the loose enclosures (tile_gen_asm.hpp: TG_LOOSE_ROUTINES) */
    uint64_t serial = 0;             /* identity, so a context can cache per-tape state */
    mpr::TapeSchedule schedule;      /* dependency levels for the wide first-stage kernel (ok == false: not usable) */
    std::shared_ptr<const mpr::TapeCode> code;   /* its walks as machine code, or null */
    /* tapes the generators above do not take (more than 24 slots or 64 min / max clauses) but the interpreter with 93 slots in registers
     * does: a first stage's loose forward walk that records its choices for the interpreter's backward walk (interval_gen.hpp:
     * IW_FIRST_MASKS), or null (asin / acos / atan, or more live values than there are registers) */
    std::shared_ptr<const std::vector<uint32_t>> big_fwd;
    int32_t big_end = 0;             /* index of the end clause */
    /* ... and the backward walk to go with it (tile_gen.hpp: tile_gen_build_big_backward), or null */
    std::shared_ptr<const std::vector<uint32_t>> big_bwd;
};

namespace mpr {
int set_error(int code, const std::string& msg);
}
