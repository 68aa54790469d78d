/*
 * tile_gen_asm.hpp — device side of tile_gen.hpp: the harnesses that run the ROOT tape's generated interval code for the 64
 * tiles of a wavefront (k_eval_tiles<.., GEN>), forward (reference src/context.cu:236-279) and backward (tape pushing,
 * :351-458).  The generated code is a straight line of instructions per clause; what a clause cannot do in three or four
 * instructions it calls here: the interpreter's own interval routines (TI_BODIES_TEXT, ending in a return instead of a
 * dispatch), min / max variants that leave the lanes' decisions to the caller, and the "chunk full" step of the backward walk.
 * Registers: tile_gen.hpp.  The wave must be in round-up mode with all 64 lanes enabled.
 */
#pragma once
#include "tile_interp_asm.hpp"

namespace mprk {

#undef TI_ST
#undef TI_GO
#define TI_ST ""
#define TI_GO "s_setpc_b64 s[36:37]\n"
/* entry point of a routine, relative to L_pc (s[40:41]) */
#define TG_ADDR(lo, hi, label) "s_add_u32 s" #lo ", s40, " label "_%=-L_pc_%=\n s_addc_u32 s" #hi ", s41, 0\n"

/* ---- frames nobody reads: what a LOOSE walk owes the reference (round 4; the arithmetic itself: interval_gen.hpp, round 5) ----
 * The tile stages' exp / log (reference inc/gpu_interval.hpp:332-336, :382-390) are the correctly rounded enclosures
 * {RD(exp(lo)), RU(exp(hi))}, computed in double precision (OCML: 119 / 361 instructions per clause, most at the f64 rate): half of
 * an exact forward walk of bear's tape.  Which tiles a stage finds empty / filled / ambiguous, and which min / max clauses it
 * decides, is reference state only in frames that are READ; an ordinary frame (context.hip: !reference) owes the reference its
 * heights and normals, and those do not depend on how tight a SOUND enclosure is: a wider interval leaves a tile ambiguous that
 * the reference would have culled or filled (its children / voxels are evaluated and give the same heights: the hierarchy is
 * conservative at every level) or a min / max undecided that the reference would have decided (the comparison then picks the same
 * operand at every point).  What "sound" has to mean: the loose result of an operation encloses the exact one WHENEVER its operands
 * enclose the exact operands — by induction over the tape every loose interval then encloses the reference's, and whatever the loose
 * walk proves (a tile empty or filled, a min / max decided) the reference proves too; what it fails to prove is evaluated one level
 * further down, to the same pixels — as long as the reference's own proofs are facts about the float pass's values.  Two things
 * break that, and a loose walk stays away from both by asking for the exact walk (interval_gen.hpp):
 *  - an exact routine that is not inclusion-isotone: log returns the lower bound 0 for x.lo <= 0 and log(x.lo) — as low as -103 —
 *    for a tiny positive one (reference inc/gpu_interval.hpp:382-390), and a loose lower end can be 0 where the exact one is tiny;
 *  - exact bounds that are not facts: the same lower bound 0 where the logarithm is very negative (a steep exp / log blend far from
 *    its surface: the reference then drops the OTHER operand of a min, and its image lacks what that operand drew — tests: "smooth"),
 *    sqrt's [0, ..] over an interval whose negative part is NaN to the float pass, asin / acos outside [-1, 1] (tapes with those
 *    never run loose: mpr_tape::loose_ok). */

/* ---- the scheduled forward walk (interval_gen.hpp; round 5) ----
 * code: the walk to run (exact or loose, of the kind the stage needs); code_exact: the exact walk of the same kind, which a loose
 * walk asks for through s[60:61] when an operand left a routine's domain or a NaN came up (null: code is exact).  Registers:
 * interval_gen.hpp.  *redone: whether that happened (development: the redo rate). */
DEV void tile_gen_forward2(const uint32_t* code, const uint32_t* code_exact, unsigned char* smem_io, int lane, float2 x, float2 y, float2 z,
                           float2* res, uint32_t* chl, uint32_t* chr, unsigned long long decided_lhs, unsigned long long decided_rhs,
                           uint32_t* redone = nullptr)
{
    float* const io = reinterpret_cast<float*>(smem_io);
    io[lane] = x.x; io[64 + lane] = x.y; io[128 + lane] = y.x; io[192 + lane] = y.y; io[256 + lane] = z.x; io[320 + lane] = z.y;
    const uint32_t ioaddr = rdfirst((uint32_t)(uintptr_t)io);
    const uint32_t lane8 = (uint32_t)lane * 8u;
    if (lane == 0) {
        uint32_t* const u = reinterpret_cast<uint32_t*>(io) + 960;
        if (!code_exact) code_exact = code;
        u[0] = (uint32_t)decided_lhs; u[1] = (uint32_t)(decided_lhs >> 32);
        u[2] = (uint32_t)decided_rhs; u[3] = (uint32_t)(decided_rhs >> 32);
        u[4] = (uint32_t)(uintptr_t)code; u[5] = (uint32_t)((uintptr_t)code >> 32);
        u[6] = (uint32_t)(uintptr_t)code_exact; u[7] = (uint32_t)((uintptr_t)code_exact >> 32);
        u[8] = 0;
    }
    asm volatile(
        "L_again_%=:\n"
        "v_lshrrev_b32 v32, 1, %[lane8]\n v_add_u32 v32, %[io], v32\n"
        "ds_read_b32 v0, v32\n ds_read_b32 v1, v32 offset:256\n ds_read_b32 v2, v32 offset:512\n"
        "ds_read_b32 v3, v32 offset:768\n ds_read_b32 v4, v32 offset:1024\n ds_read_b32 v5, v32 offset:1280\n"
        "v_mov_b32 v33, %[io]\n"
        "ds_read_b128 v[44:47], v33 offset:3840\n ds_read_b128 v[48:51], v33 offset:3856\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_readfirstlane_b32 s72, v44\n v_readfirstlane_b32 s73, v45\n v_readfirstlane_b32 s74, v46\n v_readfirstlane_b32 s75, v47\n"
        "v_readfirstlane_b32 s34, v48\n v_readfirstlane_b32 s35, v49\n v_readfirstlane_b32 s76, v50\n v_readfirstlane_b32 s77, v51\n"
        "v_mov_b32 v56, 0\n v_mov_b32 v57, 0\n v_mov_b32 v58, 0\n v_mov_b32 v59, 0\n"
        "s_getpc_b64 s[40:41]\n"
        "L_pc_%=:\n"
        TG_ADDR(68, 69, "L_isqrt") TG_ADDR(82, 83, "L_gdiv") TG_ADDR(84, 85, "L_gdivi")
        TG_ADDR(86, 87, "L_casin") TG_ADDR(88, 89, "L_cacos") TG_ADDR(90, 91, "L_catan") TG_ADDR(98, 99, "L_cexp")
        TG_ADDR(96, 97, "L_clog") TG_ADDR(60, 61, "L_redo")
        "s_swappc_b64 s[38:39], s[34:35]\n"
        "v_lshrrev_b32 v32, 1, %[lane8]\n v_add_u32 v32, %[io], v32\n"
        "ds_write_b32 v32, v36 offset:1536\n ds_write_b32 v32, v37 offset:1792\n"
        "ds_write_b32 v32, v56 offset:2048\n ds_write_b32 v32, v57 offset:2304\n"
        "ds_write_b32 v32, v58 offset:2560\n ds_write_b32 v32, v59 offset:2816\n"
        "s_waitcnt lgkmcnt(0)\n"
        "s_branch L_end_%=\n"
        /* the walk again, on the exact code (the axes' intervals and the wave's words are still in LDS) */
        "L_redo_%=:\n"
        "v_mov_b32 v33, %[io]\n v_mov_b32 v52, s76\n v_mov_b32 v53, s77\n v_mov_b32 v54, 1\n"
        "ds_write_b32 v33, v52 offset:3856\n ds_write_b32 v33, v53 offset:3860\n ds_write_b32 v33, v54 offset:3872\n"
        "s_waitcnt lgkmcnt(0)\n"
        "s_branch L_again_%=\n"
        TI_BODIES_TEXT
        /* division: s[58:59] = lanes whose divisor v[38:39] contains zero (the handlers' tests, tile_interp_asm.hpp) */
        "L_gdiv_%=:\n"
        "v_cmp_ge_f32 s[58:59], 0, v38\n v_cmp_le_f32 vcc, 0, v39\n"
        "s_and_b64 s[58:59], s[58:59], vcc\n s_branch L_idiv_%=\n"
        "L_gdivi_%=:\n"                                  /* a constant divisor */
        "v_cmp_lg_f32 s[58:59], 0, v38\n s_nop 0\n"
        "s_not_b64 s[58:59], s[58:59]\n s_branch L_idiv_%=\n"
        "L_end_%=:\n"
        :
        : [lane8] "v"(lane8), [io] "s"(ioaddr)
        : "memory", "vcc", "scc",
          "s0", "s1", "s2", "s3", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17", "s18", "s19",
          "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31",
          "s34", "s35", "s36", "s37", "s38", "s39",
          "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54",
          "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71",
          "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79",
          "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96",
          "s97", "s98", "s99",
          "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", TI_V10(1), TI_V10(2), TI_V10(3), TI_V10(4), TI_V10(5), TI_V10(6), TI_V10(7), TI_V10(8), TI_V10(9), TI_V10(10),
          "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117");
    res->x = io[384 + lane];
    res->y = io[448 + lane];
    const uint32_t* const iw = reinterpret_cast<const uint32_t*>(io);
    chl[0] = iw[512 + lane];
    chl[1] = iw[576 + lane];
    chr[0] = iw[640 + lane];
    chr[1] = iw[704 + lane];
    if (redone) *redone = iw[968];
}

/* The same for LOOSE code that names v0..v63 only (interval_gen.hpp: IGEN_LEAN_VGPRS): no routines to call, a fifth of the vector
 * registers left to the compiler — the kernel around it runs six wavefronts per SIMD.  A walk that asks for the exact code is not
 * redone here: *redone = 1 and the results are garbage (k_eval_tiles<.., LEAN> hands the wavefront's tiles to the launch behind it). */
/* TIGHT: the code is tight code (interval_gen.hpp: it names v0..v79 and s[76:77] as well, and leaves a second enclosure of the
 * result in v[38:39]: *tight); the kernel around it runs five wavefronts per SIMD. */
template <bool TIGHT = false>
DEV void tile_gen_forward2_lean(const uint32_t* code, unsigned char* smem_io, int lane, float2 x, float2 y, float2 z,
                                float2* res, uint32_t* chl, uint32_t* chr, unsigned long long decided_lhs, unsigned long long decided_rhs,
                                uint32_t* redone, uint32_t* bad_lane = nullptr, float2* tight = nullptr)
{
    float* const io = reinterpret_cast<float*>(smem_io);
    io[lane] = x.x; io[64 + lane] = x.y; io[128 + lane] = y.x; io[192 + lane] = y.y; io[256 + lane] = z.x; io[320 + lane] = z.y;
    const uint32_t ioaddr = rdfirst((uint32_t)(uintptr_t)io);
    const uint32_t lane8 = (uint32_t)lane * 8u;
    if (lane == 0) {
        uint32_t* const u = reinterpret_cast<uint32_t*>(io) + 960;
        u[0] = (uint32_t)decided_lhs; u[1] = (uint32_t)(decided_lhs >> 32);
        u[2] = (uint32_t)decided_rhs; u[3] = (uint32_t)(decided_rhs >> 32);
        u[4] = (uint32_t)(uintptr_t)code; u[5] = (uint32_t)((uintptr_t)code >> 32);
        u[8] = 0;
    }
#define TG_LEAN_TEXT                                                                                                      \
        "v_lshrrev_b32 v32, 1, %[lane8]\n v_add_u32 v32, %[io], v32\n"                                                    \
        "ds_read_b32 v0, v32\n ds_read_b32 v1, v32 offset:256\n ds_read_b32 v2, v32 offset:512\n"                         \
        "ds_read_b32 v3, v32 offset:768\n ds_read_b32 v4, v32 offset:1024\n ds_read_b32 v5, v32 offset:1280\n"            \
        "v_mov_b32 v33, %[io]\n"                                                                                          \
        "ds_read_b128 v[44:47], v33 offset:3840\n ds_read_b64 v[48:49], v33 offset:3856\n"                                \
        "s_waitcnt lgkmcnt(0)\n"                                                                                          \
        "v_readfirstlane_b32 s72, v44\n v_readfirstlane_b32 s73, v45\n v_readfirstlane_b32 s74, v46\n v_readfirstlane_b32 s75, v47\n" \
        "v_readfirstlane_b32 s34, v48\n v_readfirstlane_b32 s35, v49\n"                                                   \
        "v_mov_b32 v56, 0\n v_mov_b32 v57, 0\n v_mov_b32 v58, 0\n v_mov_b32 v59, 0\n"                                     \
        "s_getpc_b64 s[40:41]\n"                                                                                          \
        "L_pc_%=:\n"                                                                                                      \
        TG_ADDR(60, 61, "L_redo")                                                                                         \
        "s_swappc_b64 s[38:39], s[34:35]\n"                                                                               \
        "v_lshrrev_b32 v32, 1, %[lane8]\n v_add_u32 v32, %[io], v32\n"                                                    \
        "ds_write_b32 v32, v36 offset:1536\n ds_write_b32 v32, v37 offset:1792\n"                                         \
        "ds_write_b32 v32, v56 offset:2048\n ds_write_b32 v32, v57 offset:2304\n"                                         \
        "ds_write_b32 v32, v58 offset:2560\n ds_write_b32 v32, v59 offset:2816\n"                                         \
        "v_cndmask_b32 v54, 0, 1, s[40:41]\n"             /* (the lanes that ask for the exact walk, of code that only reports them: tests) */ \
        "ds_write_b32 v32, v54 offset:3072\n"                                                                             \
        "ds_write_b32 v32, v38 offset:3328\n ds_write_b32 v32, v39 offset:3584\n"     /* (tight code: the second result) */ \
        "s_waitcnt lgkmcnt(0)\n"                                                                                          \
        "s_branch L_end_%=\n"                                                                                             \
        "L_redo_%=:\n"                                                                                                    \
        "v_mov_b32 v33, %[io]\n v_mov_b32 v54, 1\n"                                                                       \
        "ds_write_b32 v33, v54 offset:3872\n"                                                                             \
        "s_waitcnt lgkmcnt(0)\n"                                                                                          \
        "L_end_%=:\n"
#define TG_LEAN_CLOBBERS                                                                                                  \
          "memory", "vcc", "scc",                                                                                         \
          "s0", "s1", "s2", "s3", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17", "s18", "s19", \
          "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31",                             \
          "s34", "s35", "s36", "s37", "s38", "s39",                                                                       \
          "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54",        \
          "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", \
          "s72", "s73", "s74", "s75",                                                                                     \
          "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", \
          "s97", "s98", "s99",                                                                                            \
          "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", TI_V10(1), TI_V10(2), TI_V10(3), TI_V10(4), TI_V10(5), \
          "v60", "v61", "v62", "v63"
    if constexpr (TIGHT) {
        asm volatile(TG_LEAN_TEXT : : [lane8] "v"(lane8), [io] "s"(ioaddr)
                     : TG_LEAN_CLOBBERS, "s76", "s77", "v64", "v65", "v66", "v67", "v68", "v69", TI_V10(7));
    } else {
        asm volatile(TG_LEAN_TEXT : : [lane8] "v"(lane8), [io] "s"(ioaddr) : TG_LEAN_CLOBBERS);
    }
#undef TG_LEAN_TEXT
#undef TG_LEAN_CLOBBERS
    res->x = io[384 + lane];
    res->y = io[448 + lane];
    const uint32_t* const iw = reinterpret_cast<const uint32_t*>(io);
    chl[0] = iw[512 + lane];
    chl[1] = iw[576 + lane];
    chr[0] = iw[640 + lane];
    chr[1] = iw[704 + lane];
    *redone = iw[968];
    if (bad_lane) *bad_lane = iw[768 + lane];
    if (tight) {
        tight->x = io[832 + lane];
        tight->y = io[896 + lane];
    }
}

/* A first stage's walk of a tape beyond 24 slots / 64 min / max clauses (interval_gen.hpp: IW_FIRST_MASKS; loose, no routines) inside
 * the kernel whose interpreter keeps 93 slots in registers: the code names v0..v60 and v64..v253, records its choices where that
 * interpreter's forward walk records them — smem: ulonglong2[choice_cap], then [8][64] floats of scratch (tile_interp_asm.hpp:
 * tile_interp_asm_vgpr) — and returns the lanes that decided anything and the lanes that ask for the exact walk (the caller
 * runs the interpreter then: nothing the code wrote is kept). */
DEV void tile_gen_forward_big(const uint32_t* code, unsigned char* smem, int choice_cap, int lane, float2 x, float2 y, float2 z,
                              float2* res, uint64_t* any_choice, uint64_t* asks_exact)
{
    float* const io = reinterpret_cast<float*>(smem + (size_t)choice_cap * 16);
    io[lane] = x.x; io[64 + lane] = x.y; io[128 + lane] = y.x; io[192 + lane] = y.y; io[256 + lane] = z.x; io[320 + lane] = z.y;
    const uint32_t ioaddr = rdfirst((uint32_t)(uintptr_t)io);
    const uint32_t caddr = rdfirst((uint32_t)(uintptr_t)smem);
    const uint32_t clo = rdfirst((uint32_t)(uintptr_t)code), chi = rdfirst((uint32_t)((uintptr_t)code >> 32));
    const uint32_t lane4 = (uint32_t)lane * 4u;
    uint32_t anylo = 0, anyhi = 0, badlo = 0, badhi = 0;
    asm volatile(
        "v_add_u32 v32, %[io], %[lane4]\n"
        "ds_read_b32 v0, v32\n ds_read_b32 v1, v32 offset:256\n ds_read_b32 v2, v32 offset:512\n"
        "ds_read_b32 v3, v32 offset:768\n ds_read_b32 v4, v32 offset:1024\n ds_read_b32 v5, v32 offset:1280\n"
        "v_lshlrev_b32 v60, 2, %[lane4]\n"
        "v_add_u32 v60, %[caddr], v60\n"                    /* the lane's entry of the first 64 choices */
        "s_mov_b32 s34, %[clo]\n s_mov_b32 s35, %[chi]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "s_swappc_b64 s[38:39], s[34:35]\n"
        "v_add_u32 v32, %[io], %[lane4]\n"
        "ds_write_b32 v32, v36 offset:1536\n ds_write_b32 v32, v37 offset:1792\n"
        "s_mov_b32 %[anylo], s78\n s_mov_b32 %[anyhi], s79\n s_mov_b32 %[badlo], s40\n s_mov_b32 %[badhi], s41\n"
        "s_waitcnt lgkmcnt(0)\n"
        : [anylo] "=&s"(anylo), [anyhi] "=&s"(anyhi), [badlo] "=&s"(badlo), [badhi] "=&s"(badhi)
        : [lane4] "v"(lane4), [io] "s"(ioaddr), [caddr] "s"(caddr), [clo] "s"(clo), [chi] "s"(chi)
        : "memory", "vcc", "scc",
          "s0", "s1", "s2", "s3", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17", "s18", "s19",
          "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31",
          "s34", "s35", "s36", "s37", "s38", "s39", "s40", "s41", "s78", "s79",
          "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", TI_V10(1), TI_V10(2), TI_V10(3), TI_V10(4), TI_V10(5), "v60",
          "v64", "v65", "v66", "v67", "v68", "v69", TI_V10(7), TI_V10(8), TI_V10(9), TI_V10(10), TI_V10(11),
          TI_V10(12), TI_V10(13), TI_V10(14), TI_V10(15), TI_V10(16), TI_V10(17), TI_V10(18), TI_V10(19), TI_V10(20),
          TI_V10(21), TI_V10(22), TI_V10(23), TI_V10(24), "v250", "v251", "v252", "v253");
    res->x = io[384 + lane];
    res->y = io[448 + lane];
    *any_choice = ((uint64_t)anyhi << 32) | anylo;
    *asks_exact = ((uint64_t)badhi << 32) | badlo;
}

/* The backward walk.  Per lane in: active (bit = slot: the end clause's out slot for a pushing lane, 0 otherwise), pos = pool
 * index of the last word written (the end clause), first = first index of its chunk, run_end = first index past its run of
 * chunks; the decisions.  Out: pos and first where the walk stopped (the head goes to pos - 1), overflow (ran out of chunks),
 * kept = min / max clauses the lane's tape keeps.  pool_limit: last index a chunk may start at. */
struct TileGenPush {
    uint32_t active, pos, first, run_end;
    uint32_t overflow, kept;
    uint32_t kept_lo, kept_hi;     /* out: which min / max clauses the lane's tape keeps (bit k: the root tape's k-th) */
};
DEV void tile_gen_backward(const uint32_t* code, const uint64_t* pool, unsigned char* smem_io, int lane, TileGenPush& st,
                           const uint32_t* chl, const uint32_t* chr, uint32_t pool_limit)
{
    uint32_t* const io = reinterpret_cast<uint32_t*>(smem_io);
    io[lane] = st.active; io[64 + lane] = st.pos; io[128 + lane] = st.first; io[192 + lane] = st.run_end;
    io[256 + lane] = chl[0]; io[320 + lane] = chl[1]; io[384 + lane] = chr[0]; io[448 + lane] = chr[1];
    const uint32_t ioaddr = rdfirst((uint32_t)(uintptr_t)io);
    const uint32_t lane8 = (uint32_t)lane * 8u;
    const uint32_t clo = rdfirst((uint32_t)(uintptr_t)code), chi = rdfirst((uint32_t)((uintptr_t)code >> 32));
    const uint32_t plo = rdfirst((uint32_t)(uintptr_t)pool), phi = rdfirst((uint32_t)((uintptr_t)pool >> 32));
    const uint32_t plim = rdfirst(pool_limit);
    asm volatile(
        "v_lshrrev_b32 v32, 1, %[lane8]\n v_add_u32 v32, %[io], v32\n"
        "ds_read_b32 v60, v32\n ds_read_b32 v61, v32 offset:256\n ds_read_b32 v62, v32 offset:512\n ds_read_b32 v63, v32 offset:768\n"
        "ds_read_b32 v56, v32 offset:1024\n ds_read_b32 v57, v32 offset:1280\n ds_read_b32 v58, v32 offset:1536\n ds_read_b32 v59, v32 offset:1792\n"
        "v_mov_b32 v47, 0\n v_mov_b32 v54, 0\n v_mov_b32 v55, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n"
        "s_mov_b32 s76, %[plo]\n s_mov_b32 s77, %[phi]\n s_mov_b32 s98, %[plim]\n"
        "s_getpc_b64 s[40:41]\n"
        "L_pc_%=:\n"
        TG_ADDR(62, 63, "L_chunk")
        "s_mov_b32 s34, %[clo]\n"
        "s_mov_b32 s35, %[chi]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "s_swappc_b64 s[38:39], s[34:35]\n"
        "v_lshrrev_b32 v32, 1, %[lane8]\n v_add_u32 v32, %[io], v32\n"
        "ds_write_b32 v32, v61 offset:2048\n ds_write_b32 v32, v62 offset:2304\n"
        "ds_write_b32 v32, v55 offset:2560\n ds_write_b32 v32, v54 offset:2816\n"
        "ds_write_b32 v32, v41 offset:3072\n ds_write_b32 v32, v42 offset:3328\n"
        "s_waitcnt lgkmcnt(0)\n"
        "s_branch L_end_%=\n"
        /* some lane's chunk is full (pos == first: word 0 is the link's): it moves to the next chunk of its run and writes the
         * two links (reference :384-413) or, out of room, stops pushing (its flags v32 / v35 of the clause under way are
         * cleared, so that the clause does nothing for it) */
        "L_chunk_%=:\n"
        "v_cmp_eq_u32 vcc, v61, v62\n"
        "s_mov_b64 exec, vcc\n"
        "v_mov_b32 v52, v62\n"
        "v_add_u32 v62, 64, v62\n"
        "v_cmp_ge_u32 vcc, v62, v63\n"
        "v_cmp_gt_u32 s[92:93], v62, s98\n"
        "s_or_b64 vcc, vcc, s[92:93]\n"
        "v_cndmask_b32 v60, v60, 0, vcc\n"
        "v_cndmask_b32 v32, v32, 0, vcc\n"
        "v_cndmask_b32 v35, v35, 0, vcc\n"
        "v_cndmask_b32 v55, v55, 1, vcc\n"
        "s_andn2_b64 exec, exec, vcc\n"
        "v_add_lshl_u32 v44, v62, 63, 3\n"
        "v_lshlrev_b32 v45, 3, v52\n"
        "v_mov_b32 v48, 1\n"                              /* word 63 of the new chunk: JUMP back to the previous one (-127) */
        "v_mov_b32 v49, 0xffffff81\n"
        "v_mov_b32 v50, 1\n"                              /* word 0 of the previous chunk: JUMP forward (+127) */
        "v_mov_b32 v51, 127\n"
        "global_store_dwordx2 v44, v[48:49], s[76:77]\n"
        "global_store_dwordx2 v45, v[50:51], s[76:77]\n"
        "v_add_u32 v61, 62, v62\n"
        "s_mov_b64 exec, -1\n"
        "s_setpc_b64 s[36:37]\n"
        "L_end_%=:\n"
        :
        : [lane8] "v"(lane8), [io] "s"(ioaddr), [clo] "s"(clo), [chi] "s"(chi), [plo] "s"(plo), [phi] "s"(phi), [plim] "s"(plim)
        : "memory", "vcc", "scc", "s34", "s35", "s36", "s37", "s38", "s39", "s40", "s41", "s62", "s63", "s76", "s77", "s92", "s93", "s98",
          "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52",
          "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
    st.pos = io[512 + lane];
    st.first = io[576 + lane];
    st.overflow = io[640 + lane];
    st.kept = io[704 + lane];
    st.kept_lo = io[768 + lane];
    st.kept_hi = io[832 + lane];
}

/* The backward walk of a first stage for tapes beyond 24 slots / 64 min / max clauses (tile_gen.hpp: tile_gen_build_big_backward) inside the
 * kernel whose interpreter keeps 93 slots in registers.  smem: ulonglong2[choice_cap] — the forward walk's records of its choices —, then
 * [8][64] words of scratch (tile_interp_asm.hpp: tile_interp_asm_vgpr).  active3: the lane's active slots (bit s & 31 of word s >> 5). */
DEV void tile_gen_backward_big(const uint32_t* code, const uint64_t* pool, unsigned char* smem, int choice_cap, int lane, TileGenPush& st,
                               uint32_t active1, uint32_t active2, uint32_t pool_limit)
{
    uint32_t* const io = reinterpret_cast<uint32_t*>(smem + (size_t)choice_cap * 16);
    io[lane] = st.active; io[64 + lane] = active1; io[128 + lane] = active2; io[192 + lane] = st.pos; io[256 + lane] = st.first; io[320 + lane] = st.run_end;
    const uint32_t ioaddr = rdfirst((uint32_t)(uintptr_t)io);
    const uint32_t caddr = rdfirst((uint32_t)(uintptr_t)smem);
    const uint32_t lane4 = (uint32_t)lane * 4u;
    const uint32_t clo = rdfirst((uint32_t)(uintptr_t)code), chi = rdfirst((uint32_t)((uintptr_t)code >> 32));
    const uint32_t plo = rdfirst((uint32_t)(uintptr_t)pool), phi = rdfirst((uint32_t)((uintptr_t)pool >> 32));
    const uint32_t plim = rdfirst(pool_limit);
    asm volatile(
        "v_add_u32 v32, %[io], %[lane4]\n"
        "ds_read_b32 v60, v32\n ds_read_b32 v64, v32 offset:256\n ds_read_b32 v65, v32 offset:512\n"
        "ds_read_b32 v61, v32 offset:768\n ds_read_b32 v62, v32 offset:1024\n ds_read_b32 v63, v32 offset:1280\n"
        "v_mov_b32 v47, 0\n v_mov_b32 v54, 0\n v_mov_b32 v55, 0\n"
        "v_lshrrev_b32 v74, 2, %[lane4]\n v_and_b32 v74, 31, v74\n"
        "v_mov_b32 v75, %[caddr]\n"
        "s_mov_b32 s64, 0\n s_mov_b32 s65, -1\n"                 /* lanes 32..63 */
        "s_mov_b32 s76, %[plo]\n s_mov_b32 s77, %[phi]\n s_mov_b32 s98, %[plim]\n"
        "s_getpc_b64 s[40:41]\n"
        "L_pc_%=:\n"
        TG_ADDR(62, 63, "L_chunk")
        "s_mov_b32 s34, %[clo]\n"
        "s_mov_b32 s35, %[chi]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "s_swappc_b64 s[38:39], s[34:35]\n"
        "v_add_u32 v32, %[io], %[lane4]\n"
        "ds_write_b32 v32, v61\n ds_write_b32 v32, v62 offset:256\n"
        "ds_write_b32 v32, v55 offset:512\n ds_write_b32 v32, v54 offset:768\n"
        "s_waitcnt lgkmcnt(0)\n"
        "s_branch L_end_%=\n"
        /* some lane's chunk is full: tile_gen_backward's routine with three words of active slots to clear for a lane out of room */
        "L_chunk_%=:\n"
        "v_cmp_eq_u32 vcc, v61, v62\n"
        "s_mov_b64 exec, vcc\n"
        "v_mov_b32 v52, v62\n"
        "v_add_u32 v62, 64, v62\n"
        "v_cmp_ge_u32 vcc, v62, v63\n"
        "v_cmp_gt_u32 s[92:93], v62, s98\n"
        "s_or_b64 vcc, vcc, s[92:93]\n"
        "v_cndmask_b32 v60, v60, 0, vcc\n"
        "v_cndmask_b32 v64, v64, 0, vcc\n"
        "v_cndmask_b32 v65, v65, 0, vcc\n"
        "v_cndmask_b32 v32, v32, 0, vcc\n"
        "v_cndmask_b32 v35, v35, 0, vcc\n"
        "v_cndmask_b32 v55, v55, 1, vcc\n"
        "s_andn2_b64 exec, exec, vcc\n"
        "v_add_lshl_u32 v44, v62, 63, 3\n"
        "v_lshlrev_b32 v45, 3, v52\n"
        "v_mov_b32 v48, 1\n"
        "v_mov_b32 v49, 0xffffff81\n"
        "v_mov_b32 v50, 1\n"
        "v_mov_b32 v51, 127\n"
        "global_store_dwordx2 v44, v[48:49], s[76:77]\n"
        "global_store_dwordx2 v45, v[50:51], s[76:77]\n"
        "v_add_u32 v61, 62, v62\n"
        "s_mov_b64 exec, -1\n"
        "s_setpc_b64 s[36:37]\n"
        "L_end_%=:\n"
        :
        : [lane4] "v"(lane4), [io] "s"(ioaddr), [caddr] "s"(caddr), [clo] "s"(clo), [chi] "s"(chi), [plo] "s"(plo), [phi] "s"(phi), [plim] "s"(plim)
        : "memory", "vcc", "scc", "s34", "s35", "s36", "s37", "s38", "s39", "s40", "s41", "s62", "s63", "s64", "s65", "s76", "s77", "s92", "s93", "s98",
          "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52",
          "v54", "v55", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v74", "v75");
    st.pos = io[lane];
    st.first = io[64 + lane];
    st.overflow = io[128 + lane];
    st.kept = io[192 + lane];
}

/* The backward walk for tapes that are shortened again (tile_gen.cpp: backward_full_clause).  decided_lhs / decided_rhs:
 * min / max clauses that are copies already on the tape being shortened; parent_presence: that tape's clauses, one bit per
 * clause of the root tape (TILE_GEN_PRESENCE_WORDS dwords; null: all of them — the first stage); the presence bits of the
 * tapes written go to presence_out + presence_off (per lane, bytes) for the lanes that entered with an active slot. */
DEV void tile_gen_backward_full(const uint32_t* code, const uint64_t* pool, unsigned char* smem_io, int lane, TileGenPush& st,
                                const uint32_t* chl, const uint32_t* chr, uint32_t pool_limit, unsigned long long decided_lhs,
                                unsigned long long decided_rhs, const uint32_t* parent_presence, uint32_t* presence_out, uint32_t presence_off)
{
    uint32_t* const io = reinterpret_cast<uint32_t*>(smem_io);
    io[lane] = st.active; io[64 + lane] = st.pos; io[128 + lane] = st.first; io[192 + lane] = st.run_end;
    io[256 + lane] = chl[0]; io[320 + lane] = chl[1]; io[384 + lane] = chr[0]; io[448 + lane] = chr[1];
    io[896 + lane] = presence_off;
    const uint32_t ioaddr = rdfirst((uint32_t)(uintptr_t)io);
    const uint32_t lane8 = (uint32_t)lane * 8u;
    const uint32_t clo = rdfirst((uint32_t)(uintptr_t)code), chi = rdfirst((uint32_t)((uintptr_t)code >> 32));
    const uint32_t plo = rdfirst((uint32_t)(uintptr_t)pool), phi = rdfirst((uint32_t)((uintptr_t)pool >> 32));
    const uint32_t plim = rdfirst(pool_limit);
    const uint32_t llo = rdfirst((uint32_t)decided_lhs), lhi = rdfirst((uint32_t)(decided_lhs >> 32));
    const uint32_t rlo = rdfirst((uint32_t)decided_rhs), rhi = rdfirst((uint32_t)(decided_rhs >> 32));
    const uint32_t qlo = rdfirst((uint32_t)(uintptr_t)parent_presence), qhi = rdfirst((uint32_t)((uintptr_t)parent_presence >> 32));
    const uint32_t olo = rdfirst((uint32_t)(uintptr_t)presence_out), ohi = rdfirst((uint32_t)((uintptr_t)presence_out >> 32));
    asm volatile(
        "v_lshrrev_b32 v32, 1, %[lane8]\n v_add_u32 v32, %[io], v32\n"
        "ds_read_b32 v60, v32\n ds_read_b32 v61, v32 offset:256\n ds_read_b32 v62, v32 offset:512\n ds_read_b32 v63, v32 offset:768\n"
        "ds_read_b32 v56, v32 offset:1024\n ds_read_b32 v57, v32 offset:1280\n ds_read_b32 v58, v32 offset:1536\n ds_read_b32 v59, v32 offset:1792\n"
        "ds_read_b32 v53, v32 offset:3584\n"
        "v_mov_b32 v47, 0\n v_mov_b32 v54, 0\n v_mov_b32 v55, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n"
        "v_mov_b32 v64, 0\n v_mov_b32 v65, 0\n v_mov_b32 v66, 0\n v_mov_b32 v67, 0\n v_mov_b32 v68, 0\n v_mov_b32 v69, 0\n"
        "v_mov_b32 v70, 0\n v_mov_b32 v71, 0\n v_mov_b32 v72, 0\n v_mov_b32 v73, 0\n v_mov_b32 v74, 0\n v_mov_b32 v75, 0\n"
        "v_mov_b32 v76, 0\n v_mov_b32 v77, 0\n v_mov_b32 v78, 0\n v_mov_b32 v79, 0\n v_mov_b32 v80, 0\n v_mov_b32 v81, 0\n"
        "v_mov_b32 v82, 0\n v_mov_b32 v83, 0\n v_mov_b32 v84, 0\n v_mov_b32 v85, 0\n v_mov_b32 v86, 0\n v_mov_b32 v87, 0\n"
        "s_mov_b32 s76, %[plo]\n s_mov_b32 s77, %[phi]\n s_mov_b32 s98, %[plim]\n"
        "s_mov_b32 s64, %[llo]\n s_mov_b32 s65, %[lhi]\n s_mov_b32 s66, %[rlo]\n s_mov_b32 s67, %[rhi]\n"
        "s_mov_b32 s68, %[olo]\n s_mov_b32 s69, %[ohi]\n"
        /* the parent's tape: s[0:23] */
        "s_mov_b32 s34, %[qlo]\n s_mov_b32 s35, %[qhi]\n"
        "s_cmp_eq_u64 s[34:35], 0\n"
        "s_cbranch_scc1 L_all_%=\n"
        "s_load_dwordx8 s[0:7], s[34:35], 0x0\n s_load_dwordx8 s[8:15], s[34:35], 0x20\n s_load_dwordx8 s[16:23], s[34:35], 0x40\n"
        "s_branch L_have_%=\n"
        "L_all_%=:\n"
        "s_mov_b64 s[0:1], -1\n s_mov_b64 s[2:3], -1\n s_mov_b64 s[4:5], -1\n s_mov_b64 s[6:7], -1\n s_mov_b64 s[8:9], -1\n s_mov_b64 s[10:11], -1\n"
        "s_mov_b64 s[12:13], -1\n s_mov_b64 s[14:15], -1\n s_mov_b64 s[16:17], -1\n s_mov_b64 s[18:19], -1\n s_mov_b64 s[20:21], -1\n s_mov_b64 s[22:23], -1\n"
        "L_have_%=:\n"
        "s_getpc_b64 s[40:41]\n"
        "L_pc_%=:\n"
        TG_ADDR(62, 63, "L_chunk")
        "s_mov_b32 s34, %[clo]\n"
        "s_mov_b32 s35, %[chi]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_cmp_ne_u32 s[70:71], 0, v60\n"                /* the lanes that write a tape */
        "s_swappc_b64 s[38:39], s[34:35]\n"
        "v_lshrrev_b32 v32, 1, %[lane8]\n v_add_u32 v32, %[io], v32\n"
        "ds_write_b32 v32, v61 offset:2048\n ds_write_b32 v32, v62 offset:2304\n"
        "ds_write_b32 v32, v55 offset:2560\n ds_write_b32 v32, v54 offset:2816\n"
        "ds_write_b32 v32, v41 offset:3072\n ds_write_b32 v32, v42 offset:3328\n"
        "s_mov_b64 exec, s[70:71]\n"
        "global_store_dwordx4 v53, v[64:67], s[68:69]\n global_store_dwordx4 v53, v[68:71], s[68:69] offset:16\n"
        "global_store_dwordx4 v53, v[72:75], s[68:69] offset:32\n global_store_dwordx4 v53, v[76:79], s[68:69] offset:48\n"
        "global_store_dwordx4 v53, v[80:83], s[68:69] offset:64\n global_store_dwordx4 v53, v[84:87], s[68:69] offset:80\n"
        "s_mov_b64 exec, -1\n"
        "s_waitcnt lgkmcnt(0)\n"
        "s_branch L_end_%=\n"
        "L_chunk_%=:\n"
        "v_cmp_eq_u32 vcc, v61, v62\n"
        "s_mov_b64 exec, vcc\n"
        "v_mov_b32 v52, v62\n"
        "v_add_u32 v62, 64, v62\n"
        "v_cmp_ge_u32 vcc, v62, v63\n"
        "v_cmp_gt_u32 s[92:93], v62, s98\n"
        "s_or_b64 vcc, vcc, s[92:93]\n"
        "v_cndmask_b32 v60, v60, 0, vcc\n"
        "v_cndmask_b32 v32, v32, 0, vcc\n"
        "v_cndmask_b32 v35, v35, 0, vcc\n"
        "v_cndmask_b32 v55, v55, 1, vcc\n"
        "s_andn2_b64 exec, exec, vcc\n"
        "v_add_lshl_u32 v44, v62, 63, 3\n"
        "v_lshlrev_b32 v45, 3, v52\n"
        "v_mov_b32 v48, 1\n"
        "v_mov_b32 v49, 0xffffff81\n"
        "v_mov_b32 v50, 1\n"
        "v_mov_b32 v51, 127\n"
        "global_store_dwordx2 v44, v[48:49], s[76:77]\n"
        "global_store_dwordx2 v45, v[50:51], s[76:77]\n"
        "v_add_u32 v61, 62, v62\n"
        "s_mov_b64 exec, -1\n"
        "s_setpc_b64 s[36:37]\n"
        "L_end_%=:\n"
        :
        : [lane8] "v"(lane8), [io] "s"(ioaddr), [clo] "s"(clo), [chi] "s"(chi), [plo] "s"(plo), [phi] "s"(phi), [plim] "s"(plim),
          [llo] "s"(llo), [lhi] "s"(lhi), [rlo] "s"(rlo), [rhi] "s"(rhi), [qlo] "s"(qlo), [qhi] "s"(qhi), [olo] "s"(olo), [ohi] "s"(ohi)
        : "memory", "vcc", "scc", "s0", "s1", "s2", "s3", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13", "s14", "s15",
          "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23",
          "s34", "s35", "s36", "s37", "s38", "s39", "s40", "s41", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s76", "s77",
          "s92", "s93", "s98",
          "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53",
          "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63",
          "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79",
          "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87");
    st.pos = io[512 + lane];
    st.first = io[576 + lane];
    st.overflow = io[640 + lane];
    st.kept = io[704 + lane];
    st.kept_lo = io[768 + lane];
    st.kept_hi = io[832 + lane];
}

#undef TG_ADDR

}  // namespace mprk
