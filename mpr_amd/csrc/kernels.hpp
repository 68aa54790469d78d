/* kernels.hpp — argument blocks and host-side launchers of the gfx950 kernels (kernels.hip) */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mpr_clause.h"

namespace mprk {

/* device counter slots (uint64 each) */
enum {
    CNT_FWD = 0,       /* words fetched forward, all passes */
    CNT_BWD,           /* words fetched backward */
    CNT_WRITTEN,       /* words written by tape pushes */
    CNT_LANE,          /* lane-granular clause evaluations */
    CNT_FWD_VOX,       /* forward words of the float pass */
    CNT_FWD_NORM,      /* forward words of the normals pass */
    CNT_NORMAL_PX,
    CNT_OVERFLOW,
    CNT_COUNT
};

/* Per sibling group (64 consecutive entries of a tile list = the children of one tile) of the LAST tile stage: the
 * tape the group walked, and how many min / max decisions were recorded for it; choice_masks[g * choice_cap + i] =
 * {tiles that chose lhs, tiles that chose rhs} at the i-th min / max clause of that tape.  The float pass's group
 * form evaluates a child with its group's tape and its column of these masks (kernels_voxel_jit.hip). */
struct GroupInfo {
    int tape;
    int nchoices;
    unsigned long long pushed;     /* the tiles whose own tape is this one shortened by their decisions (ambiguous tiles that
                                      chose a side somewhere): the others' own tape IS this one */
    int parent;                    /* index, in its stage's list, of the tile these 64 are the children of (its record:
                                      TileStageArgs::gen_parent) */
    int reserved;
};

/* a tile's record of what it decided (csrc/tile_gen.hpp: TILE_GEN_RECORD_U64, TILE_GEN_PRESENCE_WORDS), 64-bit words */
constexpr int GEN_RECORD_U64 = 16, GEN_PRESENCE_WORDS = 24;

struct TileStageArgs {
    const uint64_t* tape_ro;   /* tape pool, read side (parents' tapes; never written by this launch) */
    uint64_t* tape_wr;         /* same pool, write side (freshly claimed chunks) */
    unsigned long long* tape_index;
    long long pool_cap;
    int* image;                /* this level's filled image */
    int tps;                   /* tiles per side at this level */
    mpr_tile_node* tiles;
    int count;
    int nslots;                /* slots of the root tape (LDS slot file height) */
    int choice_cap;            /* min/max clauses recorded (<= 4096) */
    float z;                   /* 2-D: constant Z */
    float mat[16];             /* column-major 4x4 (3-D) or 3x3 (2-D, first 9) */
    unsigned long long* counters;
    GroupInfo* groups;         /* last tile stage, float pass in group form (else null): per-group record ...   */
    ulonglong2* choice_masks;  /* ... and decision masks, choice_cap entries per group                        */
    int debug;                 /* development only (MPR_DEBUG_TILES): 1 = skip tape pushing, 2 = skip the arithmetic */
    float* heat;               /* heatmap frames (render*_heatmap): S x S amortised work per pixel, else null */
    int heat_stride;           /* = image size in pixels */
    int* next_choices;         /* device counter (atomicMax): an upper bound on the min / max clauses of any tape this
                                * stage pushes; sizes the next stage's choice array */
    bool no_push;              /* last stage, float AND normals pass on the groups' tapes: write the groups' records only, push no tapes
                                * (with len_stats: a sixteenth of the groups still pushes, into chunks nobody reads, to measure) */
    bool vgpr_slots;           /* tapes with many slots: the assembly walk with the slot file in registers (MPR_TILES_VGPR=0: never) */
    bool compiled_walk;        /* development (MPR_TILES_ASM=0): the compiled forward / backward walks instead of the assembly ones */
    bool no_mask;              /* mpr_column_weights: no tile is culled by a fill (src/context.cu:299-305) — which tiles a stage leaves
                                * ambiguous then depends on the tape and the view alone, not on the order its wavefronts finish in */
    int measure_at[2], measure_len;   /* the sample len_stats is taken over: groups [measure_at[k], measure_at[k] + measure_len) */
    int* len_stats;            /* last stage with `groups`: [0] += clauses of the tapes handed on, [1] += clauses of the tapes
                                * walked x tiles handed on, over a sample of the groups (the float pass's form depends on it) */
    const uint32_t* gen_fwd = nullptr;   /* a launch whose tiles all walk the ROOT tape (a frame's first stage), tape of at most 24 slots and
                                          * 64 min / max clauses: its walks as generated code (tile_gen.hpp) — else null */
    const uint32_t* gen_bwd = nullptr;   /* (null with gen_fwd set: the assembly interpreter walks backward; development) */
    const uint32_t* gen_bwd_full = nullptr;   /* instead of gen_bwd: the tapes pushed here are shortened again by a stage that runs generated code
                                               * too, or this launch is such a stage (gen_parent): the walk that follows the PARENT's tape clause
                                               * by clause and records the clauses of the tape it writes (tile_gen.cpp) */
    int gen_words = 0, gen_nchoices = 0; /* words a walk of that tape visits (operations + end), its min / max clauses */
    bool gen_forward_only = false;       /* with gen_fwd and gen_decisions, a first stage whose tapes nobody will walk (context.hip: lean_first):
                                          * no backward walk, nothing pushed — a tile that decided anything leaves its record (what it
                                          * decided; as kept min / max clauses every undecided one, dead or alive: a superset of what its
                                          * shortened tape would keep) and a tape field that only says "there is a record" (1) */
    unsigned long long* self_info = nullptr;   /* with gen_fwd, the first stage of a frame that starts at the 16^3 tiles: per tile
                                                * SKIP0_INFO_U64 words — what it decided by itself (lhs, rhs: bit k = the root tape's k-th
                                                * min / max) and its verdict (SKIP0_*), for k_skip0_compare */
    const uint32_t* gen_fwd2 = nullptr;        /* round 5: the scheduled forward walk this stage runs (interval_gen.hpp: the kind the stage needs, loose or
                                                * exact) and the exact walk of the same kind a loose one falls back on (null: gen_fwd's code and harness) */
    const uint32_t* gen_fwd2_exact = nullptr;
    int lean = 0;                         /* run as k_eval_tiles<.., LEAN>: gen_fwd2 is loose code for 64 vector registers, the stage neither pushes nor
                                                * measures, six wavefronts per SIMD.  A wavefront whose walk asks for the exact code (or that belongs to the
                                                * stage's sample) only raises redo_flags[its workgroup]: a second launch (only_flagged) runs those */
    /* lean == 2: gen_fwd2 is TIGHT code (interval_gen.hpp; it names 80 vector registers: five wavefronts per SIMD) and this is the last tile
     * stage of a frame nobody reads: a tile the reference's enclosures leave ambiguous and the second, tight enclosure proves empty or filled
     * stays what it is for the records (the group's masks, which the float and normals passes read) but leaves the list the float pass walks —
     * filled: its height is drawn here.  Sound: the tight enclosure holds every value the float pass would compute for the tile's voxels on the
     * tape it would walk (the root tape with what was decided above imposed) */
    /* verdict_only (with lean == 2): a launch BEHIND a last stage that ran another way (a frame that leaves the reference's tiles and tapes:
     * exact enclosures, every tape pushed): the tight code once more over the stage's list, nothing written but tight_skip[tile] — 1: the
     * float pass need not walk it (provably empty), 2: it draws it without walking (provably filled), 0: as ever.  The lists, tapes and
     * records stay the reference's.  Sound where the loose walk raises no flag: there the exact decisions the tile's own tape carries are
     * facts about the float values, and the tight enclosure with the PARENT's decisions imposed holds them (a flagged wavefront: no verdict) */
    bool verdict_only = false;
    unsigned char* tight_skip = nullptr;
    int* tight_image = nullptr;                /* verdict_only: null, or an image of the stage's tiles (tps x tps, zero before the launch): max z index of the tiles
                                                * found filled per column — what hides the tiles behind them from the float pass's segments (the reference's own
                                                * images stay as they are) */
    unsigned char* redo_flags = nullptr;
    const unsigned char* only_flagged = nullptr;
    unsigned int* gen_redo_count = nullptr;    /* development: wavefronts whose loose walk asked for the exact one */
    /* k_eval_tiles<.., 93>, tiles on the ROOT tape (a first stage; frames nobody reads): the tape's loose forward walk with the choices
     * recorded for the interpreter's backward walk (interval_gen.hpp: IW_FIRST_MASKS); a wavefront in which a live tile asks for the
     * exact walk runs the interpreter's.  big_end: index of the end clause, big_nchoices: min / max clauses */
    const uint32_t* big_fwd = nullptr;
    int big_end = 0, big_nchoices = 0;
    const uint32_t* big_bwd = nullptr;         /* ... and the tape's generated backward walk for the tiles that push (tile_gen.hpp: tile_gen_build_big_backward) */
    bool gen_guarded = false;            /* with gen_parent: the walk that jumps over the runs the parents' decisions leave dead (interval_gen.hpp:
                                          * IW_BELOW_GUARDED; stages that push nothing) */
    bool gen_loose = false;              /* with gen_fwd, frames nobody reads: exp / log enclosures from the hardware's v_exp_f32 / v_log_f32, widened
                                          * by their error bound, instead of the correctly rounded ones (tile_gen_asm.hpp: TG_LOOSE_ROUTINES) */
    const unsigned long long* gen_parent = nullptr;   /* with gen_fwd, instead of gen_bwd: the launch is the stage BELOW the one that wrote these records
                                                       * (gen_decisions there): its tiles walk their parents' shortened tapes — as the root tape's
                                                       * generated code with the parent's decisions imposed, their own renumbered to that tape's */
    unsigned long long* gen_decisions = nullptr;   /* with gen_bwd, optional: per tile that pushes a tape, four words — the root tape's min / max
                                                    * clauses it decided for the lhs, for the rhs, those its tape keeps, 0 (the normals pass on
                                                    * the root tape's generated code applies them, kernels_normals_gen.hip) */
};

/* first tile stage, one workgroup per tile, level by level over the root tape's DAG
 * (kernels_wide.hip, tape_schedule.hpp) */
struct WideStageArgs {
    TileStageArgs t;
    const void* recs;          /* SchedRec[nclauses] in (level, opcode) order */
    const int* level_start;    /* nlevels + 1 offsets into recs */
    int nlevels;
    int nclauses;
    int root_val;              /* value index of the result (0..2 = X, Y, Z; 3 + i = clause i) */
    int root_tape;             /* pool index of the root tape's head */
    const uint32_t* bits_in;   /* later stages: two bits per root clause for the tape of every tile of the PREVIOUS stage */
    uint32_t* bits_out;        /* the same for the tapes this stage leaves (null: nobody will read them) */
    int wpt;                   /* words per tile in those tables: (nclauses + 15) / 16 */
    const uint16_t* prev_writer;   /* per clause: the previous clause with the same out slot (0xFFFF: none) */
    const uint32_t* defs;          /* per clause (tape order): value indices of its operands, pl | pr << 16 */
};

struct VoxelArgs {
    const uint64_t* tape_ro;
    int* image;                /* S x S output */
    int tps;                   /* smallest tiles per side (S/4 in 3-D, S/8 in 2-D) */
    const mpr_tile_node* tiles;
    int count;
    int nslots;
    float z;
    float mat[16];
    unsigned long long* counters;
    float* heat;               /* heatmap frames: S x S work per pixel, else null */
    bool vgpr_slots;           /* assembly interpreter, tapes with many slots: the slot file in registers (MPR_TILES_VGPR=0: never) */
};

struct NormalArgs {
    const uint64_t* tape_ro;
    const int* image;
    uint32_t* output;
    int size;
    int nslots;
    float mat[16];
    const mpr_tile_node* tiles;
    const mpr_tile_node* subtiles;
    const mpr_tile_node* microtiles;
    unsigned long long* counters;
    const int* col_list;       /* multi-GPU: the 64 x 64 columns this rank owns (null: all) ... */
    int ncols;                 /* ... and how many */
    /* last tile stage without tape pushing (TileStageArgs::no_push): the smallest tiles carry their group's tape, and a
     * tile's decisions are applied while that tape is walked (null: every tile carries its own tape) */
    const GroupInfo* groups;
    const ulonglong2* choice_masks;
    int choice_cap;
    bool vgpr_slots;           /* tapes with many slots: the slot file in registers (MPR_TILES_VGPR=0: never) */
    /* frames whose first stage ran the root tape's generated code over the 16^3 tiles and recorded its decisions
     * (TileStageArgs::gen_decisions), group form: every pixel on the ROOT tape's generated Deriv code (tile_gen.hpp) with the
     * decisions of its 16^3 and 4^3 tiles applied — one walk per footprint whatever tapes its pixels carry (else null) */
    const uint32_t* gen_code = nullptr;
    const uint32_t* gen_code_guarded = nullptr;           /* the same walk with its dead runs guarded (TileGen::deriv_guarded), or null */
    const unsigned long long* gen_decisions = nullptr;    /* the 16^3 tiles' records (GEN_RECORD_U64 words each) */
    const unsigned long long* gen_decisions0 = nullptr;   /* the 64^3 tiles', when that stage ran */
    const unsigned long long* gen_decisions2 = nullptr;   /* the 4^3 tiles', when the last stage pushed: everything decided for a pixel in one record
                                                           * (else: the 16^3 tile's record and the group's masks) */
    int gen_nchoices = 0;
    /* frames that start at the 16^3 tiles (context.hip: skip0, verified): what each 64^3 tile of the reference's first stage decides
     * (Skip0ParentsArgs::parents).  The reference's tape for a voxel inside an ambiguous 64^3 tile is that tile's or a deeper one:
     * its decisions are imposed on every pixel there — also where the 16^3 tile was culled and keeps no record of its own (round 5:
     * scripts/fuzz_sweep.py, seeds 1989 / 2074 / 2435 / 2762 / 2891: decisions that are not facts about the float values) */
    const unsigned long long* skip0_parents = nullptr;
};

/* children != null (3-D frames that start at the 16^3 tiles): also the 64 children of every first-stage tile, t0 = S / 64 */
/* ---- frames that start at the 16^3 tiles, verified (context.hip: skip0) ----
 * Such a frame lets every 16^3 tile decide by itself, on the root tape, what its 64^3 parent would have decided for it: the
 * reference's procedure wherever the interval routines are inclusion-isotone — and they are not everywhere (frame_domain.hpp: log's
 * zero bound, NaN ends that fmin / fmax drop; bear's far tiles take the former in every frame).  So the 64^3 tiles ARE walked: on a
 * second stream, beside the frame (64 wavefronts that nothing waits for until the float pass is launched), and a third kernel holds
 * every 16^3 tile against its parent: a parent the reference culls must have 64 children that culled themselves the same way, an
 * ambiguous parent's decisions must all have been made again by each child (then the child's walk of the root tape IS its walk of
 * the parent's shortened tape, value for value) — or the child must have culled itself the way it is culled on the parent's tape.
 * One violation and the frame is rendered again from the 64^3 tiles down. */
constexpr int SKIP0_INFO_U64 = 4;
enum { SKIP0_UNSEEN = 0, SKIP0_EMPTY = 1, SKIP0_FILLED = 2, SKIP0_AMBIGUOUS = 3 };   /* UNSEEN: dead, another rank's, or occluded before / while evaluated */
struct Skip0ParentsArgs {
    const uint64_t* tape_ro = nullptr;       /* the pool: [0] = the root tape's head */
    const uint32_t* gen_fwd2_first = nullptr, *gen_fwd2_below = nullptr;   /* the root tape's exact forward walks (interval_gen.hpp: IW_FIRST for the
                                              * parents, IW_BELOW for the children their decisions are imposed on) */
    unsigned long long* parents = nullptr;   /* [count][SKIP0_INFO_U64] */
    int count = 0, tps = 0;                  /* 64^3 tiles, per side */
    float mat[16] = {0};
};
void launch_skip0_parents(hipStream_t s, const Skip0ParentsArgs& a);
void launch_count_differences(hipStream_t s, const int* a, const int* b, size_t n, unsigned long long* out);
void launch_test_interval_gen(hipStream_t s, const uint32_t* code, int loose, int n, const float* a_lo, const float* a_hi, const float* b_lo, const float* b_hi,
                              float* out_lo, float* out_hi, int* choice, int* asks_exact);
void launch_test_tight_trig(hipStream_t s, const uint32_t* code, int is_sin, unsigned long long first, unsigned long long count, unsigned long long* out);
void launch_test_loose_gen(hipStream_t s, const uint32_t* code, int op, float imm, float other_lo, float other_hi, int x_is_rhs, unsigned long long first,
                           unsigned long long count, unsigned long long* out);
void launch_test_float_in_enclosure(hipStream_t s, int op, float imm, unsigned long long first, unsigned long long count, unsigned long long* out);
void launch_debug_walk_cycles(hipStream_t s, const uint32_t* code, const uint32_t* code_exact, int reps, long long* out, unsigned int* redone, int waves);
/* children: [64 a.count][SKIP0_INFO_U64], what the frame's first stage left (TileStageArgs::self_info); flag: host-coherent
 * memory, set to 1, never cleared.  A child that did not make a decision of its parent's again gets the reference's walk — the
 * parent's decisions imposed — and passes if it ends culled the way it culled itself. */
void launch_skip0_compare(hipStream_t s, const Skip0ParentsArgs& a, const unsigned long long* children, int* flag);
void launch_begin_frame(hipStream_t s, int* zero_base, size_t zero_words, unsigned long long* tape_index, int tape_len, int* num_active,
                        mpr_tile_node* tiles, int count, int cols, const int* owner, int rank, mpr_tile_node* children = nullptr, int t0 = 0);
/* levels: 3 = the three tile stages' images, 4 = + the heightmap / 2-D image, 5 = + the normals */
void launch_zero_owned(hipStream_t s, int* arena, int levels, int S, const int* owner, int rank);
bool zsort_supported(int tps);
void launch_compact_zsorted(hipStream_t s, bool last, mpr_tile_node* tiles, int count, int tps, const int* image,
                            mpr_tile_node* out, int* hist, int* cursor, int* pub, int seq, int* next_image, int next_size,
                            int* need, unsigned char* group_alive = nullptr, const unsigned long long* tape_index = nullptr,
                            int* source_out = nullptr, int* clear = nullptr, int nclear = 0);     /* clear: nclear ints the scan zeroes on the way */
void launch_list_alive_groups(hipStream_t s, const unsigned char* alive, int ngroups, int* list);
void launch_mask_filled(hipStream_t s, mpr_tile_node* tiles, int count, int tps, const int* image);
size_t tile_stage_lds_bytes(int nslots, int choice_cap);
/* returns whether the launch ran the root tape's generated code (a.gen_fwd given and tile_stage_gen_possible) */
bool launch_eval_tiles(hipStream_t s, int dim, const TileStageArgs& a);
bool tile_stage_gen_possible(int nslots, long long pool_cap, bool compiled_walk, bool vgpr_slots, int debug);
bool tile_stage_big_possible(int nslots, int choice_cap, long long pool_cap, bool compiled_walk, bool vgpr_slots, int debug);
/* host-generated code (tile_gen.hpp) into executable memory: copied by a kernel, then every CU drops its instruction cache */
void launch_install_code(hipStream_t s, uint32_t* exec_dst, const uint32_t* src, size_t dwords, int cus);
bool wide_stage_fits(int nclauses);
size_t wide_stage_lds_bytes(int nclauses);
void launch_eval_tiles_wide(hipStream_t s, int dim, const WideStageArgs& w, int threads_forced = 0);
void launch_compact_subdivide(hipStream_t s, int dim, bool last, mpr_tile_node* tiles, int count, int tps,
                              const int* image, int* num_active, mpr_tile_node* out,
                              int* pub, int seq, int* next_image, int next_size, unsigned char* group_alive = nullptr,
                              const unsigned long long* tape_index = nullptr, int* source_out = nullptr);
/* source_out (last stage): per tile of the list handed on, its index in THIS stage's list (group = index / 64, child = index % 64) */
void launch_copy_filled(hipStream_t s, int dim, const int* prev, int* image, int size);
size_t voxel_lds_bytes(int nslots);
void launch_eval_voxels(hipStream_t s, int dim, const VoxelArgs& a);
/* same pass, interpreter in gfx950 assembly (kernels_voxel_asm.hip); no counters */
void launch_eval_voxels_asm(hipStream_t s, int dim, const VoxelArgs& a);

void launch_test_sqrt_all(hipStream_t s, unsigned long long first, unsigned long long count, unsigned long long* out);
void launch_test_float_asm(hipStream_t s, const uint64_t* tape3, int n, const float* a, const float* b, float* out);
/* same pass, every tape translated to machine code on the device (kernels_voxel_jit.hip); no counters.
 * code: executable device memory, `grid` regions of region_dwords each */
size_t jit_code_dwords(const uint64_t* clauses, int n, bool group);
int jit_slot_class(int nslots);
int jit_max_choices();        /* recorded min / max decisions per tape the group form of the generated code takes */                      /* 24 / 40 / 96 / 192, or 0: too many slots for registers */
int jit_grid(int dim, int nslots, int cus, bool group);
/* groups != null: the group form over the last tile stage's list (a.tiles / a.count), else one wavefront per smallest tile */
void launch_eval_voxels_jit(hipStream_t s, int dim, const VoxelArgs& a, uint32_t* code, uint32_t region_dwords, int slot_dwords, int slots, int grid,
                            int tape_len, const GroupInfo* groups, const ulonglong2* choice_masks, int choice_cap, int* group_counter,
                            const int* group_list, bool always_invalidate = false);
/* same pass on the ROOT tape's host-generated code (voxel_gen.hpp) with the tiles' recorded decisions: a.tiles / a.count = the smallest
 * tiles, source[t] = tile t's index in the last tile stage's list (launch_compact_*: source_out), groups / masks of that stage,
 * parent_records: the records of the tiles of the stage above it; tile_counter: voxel_gen_counter_ints() zeroed ints */
int voxel_gen_grid(int dim, int cus);
int voxel_gen_counter_ints();
/* ... by FOOTPRINT SEGMENTS (round 6; kernels.hip: k_compact_footprints makes them of the last tile stage's list, in place of the z-sorted
 * list of tiles; kernels_voxel_jit.hip: k_eval_voxels_gen_fp): items: one word per segment (capacity: a quarter of the stage's tiles);
 * meta: 2 ints, zero before the first use; a.tiles = the LAST TILE STAGE's list; counter: voxel_gen_counter_ints() ints the compaction clears */
int voxel_gen_fp_grid(int cus);
void launch_compact_footprints(hipStream_t s, mpr_tile_node* tiles, int count, int tps, const int* image, int* num_active, unsigned* items, int* meta,
                               int* clear, int nclear, int cstride, int* pub, int seq, int* next_image, int next_size, const unsigned long long* tape_index);
/* the segments alone, of a list another compaction has been through (skip: one byte per tile, 1 = no segment for it; never null — a frame
 * without a second verdict hands a zeroed array); meta: 4 ints; tight_image (TileStageArgs::tight_image; may be null): tiles behind its
 * entries make no segment either, and the entries are drawn into heights (size x size pixels) */
void launch_footprint_segments(hipStream_t s, mpr_tile_node* tiles, int count, int tps, const int* image, unsigned* items, int* meta, int* clear, int nclear, int cstride,
                               const unsigned char* skip, const int* tight_image, int* heights, int size);
int voxel_gen_counter_lists();      /* the float pass's work counters: this many, voxel_gen_counter_ints() / this apart */
void launch_eval_voxels_gen_fp(hipStream_t s, const VoxelArgs& a, const uint32_t* code, int grid, const unsigned* items, const int* meta, const GroupInfo* groups,
                               const ulonglong2* choice_masks, int choice_cap, int* counter, const unsigned long long* parent_records, int nchoices, int run,
                               int* walked, const unsigned char* skip);
void launch_eval_voxels_gen(hipStream_t s, int dim, const VoxelArgs& a, const uint32_t* code, int grid, const int* source, const GroupInfo* groups,
                            const ulonglong2* choice_masks, int choice_cap, int* tile_counter, const unsigned long long* parent_records,
                            int nchoices, int run = 0, int* walked = nullptr, const unsigned char* skip = nullptr);
void launch_test_float_gen_all(hipStream_t s, const uint32_t* code, int op, float imm, unsigned long long first, unsigned long long count, unsigned long long* out);
void launch_test_float_gen(hipStream_t s, const uint32_t* code, int n, const float* a, const float* b, float* out, unsigned long long dl,
                           unsigned long long dr);
void launch_test_float_jit(hipStream_t s, const uint64_t* tape3, uint32_t* code, uint32_t region_dwords, int n, const float* a,
                           const float* b, float* out);
size_t normals_lds_bytes(int nslots);
void launch_eval_normals(hipStream_t s, const NormalArgs& a);
void launch_eval_normals_asm(hipStream_t s, const NormalArgs& a);   /* kernels_normals_asm.hip; no counters */
void launch_pack(hipStream_t s, const int* heights, const uint32_t* normals, int S, const int* col_list,
                 int ncols, int capacity, int with_normals, int* out);
void launch_pack_planned(hipStream_t s, const int* heights, const uint32_t* normals, int S, const int* owner, const int* slot,
                         int rank, int capacity, int with_normals, int* out);
void launch_unpack_planned(hipStream_t s, int* heights, uint32_t* normals, int S, const int* owner, const int* slot, int rank,
                           int capacity, int with_normals, const int* in_all);
void launch_unpack(hipStream_t s, int* heights, uint32_t* normals, int S, const int* col_list, int ncols,
                   int capacity, int with_normals, const int* in);
void launch_test_interval_asm(hipStream_t s, const uint64_t* tape, int n, const float* a_lo, const float* a_hi,
                              const float* b_lo, const float* b_hi, float* out_lo, float* out_hi, int* choice);
void launch_debug_interp_cycles(hipStream_t s, const uint64_t* tape, int reps, long long* out, int waves);
void launch_test_interval(hipStream_t s, int op, int n, const float* a_lo, const float* a_hi, const float* b_lo,
                          const float* b_hi, float imm, float* out_lo, float* out_hi, int* choice);
void launch_test_float(hipStream_t s, int op, int n, const float* a, const float* b, float imm, float* out);
void launch_test_deriv(hipStream_t s, int op, int n, const float* a, const float* b, float imm, float* out);

/* mpr::Effects (kernels_effects.hip) */
size_t effect_tables_bytes();
void launch_draw_ssao(hipStream_t s, const int32_t* depth, const uint32_t* norm, const void* tables, int S, int32_t* out);
void launch_blur_ssao(hipStream_t s, const int32_t* image, const int32_t* ssao, int S, int32_t* out);
void launch_draw_shaded(hipStream_t s, const int32_t* depth, const uint32_t* norm, const int32_t* ssao, int S, int32_t* out);

}  // namespace mprk
