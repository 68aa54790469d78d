/*
 * mpr_amd.h — C ABI of the MI355X-native mpr hot path (libmpr_amd.so).
 *
 * The reference (mkeeter/mpr) has no FFI layer: callers use two C++ structs directly,
 *     mpr::Tape(const libfive::Tree&)                         inc/tape.hpp:24-30
 *     mpr::Context(int32_t image_size_px)                     inc/context.hpp:38-39
 *     Context::render2D(tape, Eigen::Matrix3f, float z = 0)   inc/context.hpp:41-42
 *     Context::render3D(tape, Eigen::Matrix4f)                inc/context.hpp:40
 *     Context::render2D_brute(...)                            inc/context.hpp:47-49
 * and then read public members from the host (managed memory): stages[i].filled,
 * stages[i].tiles, stages[i].tile_array_size, normals, tape_data, *tape_index
 * (benchmark/render_2d_table.cpp:57-62, render_3d_table.cpp:59-69, circle.cpp:42-103,
 * tape_shortening.cpp:56-72, render_3d_heatmap.cpp:64).
 *
 * This header is the seam a binding would use instead: plain pointers and sizes, opaque
 * handles, integer status codes (0 = ok) with mpr_last_error() for the message; nothing
 * here exits the process (the reference's CUDA_CHECK does, inc/util.hpp:19-25).
 * include/mpr.hpp re-creates mpr::Tape / mpr::Context with the reference's member names on
 * top of it.  Matrices are column-major like Eigen (element (row,col) at [row + col*N]).
 *
 * Threading: one context = one device + one HIP stream; calls on one context must be
 * serialised by the caller (the reference's Context is not re-entrant either); different
 * contexts may be used concurrently.
 */
#ifndef MPR_AMD_H
#define MPR_AMD_H

#include <stddef.h>
#include <stdint.h>

#include "mpr_clause.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MPR_OK 0
#define MPR_ERR_INVALID 1      /* bad argument */
#define MPR_ERR_PARSE 2        /* malformed .frep */
#define MPR_ERR_NO_DEVICE 3    /* no usable HIP device / HIP runtime error (see message) */
#define MPR_ERR_ALLOC 4
#define MPR_ERR_UNSUPPORTED 5

typedef struct mpr_tree mpr_tree;        /* expression DAG (stand-in for libfive::Tree) */
typedef struct mpr_tape mpr_tape;        /* flat clause tape (mpr::Tape) */
typedef struct mpr_context mpr_context;  /* render context (mpr::Context) */

/* Thread-local description of the last failing call. */
const char* mpr_last_error(void);
/* Library version string. */
const char* mpr_version(void);

/* ---- expression front end (host only; replaces the libfive::Tree the reference takes at
 *      src/tape.cpp:21; opcodes are libfive's packed numbering, see mpr_tree_op below) ---- */
enum mpr_tree_op {
    MPR_T_CONSTANT = 1, MPR_T_VAR_X = 2, MPR_T_VAR_Y = 3, MPR_T_VAR_Z = 4,
    MPR_T_SQUARE = 7, MPR_T_SQRT = 8, MPR_T_NEG = 9, MPR_T_SIN = 10, MPR_T_COS = 11,
    MPR_T_ASIN = 13, MPR_T_ACOS = 14, MPR_T_ATAN = 15, MPR_T_EXP = 16, MPR_T_ABS = 17,
    MPR_T_LOG = 18,
    MPR_T_ADD = 20, MPR_T_MUL = 21, MPR_T_MIN = 22, MPR_T_MAX = 23, MPR_T_SUB = 24,
    MPR_T_DIV = 25
};
int mpr_tree_x(mpr_tree** out);
int mpr_tree_y(mpr_tree** out);
int mpr_tree_z(mpr_tree** out);
int mpr_tree_const(float v, mpr_tree** out);
int mpr_tree_unary(int op, const mpr_tree* a, mpr_tree** out);
int mpr_tree_binary(int op, const mpr_tree* a, const mpr_tree* b, mpr_tree** out);
/* libfive::Tree::remap (benchmark/render_effects.cpp:36) */
int mpr_tree_remap(const mpr_tree* t, const mpr_tree* x, const mpr_tree* y, const mpr_tree* z,
                   mpr_tree** out);
/* libfive::Archive::deserialize(...).shapes.front().tree (benchmark/render_2d_table.cpp:34) */
int mpr_tree_from_frep(const void* bytes, size_t n, mpr_tree** out);
int mpr_tree_from_frep_file(const char* path, mpr_tree** out);
/* libfive::Archive::serialize (gui/main.cpp:394-403); returns needed size in *n */
int mpr_tree_to_frep(const mpr_tree* t, void* bytes, size_t cap, size_t* n);
int mpr_tree_size(const mpr_tree* t, size_t* nodes);
void mpr_tree_free(mpr_tree* t);

/* ---- tape (mpr::Tape, inc/tape.hpp:24-30, src/tape.cpp:21-228); host-side object, a
 *      context uploads it on first use ---- */
int mpr_tape_from_tree(const mpr_tree* t, mpr_tape** out);
/* adopt an already-flattened tape (head clause, clauses, end clause) */
/* The tape's dependency levels (operands renamed to the clause that produced them; clauses of one
 * level are independent): what the level-parallel first tile stage walks.  levels: one int per body
 * clause (length - 2), or null. */
int mpr_tape_schedule_info(const mpr_tape* tape, int32_t* nlevels, int32_t* max_width, int32_t* levels);
int mpr_tape_from_clauses(const uint64_t* clauses, int32_t length, mpr_tape** out);
int32_t mpr_tape_length(const mpr_tape* t);            /* mpr::Tape::length */
const uint64_t* mpr_tape_data(const mpr_tape* t);      /* mpr::Tape::data (host copy) */
int32_t mpr_tape_num_slots(const mpr_tape* t);         /* highest slot index + 1 */
int32_t mpr_tape_num_choices(const mpr_tape* t);       /* number of min/max clauses */
int32_t mpr_tape_flags(const mpr_tape* t);             /* bit1: unsupported opcodes (src/tape.cpp:182-196).  Bit0 is never set:
                                                          where the reference prints "Ran out of slots!" and renders with slot 0
                                                          (:79-81), mpr_tape_from_tree fails with MPR_ERR_UNSUPPORTED */
/* Does every interval operation of the tape stay, over the WHOLE view of a frame with this matrix (column-major, (dim + 1)^2; z: the
 * 2-D frame's), where the reference's interval routines are inclusion-isotone — operands finite and inside the function's domain
 * (inc/gpu_interval.hpp: log's zero bound :382-390, the NaN ends of asin / acos :306-324, a divisor that holds zero :162-190 are
 * where they are not)?  1 / 0.  A 3-D frame nobody reads starts at the 16^3 tiles either way: unverified when the view is tame,
 * verified against the 64^3 tiles it skips when it is not (csrc/frame_domain.hpp).  trace: null, or 2 * length doubles for the
 * enclosure of every clause's result (tests). */
int mpr_tape_frame_is_tame(const mpr_tape* t, int dim, const float* mat, float z, double* trace);
void mpr_tape_free(mpr_tape* t);

/* ---- context (mpr::Context, inc/context.hpp:38-73, src/context.cpp:17-49) ---- */
typedef struct mpr_ctx_options {
    int32_t device;            /* HIP device ordinal */
    int32_t image_size_px;     /* S; must be a multiple of 64 */
    int64_t pool_clauses;      /* tape pool capacity in clauses (the reference's NUM_SUBTAPES * 64,
                                  inc/parameters.hpp:14-22).  A pool that runs out is not an error: the tiles
                                  concerned keep their parents' tapes (src/context.cu:336-347) and
                                  mpr_counters::pool_overflowed is set.  0 = sized by the context: a few M clauses
                                  to begin with, doubled — and the frame rendered again — whenever a frame's
                                  pushes do not fit, up to the BIG_SERVER size (3.28 GB) */
    int32_t flags;             /* MPR_CTX_* */
} mpr_ctx_options;
#define MPR_CTX_TIMING 1       /* record HIP events around every kernel (mpr_get_timings) */
#define MPR_CTX_COUNTERS 2     /* accumulate the work counters of mpr_get_counters on the device
                                  (costs a few atomics per wave: keep off when timing) */
#define MPR_CTX_TIMING_FLOAT 8  /* HIP events around the float pass only (eval_voxels_f: the dominant kernel): two events per frame
                                  instead of sixteen, so that a timed loop can carry them at no measurable cost */
#define MPR_CTX_SERIAL_STAGES 4 /* tile stages always by the 64-tiles-per-wavefront kernels, never level-parallel: with
                                  MPR_CTX_COUNTERS the clause counters are then the implementation-independent figures of
                                  SURVEY.md 8(d) (F / R per group of 64 list entries; the level-parallel kernel has no such
                                  groups and reports its own clause counts) */

#define MPR_CTX_PARANOID 16    /* render2D / render3D render every frame that took a shortcut (a start at the 16^3 tiles, loose
                                  enclosures, a last tile stage that pushes no tapes, generated code) a second time the
                                  reference's way — every stage from the 64 px tiles down, the reference's enclosures, every tape
                                  pushed and walked — and compare heights and normals of the two on the device
                                  (mpr_ctx_paranoid_stats).  Twice the time and more: for tests and for callers that want the
                                  equivalence argument of csrc/context.hip checked frame by frame rather than trusted */

int mpr_ctx_create(int32_t device, int32_t image_size_px, mpr_context** out);
int mpr_ctx_create_ex(const mpr_ctx_options* opt, mpr_context** out);
/* frees everything the context holds except its two HIP streams, which stay with the process for its next context on the device
 * (the HIP runtime's stream destruction is not safe against its own signal handlers: csrc/context.hip, acquire_stream) */
void mpr_ctx_destroy(mpr_context* ctx);
int32_t mpr_ctx_image_size(const mpr_context* ctx);    /* Context::image_size_px */
/* device memory the context holds right now: images, tile lists, the tape pool (which starts small and doubles when a frame's
 * pushes do not fit, unless mpr_ctx_options::pool_clauses names a capacity), the float pass's code regions and records */
int64_t mpr_ctx_resident_bytes(const mpr_context* ctx);
/* MPR_CTX_PARANOID: out = {frames rendered, frames rendered a second time the reference's way, cells (heights + normals) in which
 * the two renderings of a frame differed — 0, or the shortcuts are not the reference's procedure for that tape and view} */
int mpr_ctx_paranoid_stats(const mpr_context* ctx, int64_t out[3]);

/* Context::render2D (src/context.cu:1136-1280).  Blocking: returns after the device has
 * finished, like the reference's cudaDeviceSynchronize at :1279. */
int mpr_render2d(mpr_context* ctx, const mpr_tape* tape, const float mat3_colmajor[9], float z);
/* Context::render3D (src/context.cu:1282-1458). */
int mpr_render3d(mpr_context* ctx, const mpr_tape* tape, const float mat4_colmajor[16]);
/* Context::render2D_brute (src/context.cu:1461-1508). */
int mpr_render2d_brute(mpr_context* ctx, const mpr_tape* tape, const float mat3_colmajor[9],
                       float z);
/* Context::render2D_heatmap / render3D_heatmap (inc/context.hpp:51-58; src/context.cu:1984-2146,
 * :2148-2339): a normal frame (image / heightmap / normals are produced as usual) that also returns
 * the amortised work per pixel: every tile adds the tape words it walked, divided by its area in
 * pixels, to the pixels it covers, the float pass adds its walk per pixel, and the total is divided
 * by the root tape's clause count.  heat_out: image_size^2 floats on the host, x + y * image_size.
 * 2-D heatmaps are reproducible bit for bit; in 3-D the order of the float additions, and the work
 * of tiles culled by a concurrent tile's fill, depend on timing (as upstream). */
int mpr_render2d_heatmap(mpr_context* ctx, const mpr_tape* tape, const float mat3_colmajor[9],
                         float z, float* heat_out);
int mpr_render3d_heatmap(mpr_context* ctx, const mpr_tape* tape, const float mat4_colmajor[16],
                         float* heat_out);
/* Non-blocking forms: enqueue the frame on the context's stream and return. */
int mpr_render2d_async(mpr_context* ctx, const mpr_tape* tape, const float mat3_colmajor[9],
                       float z);
int mpr_render3d_async(mpr_context* ctx, const mpr_tape* tape, const float mat4_colmajor[16]);
int mpr_ctx_sync(mpr_context* ctx);

/* Multi-GPU: render only the top-level xy columns owned by `rank`.
 * `owner` has (S/64)^2 entries (column index = x + y*(S/64)), each the owning rank; columns
 * of other ranks are skipped at stage 0 and their pixels stay 0. */
/* Work proxy for the column deal: weights[(S/64)^2] = first-stage tiles per 64 x 64 column that the interval
 * evaluation of this tape and view leaves ambiguous.  Runs the 64 px stage only; identical on every rank.  The call replaces
 * the context's previous frame: the images and normals are cleared, stages[0].tiles is this call's list, the other lists are
 * empty, until the next frame is rendered. */
int mpr_column_weights(mpr_context* ctx, const mpr_tape* tape, int32_t dim, const float* mat_colmajor, float z, float* weights);
int mpr_render3d_part(mpr_context* ctx, const mpr_tape* tape, const float mat4_colmajor[16],
                      const int32_t* owner, int32_t rank);
int mpr_render2d_part(mpr_context* ctx, const mpr_tape* tape, const float mat3_colmajor[9],
                      float z, const int32_t* owner, int32_t rank);
/* the same, not waiting for the frame (mpr_ctx_sync, or any blocking call, does) */
int mpr_render3d_part_async(mpr_context* ctx, const mpr_tape* tape, const float mat4_colmajor[16],
                            const int32_t* owner, int32_t rank);
int mpr_render2d_part_async(mpr_context* ctx, const mpr_tape* tape, const float mat3_colmajor[9], float z,
                            const int32_t* owner, int32_t rank);
/* Gather with a resident plan, for the steady state of a multi-GPU loop: mpr_gather_plan uploads
 * the ownership table once (and derives every column's position inside its owner's pack); then
 * per frame mpr_render3d_part_async -> mpr_pack_planned_async -> the caller's all-gather, ordered
 * after the context's stream (mpr_ctx_stream) -> mpr_unpack_planned_async (one launch for all
 * foreign columns; rank r's pack at dev_in_all + r * capacity_cols * 4096 * (with_normals ? 2 : 1)
 * ints) -> mpr_ctx_sync.  Nothing is uploaded and the host never waits in between. */
int mpr_gather_plan(mpr_context* ctx, const int32_t* owner, int32_t rank, int32_t world, int32_t capacity_cols,
                    int32_t with_normals);
int mpr_pack_planned_async(mpr_context* ctx, void* dev_out);
int mpr_unpack_planned_async(mpr_context* ctx, const void* dev_in_all);
/* Deterministic column -> rank deal (identical on every rank).  weights may be NULL
 * (round-robin) or (S/64)^2 non-negative work estimates (longest-processing-time first). */
int mpr_partition_columns(int32_t columns, const float* weights, int32_t nranks, int32_t* owner);
/* Pack the 64x64 blocks of the columns owned by `rank` (heights, then normals when
 * with_normals) into a device buffer, `capacity_cols` blocks each; and scatter such a pack
 * (from any rank) into this context's full-size images.  Used around the one RCCL
 * all-gather per frame. */
int mpr_pack_columns(mpr_context* ctx, const int32_t* owner, int32_t rank, int32_t capacity_cols,
                     int32_t with_normals, void* dev_out);
int mpr_unpack_columns(mpr_context* ctx, const int32_t* owner, int32_t rank, int32_t capacity_cols,
                       int32_t with_normals, const void* dev_in);

/* ---- results (what callers read from Context's public members) ---- */
/* stages[stage].filled -> host; (S / {64,16,4,1}[stage])^2 int32, index x + y*side */
int mpr_read_filled(mpr_context* ctx, int32_t stage, int32_t* host);
/* normals -> host; S*S uint32 0xFF<<24 | nz<<16 | ny<<8 | nx */
int mpr_read_normals(mpr_context* ctx, uint32_t* host);
/* stages[stage].tiles -> host; *n = number of valid entries from the last render
 * (stage 0: all top-level tiles).  cap in entries.  (Renders the last frame again the reference's way first when it was an
 * ordinary one: see mpr_ctx_last_stage_pushed.  So do mpr_read_tape_pool and mpr_get_counters.) */
int mpr_read_tiles(mpr_context* ctx, int32_t stage, mpr_tile_node* host, size_t cap, size_t* n);
/* Frames whose shortcut past the 64^3 tiles failed its verification and that were rendered again from those tiles down (csrc/kernels.hpp:
 * launch_skip0_parents; the tape's next 64 frames then start at the 64^3 tiles by themselves, twice as many after every further failure).  Tests. */
int64_t mpr_ctx_skip0_vetoes(const mpr_context* c);
/* tape_data / *tape_index -> host; copies min(cap, *tape_index) clauses */
int mpr_read_tape_pool(mpr_context* ctx, uint64_t* host, size_t cap, int32_t* tape_index);
/* device pointers (for zero-copy consumers such as the multi-GPU gather) */
int32_t* mpr_dev_filled(mpr_context* ctx, int32_t stage);
uint32_t* mpr_dev_normals(mpr_context* ctx);
void* mpr_ctx_stream(mpr_context* ctx);   /* hipStream_t */

/* Work counters of the last frame; all per frame.  tiles_* and voxel_tiles
 * are always filled; the clause counters need MPR_CTX_COUNTERS. */
typedef struct mpr_counters {
    int64_t tiles_in[3];        /* tiles evaluated per tile stage (3-D: 64/16/4 px; 2-D: 64/8 px in [0],[1]) */
    int64_t tiles_active[3];    /* tiles surviving each stage (ambiguous and not masked) */
    int64_t voxel_tiles;        /* smallest tiles handed to the float pass */
    int64_t clauses_fwd;        /* F: clause visits by wave-groups, forward (all passes) */
    int64_t clauses_bwd;        /* R: clause visits by wave-groups, backward (tape push) */
    int64_t clauses_written;    /* W: words written to the pool by tape pushes */
    int64_t clauses_fwd_voxels; /*    part of F spent in the float voxel/pixel pass */
    int64_t clauses_fwd_normals;/*    part of F spent in the normals pass */
    int64_t lane_clauses;       /* words visited forward, terminator excluded, per tile / visible voxel / pixel */
    int64_t normal_pixels;      /* pixels evaluated by the normals pass */
    int32_t tape_index;         /* pool words in use after the frame */
    int32_t pool_overflowed;    /* a push ran out of pool and fell back (src/context.cu:336-347) */
    int32_t slots_exceeded;     /* tape uses more slots than the evaluators hold */
    int32_t reserved;
} mpr_counters;
int mpr_get_counters(mpr_context* ctx, mpr_counters* out);

/* Per-kernel device time of the last frame, HIP events on the context's stream (only with
 * MPR_CTX_TIMING).  names[i] is a static string; returns count in *n (<= cap). */
int mpr_get_timings(mpr_context* ctx, const char** names, float* ms, int32_t cap, int32_t* n);

/* Name of the kernel the last frame's float pass (eval_voxels_f) ran as, as rocprofv3 prints it without
 * the namespace: "k_eval_voxels_jit_groups<3, 24>" (generated code, one translation per 64 sibling tiles),
 * "k_eval_voxels_jit<2, 24>" (generated code per tile), "k_eval_voxels_asm<3>" (assembly interpreter) or
 * "k_eval_voxels<3>" (C++ interpreter: instrumented frames).  Owned by the context; "" before a frame. */
const char* mpr_ctx_float_kernel(const mpr_context* ctx);
/* ... and its normals pass: "k_eval_normals_gen" (the root tape's generated code), "k_eval_normals_asm", "k_eval_normals_q" */
const char* mpr_ctx_normals_kernel(const mpr_context* ctx);
/* ... and the form each of its tile stages took (the last frame that ran tile stages: a reader's re-render counts), e.g.
 * "1:gen+bwd+records 2:gen/parent+guards" or "0:wide 1:interp 2:interp": <stage>:<wide | interp | gen[/parent][+guards][+bwd | +bwd_full]
 * [+records] | none>.  gen = the root tape's host-generated interval code (csrc/tile_gen.hpp), /parent = with the parent tiles' recorded
 * decisions imposed, +guards = jumping over what they left dead, +bwd / +bwd_full = tapes pushed by generated code too.  Tests assert
 * the path they mean to exercise with it.  Owned by the context. */
const char* mpr_ctx_tile_stage_forms(const mpr_context* ctx);
/* Tiles of the last frame as it ran (no re-render; mpr_get_counters gives the reference's): out[0..2] tiles evaluated per stage,
 * out[3..5] tiles left ambiguous, out[6] smallest tiles handed to the float pass.  A frame nobody reads may start at the 16^3
 * tiles and cull with sound but looser bounds than the reference: the same heights and normals from a few more tiles. */
int mpr_ctx_frame_tiles(mpr_context* ctx, int64_t out[7]);

/* 1 when the last frame's last tile stage pushed per-tile tapes (the reference's state), 0 when it did not need to (its own
 * sample of the tapes it would push said that float and normals pass do as well on the tapes it walked: DESIGN.md 3).
 *
 * render* leaves the reference's IMAGES — heights / occupancy, normals — always.  The reference's tile lists, tapes and
 * tape_index it leaves only when they are asked for: ordinary frames skip what only a reader needs (the last stage's per-tile
 * tapes; the 64^3 stage of 3-D frames up to 1024^3 whose tapes have narrow DAGs).  mpr_read_tiles, mpr_read_tape_pool AND
 * mpr_get_counters therefore first render the last frame again the reference's way (same tape — the context keeps a copy —,
 * view and partition): the images come out identical, the columns other ranks sent into a partitioned context stay in place,
 * mpr_get_timings afterwards describes that extra frame.  MPR_LAST_STAGE_PUSH=1 in the environment when the context is created
 * makes every frame the reference's way. */
int32_t mpr_ctx_last_stage_pushed(const mpr_context* ctx);

/* ---- compiled-expression baseline (reference benchmark/dump_tape.cpp + benchmark/brute.cu): the
 *      tape as straight-line HIP source, compiled for the device at run time (hiprtc), evaluated
 *      for every pixel without hierarchy.  Same image as mpr_render2d_brute. ---- */
typedef struct mpr_compiled mpr_compiled;
int mpr_compiled_create(int32_t device, const mpr_tape* tape, mpr_compiled** out);
void mpr_compiled_destroy(mpr_compiled* k);
const char* mpr_compiled_source(const mpr_compiled* k);
/* dev_image: size * size int32 on the device (e.g. mpr_dev_filled(ctx, 3)); blocking */
int mpr_compiled_render2d(mpr_compiled* k, int32_t size, const float mat3_colmajor[9], float z, int32_t* dev_image);

/* ---- mpr::Effects (reference inc/effects.hpp:21-37, src/effects.cu): image-space passes over the
 *      heightmap and normals of the context's last render3D ---- */
typedef struct mpr_effects mpr_effects;
/* Effects::Effects(): builds the SSAO sample kernel and noise vectors (mpr_effects_tables.h) */
int mpr_effects_create(int32_t device, mpr_effects** out);
void mpr_effects_destroy(mpr_effects* fx);
/* Effects::drawSSAO(ctx): image = blurred ambient occlusion, 0..255 per covered pixel */
int mpr_effects_draw_ssao(mpr_effects* fx, mpr_context* ctx);
/* Effects::drawShaded(ctx): image = 0xFFcccccc grey shading (one light, SSAO-dimmed, ambient) */
int mpr_effects_draw_shaded(mpr_effects* fx, mpr_context* ctx);
/* Effects::image / Effects::tmp: S*S int32 each; host copies, or the device pointers */
int mpr_effects_read_image(mpr_effects* fx, int32_t* host);
int mpr_effects_read_tmp(mpr_effects* fx, int32_t* host);
int32_t* mpr_effects_dev_image(mpr_effects* fx);
/* the tables (64 x 3 and 256 x 3 floats, row-major), for inspection */
int mpr_effects_tables_get(const mpr_effects* fx, float* kernel, float* rvecs);

/* (the hooks the parity suite uses to run single primitives and code generators — mpr_test_*, mpr_debug_* — are declared in
 * mpr_amd_test.h: they are not part of the boundary a caller of the reference binds) */

#ifdef __cplusplus
}
#endif
#endif
