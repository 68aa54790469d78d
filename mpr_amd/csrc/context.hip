/*
 * context.hip — render context and frame drivers (device side of the C ABI).
 *
 * Mirrors the role of the reference's mpr::Context (inc/context.hpp:38-73,
 * src/context.cpp:17-49) and of Context::render2D / render3D / render2D_brute
 * (src/context.cu:1136-1508): same stage order, same buffers, same results; different
 * launch structure (see kernels.hip).  Device memory is plain hipMalloc (no managed
 * memory); results are copied out on request.
 */
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/mpr_amd.h"
#include "../../include/mpr_amd_test.h"
#include "../../include/mpr_effects_tables.h"
#include "internal.hpp"
#include "kernels.hpp"
#include "tile_gen.hpp"
#include "interval_gen.hpp"
#include "voxel_gen.hpp"
#include "frame_domain.hpp"
static_assert(mpr::TILE_GEN_RECORD_U64 == mprk::GEN_RECORD_U64 && mpr::TILE_GEN_PRESENCE_WORDS == mprk::GEN_PRESENCE_WORDS, "one record layout");

namespace {

struct KernelTiming {
    const char* name;
    hipEvent_t start, stop;
};

}  // namespace

struct mpr_context {
    int device = 0;
    int S = 0;
    int flags = 0;
    hipStream_t stream = nullptr;

    int* filled[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t filled_n[4] = {0, 0, 0, 0};
    uint32_t* normals = nullptr;

    uint64_t* pool = nullptr;          /* Context::tape_data */
    long long pool_cap = 0;
    bool pool_auto = false;            /* the caller named no capacity: the pool starts small and doubles (up to the reference's BIG_SERVER
                                          size) whenever a frame's pushes do not fit — that frame is rendered again, so what a reader
                                          sees never depends on where the pool happened to stand */
    unsigned long long* tape_index = nullptr;   /* Context::tape_index; 64 bits on the device so that failed claims of
                                                 * concurrent waves can never wrap it (reported clamped to int32) */
    int* num_active = nullptr;         /* Context::num_active_tiles */
    unsigned long long* counters = nullptr;

    mpr_tile_node* tiles[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t tiles_cap[4] = {0, 0, 0, 0};
    size_t tiles_n[4] = {0, 0, 0, 0};


    int* owner_dev = nullptr;          /* column ownership, (S/64)^2 */
    std::vector<int32_t> owner_host;   /* what owner_dev holds (uploads are skipped when unchanged) */
    uint64_t owner_gen = 0;            /* bumped whenever owner_host changes */
    int my_cols_rank = -1, my_ncols = 0;
    uint64_t my_cols_gen = ~0ull;      /* col_list_dev holds rank my_cols_rank's columns of generation my_cols_gen */
    int* slot_dev = nullptr;           /* gather plan: position of a column inside its owner's pack */
    int plan_rank = -1, plan_world = 0, plan_capacity = 0, plan_normals = 0;
    int* col_list_dev = nullptr;
    int* zs_hist = nullptr;            /* front-to-back compaction (3-D): per-layer counts and cursors */
    int* zs_cursor = nullptr;
    int zsort = 3;                     /* bit0 = tile stages, bit1 = the list of the float pass: handed on front to back (no switch since round 6) */
    float* heat = nullptr;             /* render*_heatmap: S x S floats, allocated on first use */
    bool heat_frame = false;           /* the frame being issued accumulates into heat */
    int* h_pinned = nullptr;           /* small pinned scratch for read-backs */
    int* arena = nullptr;              /* filled[0..3] | normals, one allocation */
    size_t arena_images_words = 0, arena_words = 0;
    int* pub_host = nullptr;           /* host-coherent: survivor counts + sequence number, written by the compaction */
    int* pub_dev = nullptr;            /* the same words as the device sees them */
    int pub_seq = 0;

    bool normals_asm = true;           /* normals pass interpreter: gfx950 assembly (default) or compiled (MPR_NORMALS_ASM=0) */
    bool voxel_asm = true;             /* float pass interpreter: gfx950 assembly (default) or the compiled C++ one */
    bool voxel_jit = true;             /* float pass: tapes translated to machine code on the device (kernels_voxel_jit.hip;
                                          MPR_VOXEL_JIT=0: the assembly interpreter).  Needs executable device memory. */
    uint32_t* jit_code = nullptr;      /* executable (HSA), one region per wavefront of the float pass */
    size_t jit_code_bytes = 0;
    int cus = 0;                       /* compute units of the device */
    char float_kernel[64] = "";        /* mpr_ctx_float_kernel: the kernel the last frame's float pass ran as */
    std::string stage_forms;           /* mpr_ctx_tile_stage_forms: the form each tile stage of the last frame took */
    const char* normals_kernel = "";   /* mpr_ctx_normals_kernel: ... and its normals pass */
    /* Frames that do not leave the reference's tiles and tapes behind ("fast" frames: what render* does by default; the
     * images are the reference's bit for bit):
     *  - the last tile stage pushes no tapes (TileStageArgs::no_push) when both the float and the normals pass run on the
     *    groups' tapes.  Whether that pays is measured by the stage itself, every frame, on a sixteenth of its groups; a
     *    tape whose last stage shortens too much for it (the involute gears, prospero) gets its per-tile tapes from a second
     *    run of that stage, once — from then on `hint` makes the stage push straight away;
     *  - a 3-D frame starts at the 16^3 tiles when the 64^3 stage would be a handful of wavefronts (skip_stage0).
     * What a reader of tiles / tapes / counters needs to get the reference's state back: */
    bool reference_frames = false;     /* MPR_LAST_STAGE_PUSH=1: every frame the reference's way (all stages, all tapes) */
    bool force_reference = false;      /* set while a reader re-renders the frame the reference's way */
    bool tiles_only = false;           /* ... and only its tile stages: heights and normals of the frame being read are complete and are
                                          the reference's, so the re-render leaves them alone — no float pass, no normals pass, the last
                                          stage's fills are not copied down again (they are part of the heightmap already) */
    bool stage0_only = false;          /* set by mpr_column_weights: the frame stops behind its first tile stage's evaluation */
    bool last_frame_fast = false;      /* the last frame took one of the shortcuts above ... */
    bool last_frame_lean = false;      /* ... this one: no tapes from its last tile stage */
    bool skip_stage0 = true;           /* MPR_SKIP_STAGE0=0: never start at the 16^3 tiles */
    int measure_len_forced = -1;       /* (no switch since round 6) groups per run of the last stage's sample, or -1: the frame's own choice */
    struct FrameKey {
        int dim = 0, rank = 0;
        bool parted = false;
        float mat[16] = {0};
        float z = 0.0f;
    };
    FrameKey last_key;                 /* the last frame rendered (what a reader's reference frame repeats) ... */
    std::unique_ptr<mpr_tape> last_tape;   /* ... and a copy of its tape (the caller may have freed it by the time a reader asks) */
    enum { HINT_UNKNOWN = 0, HINT_GROUPS = 1, HINT_TAPES = 2 };
    uint64_t hint_serial = 0;          /* what the last measuring frame of this tape and dimension found: the groups' tapes are ... */
    int hint_dim = 0, hint_mode = HINT_UNKNOWN;   /* ... short enough for float and normals pass (GROUPS), or per-tile tapes pay (TAPES) */
    bool tiles_vgpr = true;            /* MPR_TILES_VGPR=0 (development): tile stages keep every slot file in LDS */
    bool tiles_asm = true;             /* MPR_TILES_ASM=0 (development): compiled forward / backward walks in the tile stages */
    bool groups_always = false;        /* MPR_VOXEL_GROUPS=2 (development): group form whatever the tapes' lengths */
    int jit_slots = 0;                 /* (no switch since round 6) code slots per workgroup of the group form, 0: 16 */
    int jit_wgs_per_cu = 0;            /* (no switch since round 6) workgroups per CU of the group form's persistent grid, 0: what fits */
    int jit_gap = 0;                   /* MPR_JIT_GAP (development): dwords between code slots of the group form's ring */
    bool jit_always_invalidate = false;/* MPR_VOXEL_JIT=3 (development): the group form invalidates the instruction cache after every translation */
    bool jit_validated_arch = false;   /* gfx950: the ring's one-invalidate-per-trip was validated there and nowhere else */
    int jit_grid_cache[2][2][4] = {};  /* workgroups the device holds, per form (tile / group), dimension and slot class */
    bool voxel_jit_tiles = false;      /* MPR_VOXEL_JIT=2: generated code per smallest tile where the group form is not possible (development;
                                          slower than the interpreter for short tapes: a translation per tile).  Brute-force frames always use it:
                                          every tile runs the root tape, translated once per wavefront. */
    bool voxel_groups = true;          /* float pass in group form when the last tile stage's tapes allow it (MPR_VOXEL_GROUPS=0: never) */
    mprk::GroupInfo* groups = nullptr; /* per sibling group of the last tile stage (group form) */
    size_t groups_cap = 0;
    ulonglong2* choice_masks = nullptr;
    size_t masks_cap = 0;
    unsigned char* group_alive = nullptr;  /* per group: a tile left for the float pass (written by the last compaction) */
    int* group_list = nullptr;             /* those groups in list order, then their number (k_list_alive_groups) */
    size_t group_alive_cap = 0, group_list_cap = 0;
    int* vox_counters = nullptr;           /* the float pass on the root tape's code: its tile counters (kernels_voxel_jit.hip: VG_LISTS) */
    /* ... by footprint segments (round 6; kernels.hip: k_compact_footprints, kernels_voxel_jit.hip: k_eval_voxels_gen_fp) */
    unsigned* fp_items = nullptr;
    size_t fp_items_cap = 0;
    int* fp_meta = nullptr;                /* {segments appended, their total} */
    int fp_grid = 0;
    bool voxel_fp = true;                  /* MPR_VOXEL_FP=0: that float pass tile by tile in z order (round 3's list) */
    int* walked_dev = nullptr;             /* MPR_DEBUG_WALKED=1: tiles the float pass on the root tape's code walked (the others it found hidden), 32 x 32 ints */
    int* tile_source = nullptr;            /* per smallest tile: its index in the last tile stage's list (the float pass on the root tape's code) */
    size_t tile_source_cap = 0;

    uint64_t tape_serial = 0;          /* tape currently resident at pool[0..] */
    /* the resident tape's interval walks as generated code (tile_gen.hpp), for the frame's first tile stage; MPR_TILE_GEN=0: never,
     * 2: the forward walk only */
    int tile_gen = 1;
    uint32_t* gen_code = nullptr;      /* executable memory: forward code, then backward code */
    uint32_t* gen_stage = nullptr;     /* what the host hands over (device memory; copied into gen_code by a kernel) */
    size_t gen_cap_dw = 0;
    bool gen_ok = false;
    int gen_fwd_dw = 0, gen_bwd_dw = 0, gen_deriv_dw = 0, gen_words = 0, gen_nchoices = 0;
    /* ... and the normals pass on that tape's generated Deriv code, for frames whose first stage recorded its tiles' decisions
     * (MPR_NORMALS_GEN=0: never) */
    bool normals_gen = true;
    bool tile_gen_last = true;         /* MPR_TILE_GEN_LAST=0: the last stage of such a frame interprets its parents' tapes */
    unsigned long long* gen_dec[3] = {nullptr, nullptr, nullptr};   /* the tiles' records, per stage (TileStageArgs::gen_decisions) */
    size_t gen_dec_cap[3] = {0, 0, 0};
    int gen_full_dw = 0;               /* dwords of the backward code for tapes that are shortened again (0: the tape is too long for it) */
    std::shared_ptr<const mpr::TapeCode> resident_code;   /* what gen_code holds (kept alive: the upload is asynchronous) */
    int gen_vox_at = 0, gen_derivg_at = 0, gen_derivg_dw = 0;   /* where the float walk / the guarded forward walk / the guarded
                                                                                   Deriv walk start in gen_code (dwords) */
    bool normals_guards = true;        /* the normals pass runs the Deriv walk that jumps over what every pixel of the wavefront left dead (no switch since round 6) */
    /* frame_domain.hpp: does the last (tape, view) asked about keep every interval operation where the reference's routines are
     * isotone (the shortcuts below that are only then the reference's procedure: skip0, the loose enclosures) */
    uint64_t tame_serial = 0;
    FrameKey tame_key;
    bool tame_value = false;
    bool tame_check = true;            /* (no switch since round 6: false = every frame counts as tame) */
    /* frames that start at the 16^3 tiles and are not tame: the 64^3 tiles walked beside the frame, every 16^3 tile held against
     * its parent before the float pass is launched (kernels.hpp: launch_skip0_parents / launch_skip0_compare) */
    hipStream_t side = nullptr;
    hipEvent_t ev_begin = nullptr, ev_stage = nullptr, ev_check = nullptr, ev_done = nullptr;   /* code installed (frame's stream) / first stage through
                                          (frame's) / 64^3 tiles walked (side) / compared (side; frames that do not block) */
    bool side_must_wait = false;       /* the tape's code was installed since the side stream last looked */
    bool skip0_unchecked = false;      /* a comparison has been launched whose verdict nobody has read yet */
    unsigned long long* skip0_parents = nullptr;
    size_t skip0_parents_cap = 0;
    unsigned long long* skip0_children = nullptr;
    size_t skip0_children_cap = 0;
    int* skip0_flag_host = nullptr;    /* host-coherent; 1: some 16^3 tile did not do what its parent would have made it do */
    int* skip0_flag_dev = nullptr;
    bool skip0_verify = true;          /* MPR_SKIP0_CHECK=0 (development): such frames go unverified */
    /* per tape (by serial) whose last verified frame failed: its next frames start at the 64^3 tiles — `left` of them, then one tries
     * again —, twice as many after every failure in a row (`span`), 64 again once a verified frame has passed (one triple
     * for the whole context let two alternating tapes overwrite each other's veto, and never came down) */
    struct Skip0Veto { int left = 0, span = 64; };
    std::map<uint64_t, Skip0Veto> skip0_veto;
    bool side_compare_pending = false;  /* a verification queued on the side stream has not been waited for (frame_begin does) */
    long long frames_restarted = 0;    /* frames that started over (the pool grew, a shortcut's veto, ...): mpr_debug_frame_stats */
    long long pool_growths = 0;
    bool pool_grow_pending = false;    /* the last frame's pushes filled more than 3/4 of a pool this context sized itself: it doubles before the next
                                        * frame starts — not inside a later one whose pushes happen to run over (which tiles a neighbour's fill culls before
                                        * they push is a matter of timing, the fill level moves by a few per cent from frame to frame): round 5 */
    bool skip0_normals_veto = false;   /* the last frame's normals pass could not take the 64^3 tiles' decisions (frame_normals_pass) */
    long long skip0_vetoes = 0;        /* frames rendered again (mpr_ctx_skip0_vetoes: tests) */
    /* MPR_CTX_PARANOID: every frame that took a shortcut is rendered again the reference's way and the two frames' heights and normals
     * are compared on the device (render_checked) */
    int* paranoid_image = nullptr;
    uint32_t* paranoid_normals = nullptr;
    unsigned long long* paranoid_count = nullptr;      /* device: cells that differ (heights, normals) */
    long long paranoid_frames = 0, paranoid_compared = 0, paranoid_cells = 0;
    /* frames that start at the 16^3 tiles, of a tape whose float pass and normals pass run on its root code with records: nobody
     * walks the tapes the first stage pushes (the sample of the last stage apart: such frames take none; every 32nd frame is an
     * ordinary one and refreshes the hint) -> no backward walk in that stage, records only (TileStageArgs::gen_forward_only) */
    bool lean_first = true;            /* MPR_LEAN_FIRST=0: the first stage always walks backward and pushes */
    uint64_t lean_first_veto = 0;      /* the tape whose frame found a later stage that needs the tapes after all */
    uint64_t lean_first_serial = 0;    /* ... frames of this tape since it became resident */
    unsigned lean_first_frames = 0;
    uint64_t tapes_hint_serial = 0;    /* a tape whose last stage pushes (hint: per-tile tapes): only every 32nd of its frames keeps the groups' records and
                                        * measures the tapes again — the bookkeeping of a form the frame does not take cost architecture 1024^3 0.1 ms of
                                        * 0.79 once the stages' atomics were out of the way (round 5) */
    unsigned tapes_hint_frames = 0;
    bool tile_gen_loose = true;        /* MPR_TILE_GEN_LOOSE=0: frames nobody reads keep the correctly rounded exp / log enclosures in their tile stages */
    /* the scheduled interval forward walks (interval_gen.hpp) in gen_code: [kind][exact, loose, tight] */
    int gen_iw_at[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, gen_iw_dw[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    unsigned char* tight_skip = nullptr;   /* per tile of the last stage's list (TileStageArgs::tight_skip) */
    size_t tight_skip_cap = 0;
    int* tight_image = nullptr;            /* TileStageArgs::tight_image */
    size_t tight_image_cap = 0;
    bool tile_tight = true;            /* MPR_TILE_TIGHT=0: the last tile stage of a frame nobody reads hands the float pass every tile the reference's
                                        * enclosures leave ambiguous (round 6: with it, those a sound sin / cos enclosure decides stay away) */
    /* the resident tape's first-stage walk for the kernel with 93 slots in registers (internal.hpp: mpr_tape::big_fwd), or none */
    uint32_t* big_code = nullptr;
    uint32_t* big_stage = nullptr;
    size_t big_cap_dw = 0;
    bool big_ok = false;
    int big_end = 0;
    size_t big_bwd_at = 0;             /* dwords into big_code where the backward walk starts */
    bool big_bwd_ok = false;           /* ... and whether there is one */
    bool raw_reads = false;            /* MPR_DEBUG_RAW_READS=1 (tests): readers get the tiles and tapes of the frame AS IT RAN, no second rendering */
    bool tile_gen_big_bwd = true;      /* MPR_TILE_GEN_BIG_BWD=0 (development): the interpreter's backward walk behind the generated forward walk */
    std::shared_ptr<const std::vector<uint32_t>> big_resident;
    bool tile_gen_big = true;          /* MPR_TILE_GEN_BIG=0 (development): such tapes' first stages walk on the interpreter */
    int big_min_tiles = 8193;          /* MPR_TILE_GEN_BIG_TILES: first stages of at least this many tiles take it rather than the level-parallel kernel (which wins up to its own limit of 8192: scripts/probe_big.py) */
    unsigned int* redo_count = nullptr;   /* MPR_DEBUG_REDO=1: {wavefronts that ran generated forward code, of them: redone on the exact code} */
    bool tile_gen_lean = true;         /* MPR_TILE_GEN_LEAN=0: loose stages that push nothing in the 128-register kernel too (four wavefronts per SIMD) */
    unsigned char* redo_flags = nullptr;  /* per workgroup of a lean stage: the launch behind it runs the wavefronts flagged here */
    size_t redo_flags_cap = 0;
    bool tile_gen_guards = true;       /* MPR_TILE_GEN_GUARDS=0: a lean last stage runs the plain forward walk */
    int gen_vox_dw = 0;                /* dwords of the float walk (voxel_gen.hpp), behind the four above (0: none) */
    bool voxel_gen = true;             /* MPR_VOXEL_GEN=0: the float pass never runs the root tape's host-generated code */
    int voxel_gen_min_run = 5;         /* MPR_VOXEL_GEN_RUN (development): shortest run of dead clauses that gets a guard (0: none) */
    int vox_grid_cache[2] = {0, 0};
    int voxel_gen_tiles = 0;           /* MPR_VOXEL_GEN_TILES (development): consecutive tiles a wavefront takes per atomic (default 4) */
    int voxel_gen_wgs = 0;             /* (no switch since round 6) at most this many persistent workgroups per CU, 0: what fits */
    bool tile_gen_chain = true;        /* MPR_TILE_GEN_CHAIN=0: only a frame's first stage (and, in frames that start at the 16^3 tiles, the last) */
    void* sched_recs = nullptr;        /* the resident tape's level schedule (tape_schedule.hpp), or unused */
    int* sched_levels = nullptr;
    uint16_t* sched_prev = nullptr;    /* TapeSchedule::prev_writer */
    uint32_t* sched_defs = nullptr;    /* TapeSchedule::defs */
    size_t sched_recs_cap = 0, sched_levels_cap = 0, sched_prev_cap = 0, sched_defs_cap = 0;
    bool sched_ok = false;
    int sched_nlevels = 0, sched_nclauses = 0, sched_root = 0;
    bool wide_stage0 = true;           /* MPR_WIDE_STAGE0=0: first stage with the one-lane-per-tile kernel */
    int wide_later = -1;               /* MPR_WIDE_LATER: later stages run level-parallel too while they have at most this many
                                          tiles and the stage before did (0: never; default: two rounds of workgroups on the chip) */
    int wide_threads = 0;              /* (no switch since round 6) workgroup size of the level-parallel kernel, 0: by the tape */
    uint32_t* wide_bits[2] = {nullptr, nullptr};   /* kernels_wide.hip: inherited-tape tables, ping-pong between stages */
    size_t wide_bits_cap[2] = {0, 0};
    /* development switches, read once when the context is created (never per frame) */
    bool wide_force = false;           /* MPR_WIDE_FORCE: level-parallel first stage whatever the DAG's shape */
    bool dynamic_choices = true;       /* a stage's choice array sized by what the stage above reported, not by the root tape (no switch since round 6) */
    int debug_tiles = 0;               /* MPR_DEBUG_TILES: 1 = skip tape pushing, 2 = skip the arithmetic, 4 = cycle breakdown */
    bool debug_choices = false;        /* (no switch since round 6) print the stages' samples */
    int tape_len = 0;

    mpr_counters last = {};
    std::vector<KernelTiming> timings;
    size_t timings_used = 0;
    bool frame_pending = false;
    int pending_dim = 0;
};

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            return mpr::set_error(MPR_ERR_NO_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
        }                                                                                    \
    } while (0)

/* Streams are recycled, never destroyed.  hipStreamDestroy of the HIP runtime torch bundles (ROCm 7.0's ROCclr: commandqueue.cpp,
 * HostQueue::terminate, "Marker queued to ensure finish") deletes the stream's roc::VirtualGPU while the HSA signal handlers of its
 * last commands may still wait for the runtime's handler thread; each of them then decrements a 64-bit counter at offset 0x98 of
 * the freed object (`lock subq $1, 0x98(gpu)` at the head of the handler registered with hsa_amd_signal_async_handler) — a write
 * after free into whatever the process' heap placed there next.  Found with scripts/heapguard.c: 1944 such writes in 24 processes
 * of the fuzz sweep; round 5's segmentation fault was one that landed in a std::vector being read (profiles/r06_segv_hunt.txt).
 * A context takes its two streams from the idle ones of its device and gives them back, drained, when it is destroyed. */
namespace {
std::mutex g_idle_streams_mutex;
std::vector<std::pair<int, hipStream_t>> g_idle_streams;
}

static hipError_t acquire_stream(int device, hipStream_t* out)
{
    {
        std::lock_guard<std::mutex> lock(g_idle_streams_mutex);
        for (size_t i = g_idle_streams.size(); i-- > 0;)
            if (g_idle_streams[i].first == device) {
                *out = g_idle_streams[i].second;
                g_idle_streams.erase(g_idle_streams.begin() + (long)i);
                return hipSuccess;
            }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

static void release_stream(int device, hipStream_t s)
{
    (void)hipStreamSynchronize(s);
    std::lock_guard<std::mutex> lock(g_idle_streams_mutex);
    g_idle_streams.emplace_back(device, s);
}

static int ensure_tiles(mpr_context* c, int stage, size_t n)
{
    if (n <= c->tiles_cap[stage]) return MPR_OK;
    /* grow geometrically so that a sequence of frames settles after a few allocations */
    size_t cap = std::max(n, c->tiles_cap[stage] + c->tiles_cap[stage] / 2);
    if (c->tiles[stage]) HIP_TRY(hipFree(c->tiles[stage]));
    c->tiles[stage] = nullptr;
    c->tiles_cap[stage] = 0;
    HIP_TRY(hipMalloc((void**)&c->tiles[stage], cap * sizeof(mpr_tile_node)));
    c->tiles_cap[stage] = cap;
    return MPR_OK;
}

/* ---- executable device memory (for the float pass's generated code): HIP has no such allocation, the
 * HSA runtime underneath it has (hsa_amd_memory_pool_allocate + HSA_AMD_MEMORY_POOL_EXECUTABLE_FLAG) ---- */
namespace {
struct ExecPoolSearch {
    uint32_t bdf = 0, domain = 0;
    int ordinal = 0, seen = 0;
    bool by_bdf = false, found = false;
    hsa_agent_t agent{};
    hsa_amd_memory_pool_t pool{};
    bool have_pool = false;
};
hsa_status_t exec_agent_cb(hsa_agent_t a, void* data)
{
    auto* q = static_cast<ExecPoolSearch*>(data);
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS || t != HSA_DEVICE_TYPE_GPU) return HSA_STATUS_SUCCESS;
    uint32_t bdf = 0, dom = 0;
    (void)hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf);
    (void)hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &dom);
    const bool match = q->by_bdf ? (bdf == q->bdf && dom == q->domain) : (q->seen == q->ordinal);
    q->seen++;
    if (match && !q->found) {
        q->agent = a;
        q->found = true;
    }
    return HSA_STATUS_SUCCESS;
}
hsa_status_t exec_pool_cb(hsa_amd_memory_pool_t p, void* data)
{
    auto* q = static_cast<ExecPoolSearch*>(data);
    hsa_amd_segment_t seg;
    if (hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg) != HSA_STATUS_SUCCESS || seg != HSA_AMD_SEGMENT_GLOBAL)
        return HSA_STATUS_SUCCESS;
    uint32_t flags = 0;
    bool alloc = false;
    (void)hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    (void)hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
    if (alloc && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !q->have_pool) {
        q->pool = p;
        q->have_pool = true;
    }
    return HSA_STATUS_SUCCESS;
}
/* null when the runtime refuses (the caller then keeps the interpreter).  The HSA agent is the one that owns a piece of memory
 * hipMalloc'ed on `device` (hsa_amd_pointer_info): no guessing from PCI addresses or ordinals, which HIP_VISIBLE_DEVICES reorders.
 * The PCI / ordinal match remains as the fallback for a runtime that does not answer. */
void* alloc_executable(int device, size_t bytes)
{
    static bool hsa_up = (hsa_init() == HSA_STATUS_SUCCESS);       /* reference counted: HIP has initialised it already */
    if (!hsa_up) return nullptr;
    ExecPoolSearch q;
    {
        int current = -1;
        (void)hipGetDevice(&current);
        void* probe = nullptr;
        if (hipSetDevice(device) == hipSuccess && hipMalloc(&probe, 256) == hipSuccess) {
            hsa_amd_pointer_info_t info;
            std::memset(&info, 0, sizeof(info));
            info.size = sizeof(info);
            if (hsa_amd_pointer_info(probe, &info, nullptr, nullptr, nullptr) == HSA_STATUS_SUCCESS && info.agentOwner.handle != 0) {
                q.agent = info.agentOwner;
                q.found = true;
            }
            (void)hipFree(probe);
        }
        if (current >= 0 && current != device) (void)hipSetDevice(current);
    }
    if (!q.found) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) return nullptr;
        q.by_bdf = true;
        q.bdf = ((uint32_t)prop.pciBusID << 8) | ((uint32_t)prop.pciDeviceID << 3);
        q.domain = (uint32_t)prop.pciDomainID;
        (void)hsa_iterate_agents(exec_agent_cb, &q);
        if (!q.found) {                      /* no PCI match (virtualised box): HIP ordinals follow the order of the GPU agents */
            q = ExecPoolSearch();
            q.ordinal = device;
            (void)hsa_iterate_agents(exec_agent_cb, &q);
        }
    }
    if (!q.found) return nullptr;
    (void)hsa_amd_agent_iterate_memory_pools(q.agent, exec_pool_cb, &q);
    if (!q.have_pool) return nullptr;
    void* p = nullptr;
    if (hsa_amd_memory_pool_allocate(q.pool, bytes, HSA_AMD_MEMORY_POOL_EXECUTABLE_FLAG, &p) != HSA_STATUS_SUCCESS) return nullptr;
    return p;
}
void free_executable(void* p) { if (p) (void)hsa_amd_memory_pool_free(p); }
}  // namespace

template <typename T>
static int ensure_buffer(T** ptr, size_t* cap, size_t n)
{
    if (n <= *cap) return MPR_OK;
    const size_t want = std::max(n, *cap + *cap / 2);
    if (*ptr) HIP_TRY(hipFree(*ptr));
    *ptr = nullptr;
    *cap = 0;
    HIP_TRY(hipMalloc((void**)ptr, want * sizeof(T)));
    *cap = want;
    return MPR_OK;
}

struct TimedScope {
    mpr_context* c;
    bool on;
    size_t idx = 0;
    hipStream_t st;
    TimedScope(mpr_context* ctx, const char* name, hipStream_t on_stream = nullptr)
        : c(ctx), on((ctx->flags & MPR_CTX_TIMING) != 0 || ((ctx->flags & MPR_CTX_TIMING_FLOAT) != 0 && std::strcmp(name, "eval_voxels_f") == 0)),
          st(on_stream ? on_stream : ctx->stream)
    {
        if (!on) return;
        if (c->timings_used == c->timings.size()) {
            KernelTiming t;
            t.name = name;
            if (hipEventCreate(&t.start) != hipSuccess || hipEventCreate(&t.stop) != hipSuccess) {
                on = false;
                return;
            }
            c->timings.push_back(t);
        }
        idx = c->timings_used++;
        c->timings[idx].name = name;
        (void)hipEventRecord(c->timings[idx].start, st);
    }
    ~TimedScope()
    {
        if (on) (void)hipEventRecord(c->timings[idx].stop, st);
    }
};

extern "C" {

int mpr_ctx_create_ex(const mpr_ctx_options* opt, mpr_context** out)
{
    if (!opt || !out) return mpr::set_error(MPR_ERR_INVALID, "null argument");
    const int S = opt->image_size_px;
    if (S < 64 || S % 64 != 0 || S > 8192)
        return mpr::set_error(MPR_ERR_INVALID, "image_size_px must be a multiple of 64 in [64, 8192]");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return mpr::set_error(MPR_ERR_NO_DEVICE, std::string("no HIP device available: ") +
                                                     (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
    if (opt->device < 0 || opt->device >= ndev) return mpr::set_error(MPR_ERR_INVALID, "device ordinal out of range");
    HIP_TRY(hipSetDevice(opt->device));

    mpr_context* c = new mpr_context();
    c->device = opt->device;
    c->S = S;
    c->flags = opt->flags;
    if (const char* e = getenv("MPR_VOXEL_ASM")) c->voxel_asm = atoi(e) != 0;
    if (const char* e = getenv("MPR_VOXEL_JIT")) { c->voxel_jit = atoi(e) != 0; c->voxel_jit_tiles = atoi(e) == 2; c->jit_always_invalidate = atoi(e) == 3; }
    if (const char* e = getenv("MPR_VOXEL_GROUPS")) { c->voxel_groups = atoi(e) != 0; c->groups_always = atoi(e) == 2; }
    if (const char* e = getenv("MPR_JIT_GAP")) c->jit_gap = atoi(e);
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, opt->device) == hipSuccess) {
            c->cus = prop.multiProcessorCount;
            /* the ring of code slots that is invalidated once per trip was validated on gfx950 (scripts/ubench/jit_probe.hip) */
            c->jit_validated_arch = std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
        }
    }
    if (const char* e = getenv("MPR_NORMALS_ASM")) c->normals_asm = atoi(e) != 0;
    if (const char* e = getenv("MPR_TILE_GEN")) c->tile_gen = atoi(e);
    if (const char* e = getenv("MPR_NORMALS_GEN")) c->normals_gen = atoi(e) != 0;
    if (const char* e = getenv("MPR_VOXEL_GEN")) c->voxel_gen = atoi(e) != 0;
    if (const char* e = getenv("MPR_TILE_GEN_GUARDS")) c->tile_gen_guards = atoi(e) != 0;
    if (const char* e = getenv("MPR_TILE_GEN_LEAN")) c->tile_gen_lean = atoi(e) != 0;
    if (const char* e = getenv("MPR_TILE_TIGHT")) c->tile_tight = atoi(e) != 0;
    if (const char* e = getenv("MPR_VOXEL_FP")) c->voxel_fp = atoi(e) != 0;
    if (const char* e = getenv("MPR_DEBUG_WALKED"))
        if (atoi(e) != 0) {
            if (hipMalloc((void**)&c->walked_dev, 1024 * sizeof(int)) != hipSuccess) c->walked_dev = nullptr;
            else (void)hipMemset(c->walked_dev, 0, 1024 * sizeof(int));
        }
    if (const char* e = getenv("MPR_DEBUG_REDO"))
        if (atoi(e) != 0 && hipMalloc((void**)&c->redo_count, 2 * sizeof(unsigned int)) == hipSuccess) (void)hipMemset(c->redo_count, 0, 2 * sizeof(unsigned int));
    if (const char* e = getenv("MPR_TILE_GEN_LOOSE")) c->tile_gen_loose = atoi(e) != 0;
    if (const char* e = getenv("MPR_SKIP0_CHECK")) c->skip0_verify = atoi(e) != 0;
    if (const char* e = getenv("MPR_LEAN_FIRST")) c->lean_first = atoi(e) != 0;
    if (const char* e = getenv("MPR_VOXEL_GEN_RUN")) c->voxel_gen_min_run = atoi(e);
    if (const char* e = getenv("MPR_VOXEL_GEN_TILES")) c->voxel_gen_tiles = atoi(e);
    if (const char* e = getenv("MPR_TILE_GEN_LAST")) c->tile_gen_last = atoi(e) != 0;
    if (const char* e = getenv("MPR_TILE_GEN_CHAIN")) c->tile_gen_chain = atoi(e) != 0;
    if (const char* e = getenv("MPR_WIDE_STAGE0")) c->wide_stage0 = atoi(e) != 0;
    c->wide_force = getenv("MPR_WIDE_FORCE") != nullptr;
    if (const char* e = getenv("MPR_TILES_ASM")) c->tiles_asm = atoi(e) != 0;
    if (const char* e = getenv("MPR_TILES_VGPR")) c->tiles_vgpr = atoi(e) != 0;
    if (const char* e = getenv("MPR_TILE_GEN_BIG")) c->tile_gen_big = atoi(e) != 0;
    if (const char* e = getenv("MPR_TILE_GEN_BIG_TILES")) c->big_min_tiles = atoi(e);
    if (const char* e = getenv("MPR_TILE_GEN_BIG_BWD")) c->tile_gen_big_bwd = atoi(e) != 0;
    c->raw_reads = getenv("MPR_DEBUG_RAW_READS") != nullptr;
    if (const char* e = getenv("MPR_LAST_STAGE_PUSH")) c->reference_frames = atoi(e) != 0;
    if (const char* e = getenv("MPR_SKIP_STAGE0")) c->skip_stage0 = atoi(e) != 0;
    if (const char* e = getenv("MPR_WIDE_LATER")) c->wide_later = atoi(e);
    if (const char* e = getenv("MPR_DEBUG_TILES")) c->debug_tiles = atoi(e);
    /* Tape pool.  The reference allocates NUM_SUBTAPES * 64 clauses whatever the frame (inc/parameters.hpp:14-22: 328 MB, 3.28 GB
     * with BIG_SERVER); a frame of bear 1024^3 that leaves the reference's tapes behind fills 1.1 GB of it, an ordinary frame
     * (no tapes from its last tile stage) under 0.2 GB, prospero 1024^2 a few MB.  A capacity the caller names is kept, with
     * the reference's behaviour when it runs out (tiles keep their parents' tapes, src/context.cu:336-347).  Otherwise: 2 M
     * clauses (16 MB) per 256 px of image side to begin with, at least 4 M, doubled on demand. */
    c->pool_auto = opt->pool_clauses <= 0;
    if (c->pool_auto) c->pool_cap = std::max<long long>(4ll << 20, (long long)(S / 256) * (2ll << 20));
    else if (c->pool_cap == 0) c->pool_cap = opt->pool_clauses;
    if (c->pool_cap > 0x7FFFFFFFll) c->pool_cap = 0x7FFFFFFFll;   /* tape indices are int32 (inc/context.hpp:25) */
    *out = nullptr;
#define CT(expr)                                                     \
    do {                                                             \
        hipError_t _e = (expr);                                      \
        if (_e != hipSuccess) {                                      \
            mpr::set_error(MPR_ERR_ALLOC, std::string(#expr) + ": " + hipGetErrorString(_e)); \
            mpr_ctx_destroy(c);                                      \
            return MPR_ERR_ALLOC;                                    \
        }                                                            \
    } while (0)
    CT(acquire_stream(c->device, &c->stream));
    {
        /* the four filled images and the normals (src/context.cpp:21-27) live in one allocation, in this
         * order, so that one kernel resets them at the start of a frame (mprk::launch_begin_frame) */
        size_t off[5], words = 0;
        for (int i = 0; i < 4; ++i) {
            const int ts = 64 >> (2 * i);
            c->filled_n[i] = (size_t)(S / ts) * (S / ts);
            off[i] = words;
            words += (c->filled_n[i] + 63) & ~(size_t)63;
        }
        off[4] = words;
        words += (size_t)S * S;
        CT(hipMalloc((void**)&c->arena, words * sizeof(int)));
        for (int i = 0; i < 4; ++i) c->filled[i] = c->arena + off[i];
        c->normals = reinterpret_cast<uint32_t*>(c->arena + off[4]);
        c->arena_images_words = off[4];
        c->arena_words = words;
    }
    CT(hipMalloc((void**)&c->pool, ((size_t)c->pool_cap + 128) * sizeof(uint64_t)));   /* + slack: walkers fetch 64-word blocks */
    CT(hipMalloc((void**)&c->tape_index, 2 * sizeof(unsigned long long)));   /* [0] index, [1] sticky overflow flag of the frame */
    CT(hipMalloc((void**)&c->num_active, 8 * sizeof(int)));      /* [0..2] counts, [3] workgroups done, [4] choices the next stage needs, [5..6] tape lengths of the last stage, [7] the float pass's next group */
    CT(hipMalloc((void**)&c->zs_hist, 1024 * sizeof(int)));
    CT(hipMalloc((void**)&c->zs_cursor, 1024 * sizeof(int)));
    CT(hipMemsetAsync(c->zs_hist, 0, 1024 * sizeof(int), c->stream));
    CT(hipMalloc((void**)&c->counters, (mprk::CNT_COUNT + 32) * sizeof(unsigned long long)));
    CT(hipMalloc((void**)&c->owner_dev, (size_t)(S / 64) * (S / 64) * sizeof(int)));
    CT(hipMalloc((void**)&c->col_list_dev, (size_t)(S / 64) * (S / 64) * sizeof(int)));
    CT(hipMalloc((void**)&c->slot_dev, (size_t)(S / 64) * (S / 64) * sizeof(int)));
    CT(hipHostMalloc((void**)&c->h_pinned, 64 * sizeof(unsigned long long), hipHostMallocDefault));
    CT(hipHostMalloc((void**)&c->pub_host, 16 * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(c->pub_host, 0, 16 * sizeof(int));
    CT(hipHostGetDevicePointer((void**)&c->pub_dev, c->pub_host, 0));
    CT(hipHostMalloc((void**)&c->skip0_flag_host, 16 * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(c->skip0_flag_host, 0, 16 * sizeof(int));
    CT(hipHostGetDevicePointer((void**)&c->skip0_flag_dev, c->skip0_flag_host, 0));
    CT(acquire_stream(c->device, &c->side));
    CT(hipEventCreateWithFlags(&c->ev_begin, hipEventDisableTiming));
    CT(hipEventCreateWithFlags(&c->ev_stage, hipEventDisableTiming));
    CT(hipEventCreateWithFlags(&c->ev_check, hipEventDisableTiming));
    CT(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
    CT(hipMemsetAsync(c->num_active, 0, 8 * sizeof(int), c->stream));
    CT(hipMemsetAsync(c->arena, 0, c->arena_words * sizeof(int), c->stream));
    CT(hipStreamSynchronize(c->stream));
#undef CT
    *out = c;
    return MPR_OK;
}

int mpr_ctx_create(int32_t device, int32_t image_size_px, mpr_context** out)
{
    mpr_ctx_options o;
    std::memset(&o, 0, sizeof(o));
    o.device = device;
    o.image_size_px = image_size_px;
    return mpr_ctx_create_ex(&o, out);
}

void mpr_ctx_destroy(mpr_context* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->side) {
        release_stream(c->device, c->side);
    }
    if (c->ev_begin) (void)hipEventDestroy(c->ev_begin);
    if (c->ev_stage) (void)hipEventDestroy(c->ev_stage);
    if (c->ev_check) (void)hipEventDestroy(c->ev_check);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    if (c->paranoid_image) (void)hipFree(c->paranoid_image);
    if (c->paranoid_normals) (void)hipFree(c->paranoid_normals);
    if (c->paranoid_count) (void)hipFree(c->paranoid_count);
    if (c->big_code) free_executable(c->big_code);
    if (c->big_stage) (void)hipFree(c->big_stage);
    if (c->skip0_parents) (void)hipFree(c->skip0_parents);
    if (c->skip0_children) (void)hipFree(c->skip0_children);
    if (c->skip0_flag_host) (void)hipHostFree(c->skip0_flag_host);
    for (int i = 0; i < 4; ++i) {
        c->filled[i] = nullptr;
        if (c->tiles[i]) (void)hipFree(c->tiles[i]);
    }
    if (c->arena) (void)hipFree(c->arena);
    if (c->pool) (void)hipFree(c->pool);
    if (c->tape_index) (void)hipFree(c->tape_index);
    free_executable(c->jit_code);
    free_executable(c->gen_code);
    if (c->redo_flags) (void)hipFree(c->redo_flags);
    if (c->gen_stage) (void)hipFree(c->gen_stage);
    for (int i = 0; i < 3; ++i) if (c->gen_dec[i]) (void)hipFree(c->gen_dec[i]);
    if (c->groups) (void)hipFree(c->groups);
    if (c->group_alive) (void)hipFree(c->group_alive);
    if (c->group_list) (void)hipFree(c->group_list);
    if (c->tile_source) (void)hipFree(c->tile_source);
    if (c->vox_counters) (void)hipFree(c->vox_counters);
    if (c->fp_items) (void)hipFree(c->fp_items);
    if (c->fp_meta) (void)hipFree(c->fp_meta);
    if (c->tight_skip) (void)hipFree(c->tight_skip);
    if (c->tight_image) (void)hipFree(c->tight_image);
    if (c->walked_dev) (void)hipFree(c->walked_dev);
    for (int i = 0; i < 2; ++i) if (c->wide_bits[i]) (void)hipFree(c->wide_bits[i]);
    if (c->choice_masks) (void)hipFree(c->choice_masks);
    if (c->num_active) (void)hipFree(c->num_active);
    if (c->zs_hist) (void)hipFree(c->zs_hist);
    if (c->zs_cursor) (void)hipFree(c->zs_cursor);
    if (c->counters) (void)hipFree(c->counters);
    if (c->owner_dev) (void)hipFree(c->owner_dev);
    if (c->col_list_dev) (void)hipFree(c->col_list_dev);
    if (c->heat) (void)hipFree(c->heat);
    if (c->slot_dev) (void)hipFree(c->slot_dev);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    if (c->pub_host) (void)hipHostFree(c->pub_host);
    for (auto& t : c->timings) {
        (void)hipEventDestroy(t.start);
        (void)hipEventDestroy(t.stop);
    }
    if (c->sched_recs) (void)hipFree(c->sched_recs);
    if (c->sched_levels) (void)hipFree(c->sched_levels);
    if (c->sched_prev) (void)hipFree(c->sched_prev);
    if (c->sched_defs) (void)hipFree(c->sched_defs);
    if (c->stream) release_stream(c->device, c->stream);
    delete c;
}

int32_t mpr_ctx_image_size(const mpr_context* c) { return c ? c->S : 0; }

/* device memory the context holds right now (every allocation that grows with the frame; the fixed few KB of counters apart) */
int64_t mpr_ctx_resident_bytes(const mpr_context* c)
{
    if (!c) return 0;
    size_t b = c->arena_words * sizeof(int) + ((size_t)c->pool_cap + 128) * sizeof(uint64_t) + c->jit_code_bytes;
    for (int i = 0; i < 4; ++i) b += c->tiles_cap[i] * sizeof(mpr_tile_node);
    b += c->groups_cap * sizeof(mprk::GroupInfo) + c->masks_cap * sizeof(ulonglong2) + c->group_alive_cap + (c->group_list_cap + c->tile_source_cap) * sizeof(int);
    b += (c->wide_bits_cap[0] + c->wide_bits_cap[1]) * sizeof(uint32_t);
    b += c->sched_recs_cap + c->sched_levels_cap + c->sched_prev_cap + c->sched_defs_cap + 2 * c->gen_cap_dw * sizeof(uint32_t) + (c->gen_dec_cap[0] + c->gen_dec_cap[1] + c->gen_dec_cap[2]) * sizeof(unsigned long long);
    b += 3 * (size_t)(c->S / 64) * (c->S / 64) * sizeof(int) + (c->heat ? (size_t)c->S * c->S * sizeof(float) : 0);
    b += (c->skip0_parents_cap + c->skip0_children_cap) * sizeof(unsigned long long);       /* 8 MB at 1024^3: the verification's notes */
    return (int64_t)b;
}

}  // extern "C"

/* ---- the frame ------------------------------------------------------------------------- */
static int begin_frame(mpr_context* c, const mpr_tape* tape, const int32_t* owner)
{
    if (!c || !tape) return mpr::set_error(MPR_ERR_INVALID, "null context or tape");
    HIP_TRY(hipSetDevice(c->device));
    const int len = (int)tape->clauses.size();
    if (len < 2) return mpr::set_error(MPR_ERR_INVALID, "empty tape");
    if (c->frame_pending) HIP_TRY(hipStreamSynchronize(c->stream));
    c->frame_pending = false;
    if (c->pool_auto && c->pool_grow_pending && c->pool_cap < (long long)MPR_NUM_SUBTAPES_BIG * MPR_SUBTAPE_CHUNK) {
        /* (by half, not by the whole — three architecture 2048^3 contexts went from under 3 GB to 4.2 — and, no frame being
         * in flight, the old pool goes before the new one comes: the peak is the new pool, not old + new) */
        const long long bigger = std::min<long long>(c->pool_cap + c->pool_cap / 2, (long long)MPR_NUM_SUBTAPES_BIG * MPR_SUBTAPE_CHUNK);
        const long long old_cap = c->pool_cap;
        (void)hipFree(c->pool);
        c->pool = nullptr;
        uint64_t* fresh = nullptr;
        if (hipMalloc((void**)&fresh, ((size_t)bigger + 128) * sizeof(uint64_t)) == hipSuccess) {
            c->pool = fresh;
            c->pool_cap = bigger;
            ++c->pool_growths;
        } else {
            (void)hipGetLastError();
            if (hipMalloc((void**)&fresh, ((size_t)old_cap + 128) * sizeof(uint64_t)) != hipSuccess)
                return mpr::set_error(MPR_ERR_ALLOC, "no memory for the tape pool");
            c->pool = fresh;
        }
        c->tape_serial = 0;
    }
    c->pool_grow_pending = false;
    if (c->pool_auto && (long long)len + 4096 >= c->pool_cap) {
        /* a pool this context sized itself also grows for the root tape (pushes make it grow later, frame by frame) */
        long long bigger = c->pool_cap;
        while ((long long)len + 4096 >= bigger && bigger < (long long)MPR_NUM_SUBTAPES_BIG * MPR_SUBTAPE_CHUNK) bigger *= 2;
        bigger = std::min<long long>(bigger, (long long)MPR_NUM_SUBTAPES_BIG * MPR_SUBTAPE_CHUNK);
        uint64_t* fresh = nullptr;
        if (bigger > c->pool_cap && hipMalloc((void**)&fresh, ((size_t)bigger + 128) * sizeof(uint64_t)) == hipSuccess) {
            (void)hipFree(c->pool);
            c->pool = fresh;
            c->pool_cap = bigger;
            c->tape_serial = 0;
        } else {
            (void)hipGetLastError();
        }
    }
    if ((long long)len >= c->pool_cap) return mpr::set_error(MPR_ERR_INVALID, "tape does not fit the pool");
    /* copy the tape to pool[0..len) (src/context.cu:1139-1142); skipped when already resident,
     * the pool's first len words are never overwritten by pushes */
    if (c->tape_serial != tape->serial) {
        HIP_TRY(hipMemcpyAsync(c->pool, tape->clauses.data(), (size_t)len * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));   /* pageable source must stay valid */
        c->tape_serial = tape->serial;
        c->tape_len = len;
        /* the tape's walks as machine code (made with the tape: internal.hpp TapeCode; a development switch that asks for other
         * code has it made here): copied into executable memory by a kernel on the frame's stream — the frame's launches queue
         * behind it, nothing waits on the host; the context keeps the source alive */
        c->gen_ok = false;
        if (c->tile_gen && c->tiles_asm && c->tiles_vgpr && tape->num_slots <= mpr::TILE_GEN_MAX_SLOTS) {
            std::shared_ptr<const mpr::TapeCode> code = tape->code;
            if (!code || code->vox_min_run != c->voxel_gen_min_run) code = mpr::build_tape_code(tape->clauses.data(), len, c->voxel_gen_min_run);
            const size_t ndw = code ? code->words.size() : 0;
            if (ndw > 0) {
                if (ndw > c->gen_cap_dw) {
                    HIP_TRY(hipStreamSynchronize(c->stream));
                    free_executable(c->gen_code);
                    if (c->gen_stage) (void)hipFree(c->gen_stage);
                    c->gen_code = nullptr;
                    c->gen_stage = nullptr;
                    c->gen_cap_dw = 0;
                    const size_t want = (ndw + 1023) & ~(size_t)1023;
                    c->gen_code = static_cast<uint32_t*>(alloc_executable(c->device, want * sizeof(uint32_t)));
                    if (c->gen_code && hipMalloc((void**)&c->gen_stage, want * sizeof(uint32_t)) == hipSuccess) c->gen_cap_dw = want;
                }
                if (c->gen_cap_dw >= ndw) {
                    c->resident_code = code;
                    HIP_TRY(hipMemcpyAsync(c->gen_stage, code->words.data(), ndw * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
                    mprk::launch_install_code(c->stream, c->gen_code, c->gen_stage, ndw, std::max(c->cus, 1));
                    HIP_TRY(hipGetLastError());
                    HIP_TRY(hipEventRecord(c->ev_begin, c->stream));       /* (the side stream's walk of the 64^3 tiles waits for it) */
                    c->side_must_wait = true;
                    c->gen_ok = true;
                    c->gen_fwd_dw = code->fwd_dw;
                    c->gen_bwd_dw = code->bwd_dw;
                    c->gen_deriv_dw = code->deriv_dw;
                    c->gen_full_dw = code->full_dw;
                    c->gen_vox_dw = c->voxel_gen ? code->vox_dw : 0;
                    c->gen_vox_at = code->fwd_dw + code->bwd_dw + code->deriv_dw + code->full_dw;
                    c->gen_derivg_at = c->gen_vox_at + code->vox_dw;
                    c->gen_derivg_dw = c->normals_guards ? code->derivg_dw : 0;
                    c->gen_words = code->walk_words;
                    c->gen_nchoices = code->nchoices;
                    for (int k = 0; k < 3; ++k)
                        for (int l = 0; l < 3; ++l) {
                            c->gen_iw_at[k][l] = code->iw_at[k][l];
                            c->gen_iw_dw[k][l] = code->iw_dw[k][l];
                        }
                }
            }
        }
        c->big_ok = false;
        c->big_bwd_ok = false;
        if ((tape->big_fwd || tape->big_bwd) && c->tile_gen_big && c->tiles_asm && c->tiles_vgpr) {
            const size_t fdw = tape->big_fwd ? ((tape->big_fwd->size() + 63) & ~(size_t)63) : 0;
            const size_t ndw = fdw + (tape->big_bwd ? tape->big_bwd->size() : 0);
            if (ndw > c->big_cap_dw) {
                HIP_TRY(hipStreamSynchronize(c->stream));
                free_executable(c->big_code);
                if (c->big_stage) (void)hipFree(c->big_stage);
                c->big_code = nullptr;
                c->big_stage = nullptr;
                c->big_cap_dw = 0;
                const size_t want = (ndw + 1023) & ~(size_t)1023;
                c->big_code = static_cast<uint32_t*>(alloc_executable(c->device, want * sizeof(uint32_t)));
                if (c->big_code && hipMalloc((void**)&c->big_stage, want * sizeof(uint32_t)) == hipSuccess) c->big_cap_dw = want;
                else (void)hipGetLastError();
            }
            if (c->big_cap_dw >= ndw) {
                c->big_resident = tape->big_fwd;
                std::vector<uint32_t> all(ndw, 0xBF800000u);
                if (tape->big_fwd) std::copy(tape->big_fwd->begin(), tape->big_fwd->end(), all.begin());
                if (tape->big_bwd) std::copy(tape->big_bwd->begin(), tape->big_bwd->end(), all.begin() + (long)fdw);
                c->big_bwd_at = fdw;
                HIP_TRY(hipMemcpyAsync(c->big_stage, all.data(), ndw * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
                HIP_TRY(hipStreamSynchronize(c->stream));       /* (the staging vector goes out of scope) */
                mprk::launch_install_code(c->stream, c->big_code, c->big_stage, ndw, std::max(c->cus, 1));
                HIP_TRY(hipGetLastError());
                c->big_ok = tape->big_fwd != nullptr;
                c->big_bwd_ok = tape->big_bwd != nullptr;
                c->big_end = tape->big_end;
            }
        }
        /* the tape's level schedule for the wide first-stage kernel */
        const mpr::TapeSchedule& sc = tape->schedule;
        /* level-parallel first stage only for DAGs that are wide enough to feed a wavefront: with a
         * handful of clauses per level (bear: 544 clauses in 72 levels) the 64 tiles-per-wave walk,
         * whose latency per clause the assembly interpreter cut to a third, is the faster one */
        c->sched_ok = sc.ok && mprk::wide_stage_fits(sc.nclauses) &&
                      (sc.nclauses >= 12 * ((int)sc.level_start.size() - 1) || c->wide_force);
        if (c->sched_ok) {
            const size_t rb = sc.recs.size() * sizeof(mpr::SchedRec), lb = sc.level_start.size() * sizeof(int32_t);
            if (rb > c->sched_recs_cap) {
                if (c->sched_recs) (void)hipFree(c->sched_recs);
                c->sched_recs = nullptr;
                c->sched_recs_cap = 0;
                HIP_TRY(hipMalloc(&c->sched_recs, rb));
                c->sched_recs_cap = rb;
            }
            if (lb > c->sched_levels_cap) {
                if (c->sched_levels) (void)hipFree(c->sched_levels);
                c->sched_levels = nullptr;
                c->sched_levels_cap = 0;
                HIP_TRY(hipMalloc((void**)&c->sched_levels, lb));
                c->sched_levels_cap = lb;
            }
            const size_t pb = sc.prev_writer.size() * sizeof(uint16_t);
            if (pb > c->sched_prev_cap) {
                if (c->sched_prev) (void)hipFree(c->sched_prev);
                c->sched_prev = nullptr;
                c->sched_prev_cap = 0;
                HIP_TRY(hipMalloc((void**)&c->sched_prev, pb));
                c->sched_prev_cap = pb;
            }
            HIP_TRY(hipMemcpyAsync(c->sched_prev, sc.prev_writer.data(), pb, hipMemcpyHostToDevice, c->stream));
            const size_t db = sc.defs.size() * sizeof(uint32_t);
            if (db > c->sched_defs_cap) {
                if (c->sched_defs) (void)hipFree(c->sched_defs);
                c->sched_defs = nullptr;
                c->sched_defs_cap = 0;
                HIP_TRY(hipMalloc((void**)&c->sched_defs, db));
                c->sched_defs_cap = db;
            }
            HIP_TRY(hipMemcpyAsync(c->sched_defs, sc.defs.data(), db, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(c->sched_recs, sc.recs.data(), rb, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(c->sched_levels, sc.level_start.data(), lb, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            c->sched_nlevels = (int)sc.level_start.size() - 1;
            c->sched_nclauses = sc.nclauses;
            c->sched_root = sc.root_val;
        }
    }
    if (c->flags & MPR_CTX_COUNTERS)
        HIP_TRY(hipMemsetAsync(c->counters, 0, (mprk::CNT_COUNT + 32) * sizeof(unsigned long long), c->stream));
    if (owner) {
        const size_t cols = (size_t)(c->S / 64) * (c->S / 64);
        if (c->owner_host.size() != cols || std::memcmp(c->owner_host.data(), owner, cols * sizeof(int32_t)) != 0) {
            c->owner_host.assign(owner, owner + cols);
            c->owner_gen++;
            HIP_TRY(hipMemcpyAsync(c->owner_dev, c->owner_host.data(), cols * sizeof(int), hipMemcpyHostToDevice, c->stream));
        }
    }
    c->timings_used = 0;
    std::memset(&c->last, 0, sizeof(c->last));
    for (int i = 0; i < 4; ++i) c->tiles_n[i] = 0;
    return MPR_OK;
}

/* out[0] survivors; out[1], out[2]: single tiles and pairs of the float pass (last stage with pairing) */
/* The compaction publishes the counts into host-coherent memory, each tagged with the sequence number
 * `seq` (kernels.hip: publish_counts); spin until they show up.  The stream is polled now
 * and then so that a failed launch cannot hang the caller. */
static int read_active(mpr_context* c, int seq, int out[5])
{
    /* four 8-byte words, each {sequence number, value} written by one store (kernels.hip: publish_counts) */
    const unsigned long long* const p = reinterpret_cast<const unsigned long long*>(c->pub_host);
    auto arrived = [&]() {
        for (int k = 0; k < 5; ++k) {
            const unsigned long long w = __atomic_load_n(&p[k], __ATOMIC_ACQUIRE);
            if ((unsigned)(w >> 32) != (unsigned)seq) return false;
            out[k] = (int)(unsigned)w;
        }
        return true;
    };
    for (unsigned spins = 1;; ++spins) {
        if (arrived()) break;
        if ((spins & 0xFFF) == 0) {
            const hipError_t q = hipStreamQuery(c->stream);
            if (q == hipSuccess) {
                if (arrived()) break;
                return mpr::set_error(MPR_ERR_NO_DEVICE, "the compaction finished without publishing its counts");
            }
            if (q != hipErrorNotReady) return mpr::set_error(MPR_ERR_NO_DEVICE, std::string("stage failed: ") + hipGetErrorString(q));
        }
    }
    return MPR_OK;
}

static void fill_mat(float dst[16], const float* src, int n)
{
    std::memset(dst, 0, 16 * sizeof(float));
    std::memcpy(dst, src, (size_t)n * sizeof(float));
}

/* Generated code for the float pass: sizes of a code region, the persistent grid, and the executable buffer itself.
 * Tile form: a region per wavefront; group form: one per workgroup.  A region holds the root tape's code length (the bound
 * for every tape shortened from it) plus 64 dwords the instruction prefetch may run into, and 320 dwords the translator
 * dumps into.  ok = false: no generated code for this tape (too many slots) or on this system (no executable memory). */
struct JitPlan {
    bool ok = false;
    size_t region = 0, slot_dw = 0, nslot = 1;
    int grid = 0;
    bool always_inv = false;
};
static int jit_prepare(mpr_context* c, const mpr_tape* tape, int dim, int nslots, bool gf, JitPlan* out)
{
    *out = JitPlan();
    if (!(c->voxel_jit && c->voxel_asm && mprk::jit_slot_class(nslots) != 0 && c->cus > 0)) return MPR_OK;
    const size_t code_dw = mprk::jit_code_dwords(tape->clauses.data(), (int)tape->clauses.size(), gf);
    /* group form: a ring of slots 4 KB apart (MPR_JIT_GAP, dwords: development), as many as fit 256 KB, at most 16 */
    const size_t gap_dw = gf ? (c->jit_gap > 0 ? (size_t)c->jit_gap : 1024) : 64;
    out->slot_dw = ((code_dw + gap_dw + 63) / 64) * 64;
    out->nslot = (gf && c->jit_slots != 1) ? std::min<size_t>(c->jit_slots > 0 ? (size_t)c->jit_slots : 16, std::max<size_t>(1, (size_t)65536 / out->slot_dw)) : 1;
    out->region = out->slot_dw * out->nslot + 320;
    const int cls = mprk::jit_slot_class(nslots);
    int& grid = c->jit_grid_cache[gf ? 1 : 0][dim - 2][cls == 24 ? 0 : cls == 40 ? 1 : cls == 96 ? 2 : 3];
    if (grid == 0) grid = mprk::jit_grid(dim, nslots, c->cus, gf);
    out->grid = grid;
    if (gf && c->jit_wgs_per_cu > 0) out->grid = std::min(grid, c->jit_wgs_per_cu * c->cus);      /* development */
    /* A slot's code must not be reached by the sequential instruction prefetch of its neighbour before it is written: 256 B
     * between slots executed stale code, 1 KB did not, 4 KB (1024 dwords) is the validated margin.  Anything closer, or
     * another device, pays the invalidate per group. */
    out->always_inv = c->jit_always_invalidate || gap_dw < 1024 || !c->jit_validated_arch;
    const size_t need = (size_t)grid * out->region * sizeof(uint32_t);
    if (need > ((size_t)4 << 30)) return MPR_OK;
    if (need > c->jit_code_bytes) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        free_executable(c->jit_code);
        c->jit_code = nullptr;
        c->jit_code_bytes = 0;
        const size_t want = std::max(need, (size_t)32 << 20);
        c->jit_code = static_cast<uint32_t*>(alloc_executable(c->device, want));
        if (c->jit_code) c->jit_code_bytes = want;
        else c->voxel_jit = false;          /* no executable memory on this system: the interpreter from now on */
    }
    out->ok = c->jit_code != nullptr;
    return MPR_OK;
}

/* One frame being issued: what was decided before its first launch (the plan) and what its stages leave for the ones behind
 * them.  render_frame drives frame_begin -> frame_tile_stage x (2 | 3) -> frame_float_pass -> frame_normals_pass -> frame_finish. */
namespace {
enum { FRAME_AGAIN = -1001,             /* the tape pool grew: the frame starts over */
       FRAME_AGAIN_REFERENCE = -1002,   /* the tapes this frame did not push are needed after all: again, the reference's way */
       FRAME_STOP = -1003 };            /* mpr_column_weights: the frame ends behind its first stage's evaluation */
struct Frame {
    mpr_context* c = nullptr;
    const mpr_tape* tape = nullptr;
    int dim = 0;
    const float* mat = nullptr;
    float z = 0.0f;
    const int32_t* owner = nullptr;
    int rank = 0;
    bool brute = false, blocking = true;
    /* the plan */
    int S = 0;
    hipStream_t s = nullptr;
    unsigned long long* cnt = nullptr;     /* instrumented frame */
    float* heat = nullptr;                 /* heatmap frame */
    int nslots = 1, choice_cap = 0;
    int stage_list[3] = {0, 0, 0};
    int nstages = 0;
    bool reference = false;                /* the reference's way: every stage from the 64 px tiles down, every tape pushed */
    bool skip0 = false;                    /* starts at the 16^3 tiles */
    bool skip0_checked = false;            /* skip0 of a frame that is not tame: verified against the 64^3 tiles before the float pass */
    bool used_loose = false;               /* a tile stage ran the loose interval code (interval_gen.hpp) */
    bool sample_groups = true;             /* the last stage keeps its groups' records / measures its tapes (false: 31 of 32 frames of a tape that pushes) */
    mprk::Skip0ParentsArgs skip0_args;
    bool skip0_side_pending = false;       /* the 64^3 tiles' walk beside the frame is prepared, not launched yet (launch_skip0_side) */
    bool lean_first = false;               /* the first stage walks forward only and leaves records, no tapes */
    bool tiles_only = false;               /* a reader's re-render: tile stages only */
    mpr_context::FrameKey key;
    /* what the stages leave behind */
    int count = 0;                         /* tiles of the stage about to run; in the end: smallest tiles */
    int stage_choice_cap = 0;              /* min / max clauses the tapes of the stage about to run can hold (reported by the stage before) */
    int hint = 0;
    bool prev_wide = false;
    bool decisions_recorded = false;       /* the stages so far ran generated code and kept their tiles' decisions ... */
    bool presence_recorded = false;        /* ... with the clauses of the tapes they pushed */
    bool last_recorded = false;            /* ... down to the smallest tiles */
    bool vox_gen_planned = false;          /* the last stage's compaction kept, per smallest tile, where it sat in that stage's list */
    bool vox_counters_cleared = false;     /* ... and cleared the float pass's work counters */
    bool vox_fp = false;                   /* ... or made footprint segments of the stage's list instead: the float pass by segments */
    bool vox_fp_beside = false;            /* ... or both (a frame that leaves the reference's list behind): the list for its reader, the segments for the float pass */
    bool tight_skip_valid = false;         /* the last tile stage was followed by the second verdict in a launch of its own: c->tight_skip */
    bool group_form = false;               /* the last stage recorded its groups' tapes and decisions, and the float pass takes them */
    bool lean_now = false;                 /* the last stage pushed no tapes */
    int group_stage = 0, group_count = 0, group_cap = 1;
};
/* the members under their old local names (the bodies below were one function once) */
#define FRAME_LOCALS                                                                                                       \
    mpr_context* const c = f.c; const mpr_tape* const tape = f.tape; const int dim = f.dim; const float* const mat = f.mat;  \
    const float z = f.z; const int32_t* const owner = f.owner; const int rank = f.rank; const bool brute = f.brute;         \
    const int S = f.S; hipStream_t const s = f.s; unsigned long long* const cnt = f.cnt; float* const heat = f.heat;        \
    const int nslots = f.nslots; const int choice_cap = f.choice_cap; const int nstages = f.nstages;                         \
    const int* const stage_list = f.stage_list; const bool reference = f.reference; const bool skip0 = f.skip0;             \
    const bool tiles_only = f.tiles_only;                                                                                   \
    (void)tape; (void)mat; (void)z; (void)owner; (void)rank; (void)brute; (void)S; (void)s; (void)cnt; (void)heat; (void)nslots;  \
    (void)choice_cap; (void)nstages; (void)stage_list; (void)reference; (void)skip0; (void)tiles_only; (void)dim; (void)c
}  // namespace

/* validate, decide what kind of frame this is, reset the images and write the first tile list (one launch) */
static int frame_begin(Frame& f)
{
    mpr_context* const c = f.c;
    const mpr_tape* const tape = f.tape;
    const int dim = f.dim, rank = f.rank;
    const float* const mat = f.mat;
    const float z = f.z;
    const int32_t* const owner = f.owner;
    const bool brute = f.brute;
    if (c && c->side_compare_pending) {
        /* a frame that started over while its verification was still queued on the side stream: that one reads the
         * children's notes this frame's first stage is about to rewrite, and raises a flag this frame is about to clear */
        HIP_TRY(hipEventSynchronize(c->ev_done));
        c->side_compare_pending = false;
    }
    int rc = begin_frame(c, tape, owner);
    if (rc) return rc;
    if (!mat) return mpr::set_error(MPR_ERR_INVALID, "null matrix");
    const int S = c->S;
    hipStream_t s = c->stream;
    unsigned long long* cnt = (c->flags & MPR_CTX_COUNTERS) ? c->counters : nullptr;
    float* const heat = c->heat_frame ? c->heat : nullptr;
    if (heat) HIP_TRY(hipMemsetAsync(heat, 0, (size_t)S * S * sizeof(float), s));
    const int nslots = std::max(tape->num_slots, 1);
    const int choice_cap = std::min(tape->num_choices, (int)MPR_MAX_CHOICES);
    if (tape->num_slots > MPR_KERNEL_SLOTS) c->last.slots_exceeded = 1;

    /* device limits: LDS per workgroup */
    const size_t lds_limit = 160 * 1024;
    if (mprk::tile_stage_lds_bytes(nslots, choice_cap) > lds_limit || mprk::normals_lds_bytes(nslots) > lds_limit ||
        mprk::voxel_lds_bytes(nslots) > lds_limit)
        return mpr::set_error(MPR_ERR_UNSUPPORTED, "tape needs more LDS than one workgroup can hold");

    /* ONE launch resets the images (src/context.cu:1146-1151, :1295-1301: five cudaMemsetAsync), sets
     * *tape_index (:1137) and writes the first tile list (preload_tiles): 3-D frames clear the filled
     * images and the normals, 2-D frames the filled images only */
    const size_t zero_words = (dim == 3) ? c->arena_words : c->arena_images_words;

    int stage_list[3];
    int nstages = 0;
    if (brute) nstages = 0;
    else if (dim == 3) { stage_list[0] = 0; stage_list[1] = 1; stage_list[2] = 2; nstages = 3; }
    else { stage_list[0] = 0; stage_list[1] = 2; nstages = 2; }

    int count;
    /* the choice array of a stage is sized by what the previous stage reports its tapes can need (an
     * upper bound counted by the backward walks), not by the root tape's min / max count: with 488 of
     * them architecture fits two waves per CU in its last tile stage, with the 58 it needs, three */
    int stage_choice_cap = choice_cap;
    mpr_context::FrameKey key;
    key.dim = dim;
    key.rank = rank;
    key.parted = owner != nullptr;
    key.z = z;
    std::memcpy(key.mat, mat, (size_t)(dim == 3 ? 16 : 9) * sizeof(float));
    /* the reference's way (every stage from the 64 px tiles down, every tape pushed): asked for, or a frame that is inspected */
    const bool reference = c->reference_frames || c->force_reference || brute || cnt || heat;
    int hint = (c->hint_serial == tape->serial && c->hint_dim == dim) ? c->hint_mode : (int)mpr_context::HINT_UNKNOWN;
    /* 3-D: the 64^3 stage of a frame up to 1024^3 is 64 wavefronts walking the whole tape one clause after the other — 0.15 ms
     * of latency on an idle chip — while ALL of its 16^3 tiles are one round of wavefronts for the next stage.  A
     * frame nobody inspects starts there: every 64^3 tile counts as ambiguous.  The hierarchy is conservative at every level, so
     * the heights and normals are the same; tile lists and tapes of the first two stages are not the reference's, and a reader gets
     * the frame again the reference's way (ensure_reference_frame).  Tapes with wide DAGs keep the level-parallel first stage. */
    const int t16 = S / 16;
    bool skip0 = !reference && c->skip_stage0 && dim == 3 && !(c->wide_stage0 && c->sched_ok) && c->cus > 0 &&
                 (long long)t16 * t16 * t16 / 64 <= 16ll * c->cus && c->normals_asm && c->tiles_asm;
    /* ... which is the reference's procedure only while its interval routines are inclusion-isotone (the children decide by
     * themselves what the 64^3 tile would have decided for them): not where a NaN end or log's zero bound takes over — a
     * shape that leaves a function's domain somewhere in the view keeps the 64^3 stage (frame_domain.hpp) */
    bool tame = false;
    if (!reference && skip0) {                /* (the only thing the answer decides: whether the shortcut is verified) */
        if (!c->tame_check) {
            tame = true;
        } else {
            const bool known = c->tame_serial == tape->serial && c->tame_key.dim == key.dim && c->tame_key.z == key.z &&
                               std::memcmp(c->tame_key.mat, key.mat, sizeof(key.mat)) == 0;
            if (!known) {
                c->tame_value = mpr::frame_is_tame(tape->clauses.data(), (int)tape->clauses.size(), dim, mat, z);
                c->tame_serial = tape->serial;
                c->tame_key = key;
            }
            tame = c->tame_value;
        }
    }
    /* not tame (bear: the exp / log blends of its far tiles underflow in every frame): the shortcut is taken and VERIFIED — the
     * 64^3 tiles are walked beside the frame and every 16^3 tile is held against its parent (kernels.hpp); a tape whose last
     * verified frame failed keeps the 64^3 stage for a while */
    bool skip0_checked = false;
    if (skip0 && !tame && c->skip0_verify) {
        const bool can = c->gen_ok && c->gen_nchoices <= 64 && !(c->debug_tiles & 3) && c->tile_gen == 1 &&
                         mprk::tile_stage_gen_possible(nslots, c->pool_cap, !c->tiles_asm, c->tiles_vgpr, c->debug_tiles);
        auto veto = c->skip0_veto.find(tape->serial);
        const bool vetoed = veto != c->skip0_veto.end() && veto->second.left > 0;
        if (vetoed) --veto->second.left;
        if (!can || vetoed) skip0 = false;
        else skip0_checked = true;
    }
    /* ... and the stages above the last one leave records only, where everything behind them takes records: the tape's hint says its last stage
     * pushes nothing (group form), the float pass and the normals pass run the root tape's code.  Every 32nd frame of the tape is
     * an ordinary one (its last stage's sample keeps the hint honest); a frame that finds a later stage in need of tapes after all
     * starts over and the tape stays off this path */
    bool sample_groups = true;
    if (!reference && hint == mpr_context::HINT_TAPES && c->measure_len_forced < 0 && !c->debug_choices && !c->groups_always) {
        if (c->tapes_hint_serial != tape->serial) {
            c->tapes_hint_serial = tape->serial;
            c->tapes_hint_frames = 0;
        }
        sample_groups = (c->tapes_hint_frames++ % 32u) == 31u;
    }
    bool lean_first = false;
    if (!reference && dim == 3 && nstages == 3 && c->lean_first && hint == mpr_context::HINT_GROUPS && c->lean_first_veto != tape->serial && c->gen_ok && c->tile_gen == 1 &&
        !(c->wide_stage0 && c->sched_ok) && c->tiles_asm && c->tiles_vgpr && (skip0 || c->tile_gen_chain) &&
        c->tile_gen_last && c->normals_gen && c->normals_asm && c->voxel_gen && c->gen_vox_dw > 0 && c->voxel_groups && c->gen_nchoices <= 64 &&
        !(c->debug_tiles & 3) && c->measure_len_forced < 0 && !c->debug_choices) {
        if (c->lean_first_serial != tape->serial) {
            c->lean_first_serial = tape->serial;
            c->lean_first_frames = 0;
        }
        lean_first = (c->lean_first_frames++ % 32u) != 31u;
    }
    /* a reader's reference frame of a partitioned context keeps the columns other ranks sent (mpr_unpack_*): only this rank's
     * columns are cleared */
    const bool keep_foreign = c->force_reference && owner != nullptr;
    const bool tiles_only = c->tiles_only && c->force_reference && !brute && !cnt && !heat;
    if (keep_foreign) mprk::launch_zero_owned(s, c->arena, tiles_only ? 3 : dim == 3 ? 5 : 4, S, c->owner_dev, rank);
    const size_t zero_now = keep_foreign ? 0 : tiles_only ? (size_t)(c->filled[3] - c->arena) : zero_words;
    if (!brute) {
        const int t0 = S / 64;
        count = t0 * t0 * (dim == 3 ? t0 : 1);
        rc = ensure_tiles(c, 0, (size_t)count);
        if (rc) return rc;
        if (skip0) {
            rc = ensure_tiles(c, 1, (size_t)count * 64);
            if (rc) return rc;
        }
        TimedScope ts(c, "preload_tiles");
        mprk::launch_begin_frame(s, c->arena, zero_now, c->tape_index, (int)tape->clauses.size(), c->num_active,
                                 c->tiles[0], count, t0 * t0, owner ? c->owner_dev : nullptr, rank, skip0 ? c->tiles[1] : nullptr, t0);
        c->tiles_n[0] = (size_t)count;
        if (skip0_checked) {
            /* the 64^3 tiles' own walk, beside the frame: behind whatever the frame's stream has uploaded (the tape, its code) */
            rc = ensure_buffer(&c->skip0_parents, &c->skip0_parents_cap, (size_t)count * mprk::SKIP0_INFO_U64);
            if (rc) return rc;
            rc = ensure_buffer(&c->skip0_children, &c->skip0_children_cap, (size_t)count * 64 * mprk::SKIP0_INFO_U64);
            if (rc) return rc;
            *reinterpret_cast<volatile int*>(c->skip0_flag_host) = 0;
            mprk::Skip0ParentsArgs& pa = f.skip0_args;
            pa.tape_ro = c->pool;
            pa.gen_fwd2_first = c->gen_iw_dw[0][0] ? c->gen_code + c->gen_iw_at[0][0] : nullptr;
            pa.gen_fwd2_below = c->gen_iw_dw[1][0] ? c->gen_code + c->gen_iw_at[1][0] : nullptr;
            pa.parents = c->skip0_parents;
            pa.count = count;
            pa.tps = t0;
            std::memcpy(pa.mat, mat, sizeof(pa.mat));
            /* launched BEHIND the frame's first stage (launch_skip0_side): in front of it its 64 wavefronts took 64 of the 4096 places that
             * stage's wavefronts fill — 16 a compute unit, all at once — and the 64 left over started when the first of the others ended:
             * a second round for one wavefront in 64, half of the stage's 99 us */
            f.skip0_side_pending = true;
        }
        if (skip0) {
            c->last.tiles_in[0] = count;
            c->last.tiles_active[0] = count;
            count *= 64;
            c->tiles_n[1] = (size_t)count;
        }
    } else {
        const int t8 = S / 8;
        count = t8 * t8;
        rc = ensure_tiles(c, 3, (size_t)count);
        if (rc) return rc;
        TimedScope ts(c, "preload_tiles");
        mprk::launch_begin_frame(s, c->arena, zero_now, c->tape_index, (int)tape->clauses.size(), c->num_active,
                                 c->tiles[3], count, count, nullptr, 0);
        c->tiles_n[3] = (size_t)count;
    }
    f.S = S; f.s = s; f.cnt = cnt; f.heat = heat; f.nslots = nslots; f.choice_cap = choice_cap;
    for (int k = 0; k < 3; ++k) f.stage_list[k] = stage_list[k];
    f.nstages = nstages; f.reference = reference; f.skip0 = skip0; f.tiles_only = tiles_only; f.key = key; f.skip0_checked = skip0_checked; f.lean_first = lean_first; f.sample_groups = sample_groups;
    f.count = count; f.stage_choice_cap = stage_choice_cap; f.hint = hint;
    return MPR_OK;
}

/* does stage si (not level-parallel) run the tape's loose first-stage walk (mpr_tape::big_fwd)?  The first stage of a frame nobody
 * reads, every tile on the root tape, in the kernel that keeps 93 slots in registers */
static bool stage_takes_big_walk(const Frame& f, int si, bool wide_now)
{
    const mpr_context* const c = f.c;
    return si == 0 && !f.skip0 && !wide_now && !f.reference && c->big_ok && f.count > 0 && !f.heat && !f.cnt && !(c->debug_tiles & 3) &&
           c->tape_serial == f.tape->serial && f.tape->loose_ok && c->tile_gen_loose &&
           mprk::tile_stage_big_possible(f.nslots, f.stage_choice_cap, c->pool_cap, !c->tiles_asm, c->tiles_vgpr, c->debug_tiles);
}

/* Tile stages on the root tape's generated code (tile_gen.hpp): which code stage si runs, which records it reads and writes */
static int stage_pick_code(Frame& f, int si, int i, bool last, bool wide_now, bool try_lean, mprk::TileStageArgs& a)
{
    FRAME_LOCALS;
    int rc;
    const int count = f.count;
    bool& decisions_recorded = f.decisions_recorded;
    bool& presence_recorded = f.presence_recorded;
    bool& last_recorded = f.last_recorded;
    {
        /* Tile stages on the root tape's generated code (tile_gen.hpp).  The first stage: every tile walks that tape.  A stage
         * below walks its parents' tapes as the same code with the parents' recorded decisions imposed, and — where it pushes —
         * shortens them by the backward code that follows the parent's tape clause by clause (records with presence bits). */
        const bool gen_here = c->gen_ok && !wide_now && !heat && !(c->debug_tiles & 3) && count > 0 &&
                              mprk::tile_stage_gen_possible(nslots, c->pool_cap, !c->tiles_asm, c->tiles_vgpr, c->debug_tiles);
        const bool first_stage = si == (skip0 ? 1 : 0);
        const bool records = dim == 3 && nstages == 3 && c->tile_gen == 1 && c->normals_asm && !cnt;
        const uint32_t* const code_full = c->gen_full_dw ? c->gen_code + c->gen_fwd_dw + c->gen_bwd_dw + c->gen_deriv_dw : nullptr;
        const bool chain = records && c->tile_gen_chain && code_full != nullptr;
        auto record_into = [&](int k) -> int {
            const int e = ensure_buffer(&c->gen_dec[k], &c->gen_dec_cap[k], (size_t)count * mprk::GEN_RECORD_U64 + 16);
            if (e == MPR_OK) a.gen_decisions = c->gen_dec[k];
            return e;
        };
        if (gen_here && first_stage && !last) {
            a.gen_fwd = c->gen_code;
            a.gen_words = c->gen_words;
            a.gen_nchoices = c->gen_nchoices;
            decisions_recorded = false;
            presence_recorded = false;
            if (f.lean_first && records) {
                /* nobody will walk this stage's tapes: records only */
                a.gen_forward_only = true;
                rc = record_into(i);
                if (rc) return rc;
                decisions_recorded = true;
            } else if (skip0) {
                /* the stage below pushes nothing: the decisions are all it and the normals pass need */
                a.gen_bwd = c->tile_gen == 2 ? nullptr : c->gen_code + c->gen_fwd_dw;
                if (records && a.gen_bwd && (c->normals_gen || c->tile_gen_last)) {
                    rc = record_into(i);
                    if (rc) return rc;
                    decisions_recorded = true;
                }
            } else if (chain) {
                a.gen_bwd_full = code_full;
                rc = record_into(i);
                if (rc) return rc;
                decisions_recorded = presence_recorded = true;
            } else {
                a.gen_bwd = c->tile_gen == 2 ? nullptr : c->gen_code + c->gen_fwd_dw;
            }
        } else if (gen_here && !first_stage && !last && decisions_recorded && f.lean_first && records) {
            /* ... and so does a stage between the first and the last one: the root tape's forward walk with the parents' decisions
             * imposed (jumping over what they left dead), the tile's record = its parent's decisions and its own */
            a.gen_fwd = c->gen_code;
            a.gen_guarded = c->tile_gen_guards;
            a.gen_words = c->gen_words;
            a.gen_nchoices = c->gen_nchoices;
            a.gen_parent = c->gen_dec[stage_list[si - 1]];
            a.gen_forward_only = true;
            rc = record_into(i);
            if (rc) return rc;
        } else if (gen_here && !first_stage && !last && presence_recorded && chain) {
            a.gen_fwd = c->gen_code;
            a.gen_words = c->gen_words;
            a.gen_nchoices = c->gen_nchoices;
            a.gen_parent = c->gen_dec[stage_list[si - 1]];
            a.gen_bwd_full = code_full;
            rc = record_into(i);
            if (rc) return rc;
        } else if (gen_here && last && si == 2 && decisions_recorded && try_lean && c->tile_gen_last) {
            /* (pushes nothing: the walk that jumps over what the parent's decisions left dead; the groups of its sample take
             * the interpreter) */
            a.gen_fwd = c->gen_code;
            a.gen_guarded = c->tile_gen_guards;
            a.gen_words = c->gen_words;
            a.gen_nchoices = c->gen_nchoices;
            a.gen_parent = c->gen_dec[1];
        } else if (gen_here && last && si == 2 && presence_recorded && chain && !try_lean) {
            /* a last stage that pushes (a frame whose tiles and tapes are read): every tile gets a record of its own */
            a.gen_fwd = c->gen_code;
            a.gen_words = c->gen_words;
            a.gen_nchoices = c->gen_nchoices;
            a.gen_parent = c->gen_dec[1];
            a.gen_bwd_full = code_full;
            rc = record_into(2);
            if (rc) return rc;
            last_recorded = true;
        } else if (!last) {
            decisions_recorded = presence_recorded = false;       /* an interpreted stage keeps no record: the chain ends */
        }
        if (f.lean_first && count > 0 && (first_stage ? !a.gen_forward_only : !(a.gen_fwd && a.gen_parent && !a.gen_bwd_full))) {
            /* the first stage left (or would leave) records and no tapes, and this stage cannot do with records (a last stage
             * of few tiles runs level-parallel, on the tiles' own tapes): the frame again, its first stage pushing */
            c->lean_first_veto = tape->serial;
            return FRAME_AGAIN;
        }
    }
    {
        /* frames nobody reads owe the reference heights and normals, not tile occupancy: sound but wider exp / log enclosures */
        const bool verified_stage = f.skip0_checked && si == 1;
        if (verified_stage) {
            if (!a.gen_fwd || a.gen_parent) {          /* (not expected: frame_begin asked the same questions) */
                c->skip0_veto[tape->serial].left = 64;
                return FRAME_AGAIN;
            }
            a.self_info = c->skip0_children;
        }
        /* (not the stage that is held against its exact parents: a looser child decides less than they did) */
        a.gen_loose = a.gen_fwd != nullptr && !reference && c->tile_gen_loose && tape->loose_ok && !verified_stage;
        if (a.gen_fwd) {
            /* the scheduled walk of the kind this stage needs (interval_gen.hpp); a loose one falls back on the exact one of its kind */
            const int kind = !a.gen_parent ? mpr::IW_FIRST : a.gen_guarded ? mpr::IW_BELOW_GUARDED : mpr::IW_BELOW;
            if (c->gen_iw_dw[kind][0] <= 0) return mpr::set_error(MPR_ERR_INVALID, "internal: a tile stage planned on generated code the tape has none of");
            a.gen_fwd2_exact = c->gen_code + c->gen_iw_at[kind][0];
            a.gen_fwd2 = a.gen_loose && c->gen_iw_dw[kind][1] > 0 ? c->gen_code + c->gen_iw_at[kind][1] : a.gen_fwd2_exact;
            if (a.gen_fwd2 == a.gen_fwd2_exact) a.gen_loose = false;
            if (a.gen_loose && count > 0) f.used_loose = true;
            a.gen_redo_count = c->redo_count;
        }
        /* a first stage of a tape beyond the generators' 24 slots / 64 min / max clauses, on the kernel with 93 slots in registers: the
         * tape's loose forward walk in front of the interpreter's backward walk (interval_gen.hpp: IW_FIRST_MASKS) — frames nobody reads */
        const bool big_here = stage_takes_big_walk(f, si, wide_now) && !a.gen_fwd;
        if (big_here) {
            a.big_fwd = c->big_code;
            a.big_end = c->big_end;
            a.big_nchoices = tape->num_choices;
            if (c->big_bwd_ok && c->tile_gen_big_bwd) a.big_bwd = c->big_code + c->big_bwd_at;
            a.gen_redo_count = c->redo_count;
            f.used_loose = true;
        }
        /* ... and for tapes the loose arithmetic does not take (asin / acos / atan: the gears): the interpreter's forward walk — the
         * reference's enclosures — and the generated backward walk behind it (it reads the choices where either forward walk puts them,
         * and pushes the interpreter's tapes word for word) */
        const bool bwd_only_here = !big_here && si == 0 && !f.skip0 && !wide_now && !reference && c->big_bwd_ok && c->tile_gen_big_bwd && count > 0 && !heat &&
                                   !cnt && !(c->debug_tiles & 3) && c->tape_serial == tape->serial && !a.gen_fwd &&
                                   mprk::tile_stage_big_possible(f.nslots, f.stage_choice_cap, c->pool_cap, !c->tiles_asm, c->tiles_vgpr, c->debug_tiles);
        if (bwd_only_here) {
            a.big_bwd = c->big_code + c->big_bwd_at;
            a.big_end = c->big_end;
        }
        /* what this stage runs, for mpr_ctx_tile_stage_forms (tests assert the path they mean to exercise) */
        if (si == (skip0 ? 1 : 0)) c->stage_forms.clear();
        std::string f = count <= 0 ? "none" : wide_now ? "wide" : big_here ? (a.big_bwd ? "loosefwd+genbwd" : "interp+loosefwd") : bwd_only_here ? "interp+genbwd" : !a.gen_fwd ? "interp" : "gen";
        if (a.gen_fwd && count > 0) {
            if (a.gen_parent) f += "/parent";
            if (a.gen_guarded) f += "+guards";
            if (a.gen_loose) f += "+loose";
            f += a.gen_bwd_full ? "+bwd_full" : a.gen_bwd ? "+bwd" : a.gen_forward_only ? "+fwdonly" : "";
            if (a.gen_decisions) f += "+records";
        }
        if (!c->stage_forms.empty()) c->stage_forms += " ";
        c->stage_forms += std::to_string(i) + ":" + f;
    }
    return MPR_OK;
}

/* the stage's arguments and its launch: level-parallel (one workgroup per tile) or 64 sibling tiles per wavefront */
static int stage_launch(Frame& f, int si, int i, int tps, bool last, bool wide_now, bool groups_now, bool try_lean, mprk::TileStageArgs& a)
{
    FRAME_LOCALS;
    int rc;
    const int count = f.count, stage_choice_cap = f.stage_choice_cap, hint = f.hint;
    const bool dynamic_choices = c->dynamic_choices;
    a.groups = groups_now ? c->groups : nullptr;
    a.choice_masks = groups_now ? c->choice_masks : nullptr;
    a.tape_ro = c->pool;
    a.tape_wr = c->pool;
    a.tape_index = c->tape_index;
    a.pool_cap = c->pool_cap;
    a.image = c->filled[i];
    a.tps = tps;
    a.tiles = c->tiles[i];
    a.count = count;
    a.nslots = nslots;
    a.choice_cap = dynamic_choices ? stage_choice_cap : choice_cap;
    a.next_choices = dynamic_choices ? c->num_active + 4 : nullptr;
    a.len_stats = groups_now ? c->num_active + 5 : nullptr;
    if (c->debug_choices && si == nstages - 2) a.len_stats = c->num_active + 5;
    a.no_push = try_lean;
    {
        /* the sample the stage measures its tapes on (a.len_stats): about a sixteenth of the groups while nothing is known
         * about this tape; a sixty-fourth afterwards, and only where the launch is long enough to hide the sample's longer
         * waves (a wrong hint costs time, never a pixel); everything in frames that push anyway */
        const int ng = (count + 63) / 64;
        a.measure_at[0] = ng / 4;
        a.measure_at[1] = ng / 2;
        if (!try_lean || hint == mpr_context::HINT_UNKNOWN) a.measure_len = std::max(ng / 32, std::min(ng, 4));
        else a.measure_len = ng >= 32 * std::max(c->cus, 1) ? ng / 128 : 0;
        if (c->measure_len_forced >= 0) a.measure_len = c->measure_len_forced;
        if (f.lean_first) a.measure_len = 0;           /* (the sample's groups walk their parents' tapes: there are none) */
        if (a.measure_len == 0 && groups_now) a.len_stats = nullptr;
    }
    a.compiled_walk = !c->tiles_asm;
    a.vgpr_slots = c->tiles_vgpr;
    a.z = z;
    fill_mat(a.mat, mat, dim == 3 ? 16 : 9);
    a.counters = cnt;
    a.heat = heat;
    a.heat_stride = S;
    a.debug = c->debug_tiles;
    if (a.debug & 4) a.debug |= si << 4;
    if ((c->debug_tiles & 8) && last) a.debug |= 1;          /* 8: skip tape pushing in the last stage only */
    if (heat && dim == 3) mprk::launch_mask_filled(s, c->tiles[i], count, tps, c->filled[i]);
    TimedScope ts(c, "eval_tiles_i");
    /* ... and only while the stage has few tiles (a workgroup per tile is latency-bound at low lane
     * utilisation: with 32768 tiles at 2048^3 the 64-tiles-per-wave walk is 1.5x faster) */
    if (wide_now) {
        /* few tiles -> one workgroup per tile, level by level over the root tape's DAG */
        mprk::WideStageArgs w;
        w.t = a;
        w.recs = c->sched_recs;
        w.level_start = c->sched_levels;
        w.nlevels = c->sched_nlevels;
        w.nclauses = c->sched_nclauses;
        w.root_val = c->sched_root;
        w.root_tape = 0;
        w.wpt = (c->sched_nclauses + 15) / 16;
        w.prev_writer = c->sched_prev;
        w.defs = c->sched_defs;
        w.bits_in = si == 0 ? nullptr : c->wide_bits[(si - 1) & 1];
        w.bits_out = nullptr;
        if (!last && c->wide_later != 0) {          /* the next stage may run this way too */
            rc = ensure_buffer(&c->wide_bits[si & 1], &c->wide_bits_cap[si & 1], (size_t)count * (size_t)w.wpt);
            if (rc) return rc;
            w.bits_out = c->wide_bits[si & 1];
        }
        mprk::launch_eval_tiles_wide(s, dim, w, c->wide_threads);
    } else if (a.gen_fwd2 && a.gen_loose && a.gen_fwd2 != a.gen_fwd2_exact && c->tile_gen_lean && dim == 3 && !cnt && !heat && !a.self_info &&
               !a.gen_bwd && !a.gen_bwd_full && !(a.debug & 36) && (a.gen_forward_only || (a.gen_parent && a.no_push && a.groups))) {
        /* a loose stage that pushes nothing: six wavefronts per SIMD (kernels.hip: k_eval_tiles<.., LEAN>); what its walks hand back
         * — a walk that asks for the exact code, a group of the stage's sample — the ordinary kernel runs behind it */
        const size_t nwg = (size_t)(count + 63) / 64;
        if (nwg > c->redo_flags_cap) {
            if (c->redo_flags) (void)hipFree(c->redo_flags);
            c->redo_flags = nullptr;
            c->redo_flags_cap = 0;
            HIP_TRY(hipMalloc((void**)&c->redo_flags, nwg * 2 + 64));
            c->redo_flags_cap = nwg * 2 + 64;
        }
        a.lean = 1;
        /* the last stage: with a second, tight enclosure where the tape has a sin / cos (interval_gen.hpp) — the tiles it decides
         * stay out of the float pass */
        const int kind = !a.gen_parent ? mpr::IW_FIRST : a.gen_guarded ? mpr::IW_BELOW_GUARDED : mpr::IW_BELOW;
        const bool tight_here = last && a.groups && a.no_push && c->tile_tight && c->gen_iw_dw[kind][2] > 0;
        if (tight_here) {
            a.lean = 2;
            a.gen_fwd2 = c->gen_code + c->gen_iw_at[kind][2];
        }
        a.redo_flags = c->redo_flags;
        if (!mprk::launch_eval_tiles(s, dim, a))        /* (the launch behind it reads flags only this kernel writes: ADVICE r5) */
            return mpr::set_error(MPR_ERR_INVALID, "internal: a lean tile stage planned on generated code ran another kernel");
        a.lean = 0;
        a.redo_flags = nullptr;
        a.only_flagged = c->redo_flags;
        a.gen_fwd2 = a.gen_fwd2_exact;
        (void)mprk::launch_eval_tiles(s, dim, a);
        c->stage_forms += tight_here ? "+lean+tight" : "+lean";
    } else {
        const bool ran_gen = mprk::launch_eval_tiles(s, dim, a);
        if (a.gen_fwd && !ran_gen)          /* the records this frame counts on would not exist */
            return mpr::set_error(MPR_ERR_INVALID, "internal: a tile stage planned on generated code ran another kernel");
        /* a last stage that pushed its tapes (a frame somebody reads, or whose float pass wants per-tile tapes) on the root tape's generated
         * code, the float pass on that code too: the second verdict in a launch of its own (TileStageArgs::verdict_only) — the tiles it
         * decides stay in the reference's lists and skip the float pass's walk */
        const int tk = c->gen_iw_dw[mpr::IW_BELOW_GUARDED][2] > 0 ? mpr::IW_BELOW_GUARDED : mpr::IW_BELOW;
        /* (not in a frame rendered again FOR a reader or for MPR_CTX_PARANOID's comparison: that one is the reference's way and nothing else) */
        if (ran_gen && last && dim == 3 && a.gen_parent && a.gen_bwd_full && a.groups && c->tile_tight && c->tile_gen_lean && !c->force_reference && !cnt && !heat && !(a.debug & 36) &&
            c->gen_iw_dw[tk][2] > 0 && tape->loose_ok && c->tile_gen_loose) {
            int rc2 = ensure_buffer(&c->tight_skip, &c->tight_skip_cap, (size_t)count + 64);
            if (rc2) return rc2;
            mprk::TileStageArgs v = a;
            v.lean = 2;
            v.verdict_only = true;
            v.tight_skip = c->tight_skip;
            if (c->voxel_fp) {
                const size_t cells = (size_t)a.tps * a.tps;
                rc2 = ensure_buffer(&c->tight_image, &c->tight_image_cap, cells);
                if (rc2) return rc2;
                HIP_TRY(hipMemsetAsync(c->tight_image, 0, cells * sizeof(int), s));
                v.tight_image = c->tight_image;
            }
            v.gen_fwd2 = c->gen_code + c->gen_iw_at[tk][2];
            v.gen_guarded = tk == mpr::IW_BELOW_GUARDED;
            v.redo_flags = nullptr;
            v.only_flagged = nullptr;
            v.len_stats = nullptr;
            v.self_info = nullptr;
            v.gen_redo_count = nullptr;
            v.next_choices = nullptr;
            if (mprk::launch_eval_tiles(s, dim, v)) {
                f.tight_skip_valid = true;
                c->stage_forms += "+verdict";
            }
        }
    }
    return MPR_OK;
}

/* the 64^3 tiles' own walk, beside the frame: behind whatever the frame's stream has uploaded (the tape, its code) */
static int launch_skip0_side(Frame& f)
{
    mpr_context* const c = f.c;
    if (!f.skip0_side_pending) return MPR_OK;
    f.skip0_side_pending = false;
    if (c->side_must_wait) {
        HIP_TRY(hipStreamWaitEvent(c->side, c->ev_begin, 0));
        c->side_must_wait = false;
    }
    /* ... and not beside that stage but behind it: side by side its 64 wavefronts ask for the same issue slots as the stage's 4096, one
     * round of latency-bound exact walks both; beside the compaction and the last tile stage nobody misses them (its result is the
     * normals pass's business, 400 us later) */
    HIP_TRY(hipEventRecord(c->ev_stage, f.s));
    HIP_TRY(hipStreamWaitEvent(c->side, c->ev_stage, 0));
    mprk::launch_skip0_parents(c->side, f.skip0_args);
    HIP_TRY(hipEventRecord(c->ev_check, c->side));
    return MPR_OK;
}

/* one tile stage: evaluation, compaction (+ copy_filled), the survivor count */
static int frame_tile_stage(Frame& f, int si)
{
    FRAME_LOCALS;
    int rc;
    int& count = f.count;
    int& stage_choice_cap = f.stage_choice_cap;
    int& hint = f.hint;
    bool& prev_wide = f.prev_wide;
    bool& decisions_recorded = f.decisions_recorded;
    bool& vox_gen_planned = f.vox_gen_planned;
    bool& group_form = f.group_form;
    bool& lean_now = f.lean_now;
    int& group_stage = f.group_stage;
    int& group_count = f.group_count;
    int& group_cap = f.group_cap;
    const bool dynamic_choices = c->dynamic_choices;
    {
        const int i = stage_list[si];
        const bool last = (si == nstages - 1);
        const int next = (dim == 3) ? i + 1 : (i ? 3 : 2);
        const int sub = (dim == 3) ? 4 : 8;
        const int tile_size_px = (dim == 3) ? (64 >> (2 * i)) : (i ? 8 : 64);
        const int tps = S / tile_size_px;
        c->last.tiles_in[si] = count;

        /* float pass in group form: the last tile stage also writes, per group of 64 siblings, the tape it walked and
         * the tiles' min / max decisions; possible while a tape records at most 128 of them (two register pairs per side) */
        const int stage_cap = dynamic_choices ? stage_choice_cap : choice_cap;
        /* level-parallel kernel: the first stage while it has few tiles, later ones while the stage before ran that way */
        /* ... later ones: a workgroup walks the levels in about the time a wavefront of the serial kernel walks a fifth of the
         * tape, but only 160 KB / LDS-per-workgroup of them fit a CU: worth it for up to about two rounds of workgroups
         * (measured: prospero 256^2 0.48 -> 0.20 ms, gears 512^2 0.35 -> 0.11, architecture 256^3 second stage 0.25 -> 0.12;
         * prospero 512^2 at eight rounds 0.23 -> 0.49) */
        int wide_limit = c->wide_later;
        if (wide_limit < 0) {
            const size_t per_cu = std::min<size_t>(8, std::max<size_t>(1, ((size_t)160 << 10) / mprk::wide_stage_lds_bytes(c->sched_nclauses)));
            wide_limit = 2 * std::max(c->cus, 1) * (int)per_cu;
        }
        /* (a rank's first stage holds the whole frame's tiles, the other ranks' dead: what counts is its own — architecture 2048^3 dealt to
         * eight ranks is 4096 tiles each, a level-parallel stage, not 64 wavefronts stepping through 1296 clauses twice: 0.29 ms of a
         * rank's 0.68, round 5) */
        int first_stage_tiles = count;
        if (si == 0 && owner && !c->owner_host.empty())
            first_stage_tiles = (int)((long long)std::count(c->owner_host.begin(), c->owner_host.end(), rank) * (count / (long long)c->owner_host.size()));
        /* (a first stage of a couple of thousand tiles and more: the tape's loose walk as generated code, 64 tiles to the wavefront, is
         * through before the level-parallel kernel has walked its workgroup-per-tile: scripts/probe_big.py) */
        const bool big_first = si == 0 && first_stage_tiles >= c->big_min_tiles && stage_takes_big_walk(f, si, false);
        const bool wide_now = count > 0 && c->wide_stage0 && c->sched_ok && !heat && !(c->debug_tiles & 11) && !(c->flags & MPR_CTX_SERIAL_STAGES) && !big_first &&
                              (si == 0 ? first_stage_tiles <= 8192 : (prev_wide && count <= wide_limit));
        const bool groups_now = last && count > 0 && !wide_now && f.sample_groups && c->voxel_jit && c->voxel_asm && c->voxel_groups && !cnt && !heat && c->cus > 0 &&
                                mprk::jit_slot_class(nslots) != 0 && stage_cap <= mprk::jit_max_choices();
        /* no tapes from this stage unless the frame is inspected or this tape's last measurement said they pay */
        const bool try_lean = groups_now && !reference && hint != mpr_context::HINT_TAPES && (dim == 2 || c->normals_asm);
        if (groups_now) {
            const size_t ng = ((size_t)count + 63) / 64;
            rc = ensure_buffer(&c->groups, &c->groups_cap, ng);
            if (rc) return rc;
            rc = ensure_buffer(&c->choice_masks, &c->masks_cap, ng * (size_t)std::max(stage_cap, 1));
            if (rc) return rc;
            rc = ensure_buffer(&c->group_alive, &c->group_alive_cap, ng);
            if (rc) return rc;
            rc = ensure_buffer(&c->group_list, &c->group_list_cap, ng + 1);
            if (rc) return rc;
            group_form = true;
            group_stage = i;
            group_count = count;
            group_cap = std::max(stage_cap, 1);
        }
        mprk::TileStageArgs a;
        rc = stage_pick_code(f, si, i, last, wide_now, try_lean, a);
        if (rc) return rc;
        a.no_mask = c->stage0_only;
        if (count > 0) {
            rc = stage_launch(f, si, i, tps, last, wide_now, groups_now, try_lean, a);
            if (rc) return rc;
            rc = launch_skip0_side(f);
            if (rc) return rc;
            if (a.self_info && !f.blocking) {
                /* a frame that does not block (the multi-GPU pipeline packs its columns behind it on the same stream) must have
                 * its verdict before the float pass is launched: the comparison on the side stream, as soon as both are through */
                HIP_TRY(hipEventRecord(c->ev_stage, s));
                HIP_TRY(hipStreamWaitEvent(c->side, c->ev_stage, 0));
                mprk::launch_skip0_compare(c->side, f.skip0_args, c->skip0_children, c->skip0_flag_dev);
                HIP_TRY(hipEventRecord(c->ev_done, c->side));
                c->side_compare_pending = true;
            }
        }
        if (c->stage0_only) {
            HIP_TRY(hipStreamSynchronize(s));
            return FRAME_STOP;
        }
        /* worst case: every tile survives */
        rc = ensure_tiles(c, next, last ? (size_t)std::max(count, 1) : (size_t)std::max(count, 1) * 64);
        if (rc) return rc;
        const bool zs = dim == 3 && mprk::zsort_supported(tps) && (c->zsort & (last ? 2 : 1));
        int act3[5] = {0, 0, 0, 0, 0};
        /* second mask_filled_tiles + assign_next_nodes + subdivide / copy_active_tiles + copy_filled, then the survivor count */
        /* the float pass on the root tape's host-generated code (voxel_gen.hpp) takes the smallest tiles one by one and looks their
         * decisions up by where they sat in this stage's list */
        const bool vox_gen_next = last && groups_now && dim == 3 && i == 2 && decisions_recorded && c->voxel_gen && c->gen_ok && c->gen_vox_dw > 0 &&
                                  c->cus > 0;
        vox_gen_planned = vox_gen_next;
        /* ... by footprint segments: no list of tiles, the stage's own list in blocks of 64 siblings */
        /* (not in a frame that leaves the reference's state behind: its list of smallest tiles is part of that state) */
        const bool vox_fp_next = vox_gen_next && c->voxel_fp && !reference && count > 0 && (count & 63) == 0;
        f.vox_fp = vox_fp_next;
        /* a frame that leaves the reference's state behind, its tiles through the second verdict (stage_launch: "+verdict"): the list the
         * reference's way AND the segments, for a float pass that stops at the first hidden tile of a footprint (bear 1024^3: 478 -> 3xx us;
         * the extra compaction 14) */
        const bool vox_fp_beside = vox_gen_next && c->voxel_fp && reference && !c->force_reference && f.tight_skip_valid && count > 0 && (count & 63) == 0;
        f.vox_fp_beside = false;
        if (vox_gen_next) {
            if (vox_fp_next || vox_fp_beside) {
                rc = ensure_buffer(&c->fp_items, &c->fp_items_cap, (size_t)count / 4 + 64);
                if (rc) return rc;
                if (!c->fp_meta) {
                    HIP_TRY(hipMalloc((void**)&c->fp_meta, 4 * sizeof(int)));
                    HIP_TRY(hipMemsetAsync(c->fp_meta, 0, 4 * sizeof(int), s));
                }
            }
            if (!vox_fp_next) {
                rc = ensure_buffer(&c->tile_source, &c->tile_source_cap, (size_t)std::max(count, 1));
                if (rc) return rc;
            }
            if (!c->vox_counters) HIP_TRY(hipMalloc((void**)&c->vox_counters, (size_t)mprk::voxel_gen_counter_ints() * sizeof(int)));
        }
        f.vox_counters_cleared = false;
        auto compact = [&](bool mark_groups) -> int {
            const int seq = ++c->pub_seq;
            if (mark_groups && vox_fp_next) {
                TimedScope ts(c, "compact_copy");
                mprk::launch_compact_footprints(s, c->tiles[i], count, tps, c->filled[i], c->num_active, c->fp_items, c->fp_meta, c->vox_counters,
                                                mprk::voxel_gen_counter_lists(), mprk::voxel_gen_counter_ints() / mprk::voxel_gen_counter_lists(), c->pub_dev, seq,
                                                (last && tiles_only) ? nullptr : c->filled[next], S / (tile_size_px / sub), c->tape_index);
                f.vox_counters_cleared = true;
            } else if (zs) {
                TimedScope ts(c, last ? "compact_copy" : "compact_subdivide");
                mprk::launch_compact_zsorted(s, last, c->tiles[i], count, tps, c->filled[i], c->tiles[next],
                                             c->zs_hist, c->zs_cursor, c->pub_dev, seq, (last && tiles_only) ? nullptr : c->filled[next], S / (tile_size_px / sub),
                                             c->num_active + 4, mark_groups ? c->group_alive : nullptr, c->tape_index,
                                             (mark_groups && vox_gen_next) ? c->tile_source : nullptr,
                                             (mark_groups && vox_gen_next) ? c->vox_counters : nullptr, mprk::voxel_gen_counter_ints());
                if (mark_groups && vox_gen_next) f.vox_counters_cleared = true;
            } else {
                TimedScope ts(c, last ? "compact_copy" : "compact_subdivide");
                mprk::launch_compact_subdivide(s, dim, last, c->tiles[i], count, tps, c->filled[i], c->num_active, c->tiles[next],
                                               c->pub_dev, seq, (last && tiles_only) ? nullptr : c->filled[next], S / (tile_size_px / sub),
                                               mark_groups ? c->group_alive : nullptr, c->tape_index,
                                               (mark_groups && vox_gen_next) ? c->tile_source : nullptr);
            }
            if (mark_groups && !vox_gen_next && !vox_fp_next) mprk::launch_list_alive_groups(s, c->group_alive, (count + 63) / 64, c->group_list);
            if (mark_groups && vox_fp_beside) {
                TimedScope ts(c, "compact_copy");
                mprk::launch_footprint_segments(s, c->tiles[i], count, tps, c->filled[i], c->fp_items, c->fp_meta, c->vox_counters, mprk::voxel_gen_counter_lists(),
                                                mprk::voxel_gen_counter_ints() / mprk::voxel_gen_counter_lists(), c->tight_skip, c->tight_image, c->filled[next], S);
                f.vox_counters_cleared = true;
                f.vox_fp_beside = true;
            }
            return read_active(c, seq, act3);       /* the reference's blocking read-back (:1209, :1375) */
        };
        if (count > 0) {
            rc = compact(groups_now);
            if (rc) return rc;
        } else if (!(last && tiles_only)) {
            /* copy_filled rides in the compaction's launch; no compaction, a launch of its own */
            TimedScope ts(c, "copy_filled");
            mprk::launch_copy_filled(s, dim, c->filled[i], c->filled[next], S / (tile_size_px / sub));
        }
        if (groups_now && act3[2] > 0 && !c->groups_always) {
            /* a child evaluated on its group's tape walks the clauses the child's own tape dropped as well: worth it while
             * the tapes walked are less than twice the tapes handed on (measured: bear 1.03x -> float pass 1.63x faster
             * than the interpreter on per-tile tapes, architecture 1.8x -> 1.39x faster, involute_gear_3d 2.8x -> 1.44x
             * slower, involute_gear_2d 4.3x -> 5x slower) */
            /* (which groups fall into the sample varies from frame to frame — list order is the compaction's — so a tape near the
             * threshold is kept where it is: leaving a form takes 15 % more than staying out of it) */
            /* (round 5: with the tile stages' same-address atomic gone a pushing last stage costs what its walks cost, and per-tile
             * tapes win earlier — scripts/ab_frames.py, never / always the group form: architecture 1024^3 0.69 / 0.77 ms, 2048^3
             * 1.56 / 1.71, involute_gear_3d 1.64 / 1.84, hello_world 0.526 / 0.592; bear (1.03x) stays: the threshold 2.0 -> 1.4) */
            /* (a tape whose float walk the host generated — k_eval_voxels_gen, 3.4x the interpreter on bear — keeps 2.0) */
            const double mid = vox_gen_next ? 2.0 : 1.4;
            const double limit = hint == mpr_context::HINT_GROUPS ? mid + 0.2 : hint == mpr_context::HINT_TAPES ? mid - 0.2 : mid;
            const bool pays = (double)act3[2] <= limit * (double)act3[1];
            if (!pays) group_form = false;
            if (!reference) {
                c->hint_serial = tape->serial;
                c->hint_dim = dim;
                c->hint_mode = hint = pays ? mpr_context::HINT_GROUPS : mpr_context::HINT_TAPES;
            }
            if (c->debug_choices) fprintf(stderr, "last stage: tapes handed on %d clauses, tapes walked %d (sample)\n", act3[1], act3[2]);
        }
        lean_now = try_lean && group_form;
        if (try_lean && !group_form && count > 0) {
            /* the stage pushed nothing and the float pass wants per-tile tapes after all (first frame of such a tape: from now on
             * `hint` says so): the stage again, pushing.  Dead tiles stay dead, fills are idempotent, and no tile's tape field was
             * touched by the measuring run, so the second run sees what the first saw minus the tiles the second mask took. */
            a.no_push = false;
            a.len_stats = nullptr;
            a.groups = nullptr;
            a.choice_masks = nullptr;
            a.gen_fwd = nullptr;             /* (pushing a parent's tape takes walking it: the interpreter) */
            a.gen_parent = nullptr;
            {
                TimedScope ts(c, "eval_tiles_i");
                mprk::launch_eval_tiles(s, dim, a);
            }
            rc = compact(false);
            if (rc) return rc;
        }
        if (f.vox_fp_beside) {
            if (vox_gen_planned && group_form) f.vox_fp = true;      /* (the list is there whichever the float pass takes) */
            else f.vox_fp_beside = false;
        }
        if (f.vox_fp && !f.vox_fp_beside && !(vox_gen_planned && group_form)) {
            /* segments were made of the stage's list and the float pass takes the list of tiles after all (per-tile tapes: this frame's
             * sample said they pay, or the stage pushed them anyway): the list */
            if (!(try_lean && !group_form && count > 0)) {          /* (that re-run compacted already) */
                rc = compact(false);
                if (rc) return rc;
            }
            f.vox_fp = false;
        }
        if (count > 0 && (act3[4] & 0x80000000) && c->pool_auto && c->pool_cap < (long long)MPR_NUM_SUBTAPES_BIG * MPR_SUBTAPE_CHUNK) {
            /* a push of this stage found the pool full and its tile kept the parent's tape (the reference's silent fallback):
             * correct images, but not the tapes a larger pool gives.  A pool this context sized itself grows and the frame
             * starts over. */
            HIP_TRY(hipStreamSynchronize(s));
            const long long bigger = std::min<long long>(c->pool_cap * 2, (long long)MPR_NUM_SUBTAPES_BIG * MPR_SUBTAPE_CHUNK);
            uint64_t* fresh = nullptr;
            if (hipMalloc((void**)&fresh, ((size_t)bigger + 128) * sizeof(uint64_t)) == hipSuccess) {
                (void)hipFree(c->pool);
                c->pool = fresh;
                c->pool_cap = bigger;
                c->tape_serial = 0;                  /* the root tape has to be copied in again */
                ++c->pool_growths;
                return FRAME_AGAIN;
            }
            (void)hipGetLastError();                 /* no memory for a larger pool: carry on with the fallback, as the reference would */
        }
        if (count > 0 && c->pool_auto && (long long)(act3[4] & 0x7FFFFFFF) * 4 > c->pool_cap * 3) c->pool_grow_pending = true;
        const int active = act3[0];
        if (count > 0) stage_choice_cap = std::min(choice_cap, std::max(act3[3], 1));
        if (c->debug_choices) fprintf(stderr, "stage %d: %d tiles, reports %d choices for the next stage (root %d); tapes handed on %d, walked %d\n", si, count, act3[3], choice_cap, act3[1], act3[2]);
        c->last.tiles_active[si] = active;
        count = last ? active : active * 64;
        c->tiles_n[next] = (size_t)count;
        prev_wide = wide_now && c->wide_later != 0;
    }
    return MPR_OK;
}

/* eval_voxels_f: the smallest tiles' voxels / pixels */
static int frame_float_pass(Frame& f)
{
    FRAME_LOCALS;
    int rc;
    const int count = f.count;
    const bool group_form = f.group_form, lean_now = f.lean_now, vox_gen_planned = f.vox_gen_planned;
    const int group_stage = f.group_stage, group_count = f.group_count, group_cap = f.group_cap;
    if (count <= 0) return MPR_OK;
    {
        mprk::VoxelArgs v;
        v.tape_ro = c->pool;
        v.image = c->filled[3];
        v.tps = S / (dim == 3 ? 4 : 8);
        v.tiles = c->tiles[3];
        v.count = count;
        v.nslots = nslots;
        v.z = z;
        fill_mat(v.mat, mat, dim == 3 ? 16 : 9);
        v.counters = cnt;
        v.heat = heat;
        v.vgpr_slots = c->tiles_vgpr;
        TimedScope ts(c, "eval_voxels_f");
        /* the assembly interpreter keeps no work counters: instrumented and heatmap frames use the C++ one */
        bool jitted = false, on_root_code = false;
        /* The root tape's host-generated float walk (voxel_gen.hpp), the tiles' decisions as bits: frames whose tile stages kept their
         * tiles' records down to the stage above the last one, whose last stage recorded its groups' masks */
        if (vox_gen_planned && group_form && !brute && f.vox_fp) {
            /* by footprint segments over the last tile stage's own list (kernels_voxel_jit.hip: k_eval_voxels_gen_fp) */
            if (c->fp_grid == 0) c->fp_grid = mprk::voxel_gen_fp_grid(c->cus);
            mprk::VoxelArgs fv = v;
            fv.tiles = c->tiles[group_stage];
            fv.count = group_count;
            if (c->walked_dev) HIP_TRY(hipMemsetAsync(c->walked_dev, 0, 1024 * sizeof(int), s));
            /* segments per claim: one — two in large frames (measured, scripts/walked_probe.py: bear 1024^3 295 us with 1, 306 with 2, 310 with 3,
             * 336 with 4; 2048^3 900 / 897 / 926 with 1 / 2 / 3): a segment is two or three tiles, and 80 000 claims over eight counters are 90 us
             * of atomics beside a 300 us kernel */
            const int run = c->voxel_gen_tiles > 0 ? c->voxel_gen_tiles : std::max(1, std::min(2, count / (24 * std::max(c->fp_grid, 1))));
            mprk::launch_eval_voxels_gen_fp(s, fv, c->gen_code + c->gen_vox_at, c->fp_grid, c->fp_items, c->fp_meta, c->groups, c->choice_masks, group_cap,
                                            c->vox_counters, c->gen_dec[1], c->gen_nchoices, run, c->walked_dev, f.tight_skip_valid ? c->tight_skip : nullptr);
            jitted = on_root_code = true;
        } else if (vox_gen_planned && group_form && !brute) {
            int& grid = c->vox_grid_cache[dim - 2];
            if (grid == 0) grid = mprk::voxel_gen_grid(dim, c->cus);
            const int use_grid = c->voxel_gen_wgs > 0 ? std::min(grid, c->voxel_gen_wgs * c->cus) : grid;
            if (!c->vox_counters) HIP_TRY(hipMalloc((void**)&c->vox_counters, (size_t)mprk::voxel_gen_counter_ints() * sizeof(int)));
            if (!f.vox_counters_cleared)            /* (the z-sorting compaction's scan clears them on the way) */
                HIP_TRY(hipMemsetAsync(c->vox_counters, 0, (size_t)mprk::voxel_gen_counter_ints() * sizeof(int), s));
            if (c->walked_dev) HIP_TRY(hipMemsetAsync(c->walked_dev, 0, 1024 * sizeof(int), s));
            mprk::launch_eval_voxels_gen(s, dim, v, c->gen_code + c->gen_vox_at, use_grid, c->tile_source, c->groups, c->choice_masks, group_cap,
                                         c->vox_counters, c->gen_dec[1], c->gen_nchoices,
                                         /* tiles per claim: about four claims per wavefront — few tiles (a small frame, a rank's eighth of one)
                                          * take short runs, so that no wavefront is left with a whole run while the others have none; at most
                                          * 8 (measured: bear 1024^3 0.816 ms with 8, 0.831 with 4; 256^3 0.128 / 0.113) */
                                         c->voxel_gen_tiles > 0 ? c->voxel_gen_tiles : std::max(1, std::min(8, count / (4 * std::max(use_grid, 1)))), c->walked_dev,
                                         f.tight_skip_valid ? c->tight_skip : nullptr);
            jitted = on_root_code = true;
        }
        if (!cnt && !heat && !jitted) {
            /* every tape as machine code */
            const bool gf = group_form && !brute;
            JitPlan jp;
            rc = jit_prepare(c, tape, dim, nslots, gf, &jp);
            if (rc) return rc;
            if (jp.ok && gf) {
                mprk::VoxelArgs gv = v;
                gv.tiles = c->tiles[group_stage];
                gv.count = group_count;
                mprk::launch_eval_voxels_jit(s, dim, gv, c->jit_code, (uint32_t)jp.region, (int)jp.slot_dw, (int)jp.nslot, jp.grid, (int)tape->clauses.size(), c->groups,
                                             c->choice_masks, group_cap, c->num_active + 7, c->group_list, jp.always_inv);
                jitted = true;
            } else if (jp.ok && !group_form && (brute || c->voxel_jit_tiles)) {
                mprk::launch_eval_voxels_jit(s, dim, v, c->jit_code, (uint32_t)jp.region, (int)jp.slot_dw, 1, jp.grid, (int)tape->clauses.size(), nullptr,
                                             nullptr, 0, nullptr, nullptr);
                jitted = true;
            }
        }
        if (lean_now && !(jitted && group_form)) {
            /* no executable memory after all (the allocation above failed): the tapes this frame did not push are needed — the
             * whole frame again, the reference's way */
            return FRAME_AGAIN_REFERENCE;
        }
        if (on_root_code) {
            snprintf(c->float_kernel, sizeof c->float_kernel, f.vox_fp ? "k_eval_voxels_gen_fp<%d>" : "k_eval_voxels_gen<%d>", dim);
        } else if (jitted) {
            snprintf(c->float_kernel, sizeof c->float_kernel, "k_eval_voxels_jit%s<%d, %d>", group_form && !brute ? "_groups" : "", dim,
                     mprk::jit_slot_class(nslots));
        } else if (c->voxel_asm && !cnt && !heat) {
            mprk::launch_eval_voxels_asm(s, dim, v);
            snprintf(c->float_kernel, sizeof c->float_kernel, "k_eval_voxels_asm<%d>", dim);
        } else {
            mprk::launch_eval_voxels(s, dim, v);
            snprintf(c->float_kernel, sizeof c->float_kernel, "k_eval_voxels<%d>", dim);
        }
    }
    return MPR_OK;
}

/* eval_pixels_d: the normals of the heightmap's pixels */
static int frame_normals_pass(Frame& f)
{
    FRAME_LOCALS;
    const bool group_form = f.group_form, lean_now = f.lean_now, decisions_recorded = f.decisions_recorded, last_recorded = f.last_recorded;
    const int group_stage = f.group_stage, group_cap = f.group_cap;
    {
        mprk::NormalArgs n;
        n.tape_ro = c->pool;
        n.image = c->filled[3];
        n.output = c->normals;
        n.size = S;
        n.nslots = nslots;
        fill_mat(n.mat, mat, 16);
        n.tiles = c->tiles[0];
        n.subtiles = c->tiles[1];
        n.microtiles = c->tiles[2];
        n.counters = cnt;
        n.col_list = nullptr;
        n.ncols = 0;
        /* the groups' tapes and decisions whenever the last stage recorded them and the float pass took them (a frame without
         * last-stage tapes has nothing else; one with them gets the shared walks) */
        const bool normals_on_groups = (lean_now || group_form) && c->normals_asm && !cnt;
        n.groups = normals_on_groups ? c->groups : nullptr;
        n.choice_masks = normals_on_groups ? c->choice_masks : nullptr;
        n.choice_cap = group_cap;
        n.vgpr_slots = c->tiles_vgpr;
        if (c->normals_gen && decisions_recorded && ((normals_on_groups && group_stage == 2) || last_recorded)) {
            n.gen_code = c->gen_code + c->gen_fwd_dw + c->gen_bwd_dw;
            n.gen_code_guarded = c->gen_derivg_dw > 0 ? c->gen_code + c->gen_derivg_at : nullptr;
            n.gen_decisions0 = skip0 ? nullptr : c->gen_dec[0];
            n.gen_decisions = c->gen_dec[1];
            n.gen_decisions2 = last_recorded ? c->gen_dec[2] : nullptr;
            n.gen_nchoices = c->gen_nchoices;
            if (f.skip0_checked) {
                /* the 64^3 tiles' decisions, walked beside the frame (kernels.hpp: NormalArgs::skip0_parents): long there */
                {
                    const int rcs = launch_skip0_side(f);          /* (launched behind the first stage; here for a frame that had none) */
                    if (rcs) return rcs;
                }
                HIP_TRY(hipStreamWaitEvent(s, c->ev_check, 0));
                n.skip0_parents = c->skip0_parents;
            }
        } else if (f.skip0_checked) {
            /* the interpreters walk tapes, and the 16^3 tiles this frame culled carry none of their 64^3 tiles': such a frame's
             * normals are the reference's only where every decision is a fact — the frame again, from the 64^3 tiles down */
            c->skip0_normals_veto = true;
        }
        if (owner && c->normals_asm && !cnt) {
            /* owned columns only; the list is rebuilt when the ownership table or the rank changes */
            if (c->my_cols_rank != rank || c->my_cols_gen != c->owner_gen) {
                std::vector<int> list;
                for (size_t i = 0; i < c->owner_host.size(); ++i) if (c->owner_host[i] == rank) list.push_back((int)i);
                c->my_ncols = (int)list.size();
                if (!list.empty()) HIP_TRY(hipMemcpy(c->col_list_dev, list.data(), list.size() * sizeof(int), hipMemcpyHostToDevice));
                c->my_cols_rank = rank;
                c->my_cols_gen = c->owner_gen;
            }
            n.col_list = c->col_list_dev;
            n.ncols = c->my_ncols;
        }
        TimedScope ts(c, "eval_pixels_d");
        if (c->normals_asm && !cnt) mprk::launch_eval_normals_asm(s, n);
        else mprk::launch_eval_normals(s, n);
        c->normals_kernel = !(c->normals_asm && !cnt) ? "k_eval_normals_q" : n.gen_code ? "k_eval_normals_gen" : "k_eval_normals_asm";
    }
    return MPR_OK;
}

static int frame_finish(Frame& f)
{
    mpr_context* const c = f.c;
    HIP_TRY(hipGetLastError());
    c->frame_pending = true;
    c->pending_dim = f.dim;
    c->last_frame_lean = f.lean_now && !f.tiles_only;
    /* (looser enclosures decide less: tile lists and tapes of such a frame are sound, not the reference's) */
    c->last_frame_fast = (f.lean_now || f.skip0 || f.used_loose || (f.vox_fp && !f.vox_fp_beside)) && !f.tiles_only;      /* (vox_fp: no list of smallest tiles was made) */
    c->last_key = f.key;
    if (c->last_frame_fast && (!c->last_tape || c->last_tape->serial != f.tape->serial)) c->last_tape.reset(new mpr_tape(*f.tape));
    return f.blocking ? mpr_ctx_sync(c) : MPR_OK;
}

/* the comparison has been waited for: did the frame's shortcut past the 64^3 tiles fail its verification?  Then the tape's
 * next frames start at those tiles (the caller renders this one again) */
static bool skip0_verdict_failed(mpr_context* c, const mpr_tape* tape, bool whole_frame = true)
{
    if (!c->skip0_unchecked) return false;
    c->skip0_unchecked = false;
    const bool normals_veto = c->skip0_normals_veto;
    c->skip0_normals_veto = false;
    if (*reinterpret_cast<volatile int*>(c->skip0_flag_host) == 0 && !normals_veto) {
        /* a verified frame passed: the next failure starts at 64 again.  (whole_frame == false: only the tiles' verdict is in, the
         * normals pass has yet to accept the frame — its veto would find the entry gone and start at 64 every time: ADVICE r5) */
        auto it = c->skip0_veto.find(tape->serial);
        if (whole_frame && it != c->skip0_veto.end()) c->skip0_veto.erase(it);
        return false;
    }
    /* (a view that changes may pass later: the tape tries again after 64 frames, after 128 if that fails too, ... 4096) */
    mpr_context::Skip0Veto& v = c->skip0_veto[tape->serial];
    v.left = v.span;
    v.span = std::min(v.span * 2, 4096);
    if (c->skip0_veto.size() > 64) c->skip0_veto.erase(c->skip0_veto.begin());      /* (tapes long gone) */
    ++c->skip0_vetoes;
    return true;
}

static int render_frame(mpr_context* c, const mpr_tape* tape, int dim, const float* mat, float z,
                        const int32_t* owner, int rank, bool brute, bool blocking)
{
    for (;;) {
        Frame f;
        f.c = c; f.tape = tape; f.dim = dim; f.mat = mat; f.z = z; f.owner = owner; f.rank = rank; f.brute = brute; f.blocking = blocking;
        int rc = frame_begin(f);
        if (rc) return rc;
        for (int si = f.skip0 ? 1 : 0; si < f.nstages && rc == MPR_OK; ++si) rc = frame_tile_stage(f, si);
        if (rc == FRAME_AGAIN) { ++c->frames_restarted; continue; }
        if (rc == FRAME_STOP) return MPR_OK;
        if (rc) return rc;
        c->last.voxel_tiles = f.count;
        if (f.lean_first && !(f.vox_gen_planned && f.group_form && f.lean_now && f.decisions_recorded && f.group_stage == 2 && !f.last_recorded)) {
            /* a stage behind the first one, or a pass behind them, wants tapes after all (a last stage with few tiles runs
             * level-parallel, for one): the frame again with a first stage that pushes them, and so the tape's next frames */
            c->lean_first_veto = tape->serial;
            continue;
        }
        if (f.skip0_checked && !blocking) {
            HIP_TRY(hipEventSynchronize(c->ev_done));        /* (long there: two tile stages have been waited for since) */
            c->side_compare_pending = false;
            c->skip0_unchecked = true;
            if (skip0_verdict_failed(c, tape, false)) continue;
        }
        if (!f.tiles_only) {          /* (a reader's re-render: tiles and tapes are the reference's now; heights and normals were all along) */
            rc = frame_float_pass(f);
            if (rc == FRAME_AGAIN_REFERENCE) {
                c->force_reference = true;
                rc = render_frame(c, tape, dim, mat, z, owner, rank, brute, blocking);
                c->force_reference = false;
                return rc;
            }
            if (rc) return rc;
            if (dim == 3) {
                rc = frame_normals_pass(f);
                if (rc) return rc;
            }
        }
        if (f.skip0_checked && !blocking) {
            c->skip0_unchecked = true;              /* (the tiles' verdict stands; this adds the normals pass's and clears a passed tape's entry) */
            if (skip0_verdict_failed(c, tape)) continue;
        }
        if (f.skip0_checked && blocking) {
            /* every tile of the frame's first stage against its 64^3 parent, behind the frame: the side stream's walk of those
             * has long ended, nothing waits; the verdict is read once the frame has been waited for */
            rc = launch_skip0_side(f);
            if (rc) return rc;
            HIP_TRY(hipStreamWaitEvent(f.s, c->ev_check, 0));
            mprk::launch_skip0_compare(f.s, f.skip0_args, c->skip0_children, c->skip0_flag_dev);
            c->skip0_unchecked = true;
        }
        rc = frame_finish(f);
        if (rc) return rc;
        if (f.skip0_checked && blocking && skip0_verdict_failed(c, tape)) continue;      /* again, from the 64^3 tiles down */
        return MPR_OK;
    }
}


/* A reader of tiles, tapes or counters wants the state the reference leaves: if the last frame took a shortcut (no tapes from
 * its last tile stage, or no 64^3 stage) the frame is rendered again the reference's way — same tape (the context kept a copy),
 * same view, same partition.  The images come out the same bit for bit; in a partitioned context the columns other ranks
 * sent stay where mpr_unpack_* put them. */
static int ensure_full_frame(mpr_context* c)
{
    if (!c->last_frame_fast || c->raw_reads) return MPR_OK;
    if (!c->last_tape) return mpr::set_error(MPR_ERR_INVALID, "no tape to render the last frame's tapes from");
    const mpr_context::FrameKey k = c->last_key;
    c->force_reference = true;
    c->tiles_only = true;
    const int rc = render_frame(c, c->last_tape.get(), k.dim, k.mat, k.z, k.parted ? c->owner_host.data() : nullptr, k.rank, false, true);
    c->force_reference = false;
    c->tiles_only = false;
    return rc;
}

/* MPR_CTX_PARANOID: the frame both ways.  A frame that took none of the shortcuts (an instrumented one, MPR_LAST_STAGE_PUSH=1) IS the
 * reference's procedure and is only counted.  Otherwise its heights and normals are put aside, the frame is rendered again the
 * reference's way — every stage from the 64 px tiles down, exact enclosures, every tape pushed, float pass and normals pass on the
 * tapes the tiles carry — and the images are compared cell by cell on the device.  The context is left holding the second frame. */
static int render_checked(mpr_context* c, const mpr_tape* tape, int dim, const float* mat, float z, bool blocking)
{
    int rc = render_frame(c, tape, dim, mat, z, nullptr, 0, false, blocking);
    if (rc || !c || !(c->flags & MPR_CTX_PARANOID)) return rc;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    ++c->paranoid_frames;
    if (!c->last_frame_fast) return MPR_OK;
    const size_t n = (size_t)c->S * c->S;
    if (!c->paranoid_image) {
        int* pi = nullptr;
        uint32_t* pn = nullptr;
        unsigned long long* pc = nullptr;
        if (hipMalloc((void**)&pi, n * sizeof(int)) != hipSuccess || hipMalloc((void**)&pn, n * sizeof(uint32_t)) != hipSuccess ||
            hipMalloc((void**)&pc, 2 * sizeof(unsigned long long)) != hipSuccess) {
            (void)hipGetLastError();
            if (pi) (void)hipFree(pi);
            if (pn) (void)hipFree(pn);
            if (pc) (void)hipFree(pc);
            return mpr::set_error(MPR_ERR_ALLOC, "no memory for the second frame of a paranoid context");
        }
        c->paranoid_image = pi;             /* (all three or none: ADVICE r5) */
        c->paranoid_normals = pn;
        c->paranoid_count = pc;
    }
    HIP_TRY(hipMemcpyAsync(c->paranoid_image, c->filled[3], n * sizeof(int), hipMemcpyDeviceToDevice, c->stream));
    if (dim == 3) HIP_TRY(hipMemcpyAsync(c->paranoid_normals, c->normals, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->force_reference = true;
    rc = render_frame(c, tape, dim, mat, z, nullptr, 0, false, true);
    c->force_reference = false;
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(c->paranoid_count, 0, 2 * sizeof(unsigned long long), c->stream));
    mprk::launch_count_differences(c->stream, c->paranoid_image, c->filled[3], n, c->paranoid_count);
    if (dim == 3) mprk::launch_count_differences(c->stream, reinterpret_cast<const int*>(c->paranoid_normals), reinterpret_cast<const int*>(c->normals), n, c->paranoid_count + 1);
    unsigned long long h[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(h, c->paranoid_count, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    ++c->paranoid_compared;
    c->paranoid_cells += (long long)(h[0] + h[1]);
    return MPR_OK;
}

extern "C" {

/* {frames rendered, frames that took a shortcut and were rendered again the reference's way, cells (heights + normals) in which the
 * two renderings of a frame differed} since the context was made (MPR_CTX_PARANOID) */
int mpr_ctx_paranoid_stats(const mpr_context* c, int64_t out[3])
{
    if (!c || !out) return mpr::set_error(MPR_ERR_INVALID, "null argument");
    out[0] = c->paranoid_frames;
    out[1] = c->paranoid_compared;
    out[2] = c->paranoid_cells;
    return MPR_OK;
}
int32_t mpr_ctx_last_stage_pushed(const mpr_context* c) { return c ? (c->last_frame_lean ? 0 : 1) : 0; }
int64_t mpr_ctx_skip0_vetoes(const mpr_context* c) { return c ? c->skip0_vetoes : 0; }
#ifdef MPR_TEST_HOOKS
/* development (MPR_DEBUG_REDO=1): wavefronts that ran a scheduled forward walk since the context was made, and how many of them had
 * their loose walk redone on the exact code */
/* development: {capacity of the tape pool in clauses, times it grew, frames that started over, frames whose start at the 16^3 tiles
 * was vetoed} since the context was made (scripts/outlier_probe.py) */
extern "C" int mpr_debug_frame_stats(const mpr_context* c, int64_t out[4])
{
    if (!c || !out) return MPR_ERR_INVALID;
    out[0] = c->pool_cap;
    out[1] = c->pool_growths;
    out[2] = c->frames_restarted;
    out[3] = c->skip0_vetoes;
    return MPR_OK;
}
extern "C" int mpr_debug_redo_counts(mpr_context* c, uint32_t out[2])
{
    if (!c || !out) return MPR_ERR_INVALID;
    out[0] = out[1] = 0;
    if (!c->redo_count) return MPR_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    unsigned int h[2] = {0, 0};
    HIP_TRY(hipMemcpy(h, c->redo_count, sizeof h, hipMemcpyDeviceToHost));
    out[0] = h[0] + h[1];
    out[1] = h[1];
    return MPR_OK;
}
#endif  /* MPR_TEST_HOOKS */
const char* mpr_ctx_tile_stage_forms(const mpr_context* c) { return c ? c->stage_forms.c_str() : ""; }
/* tiles of the last frame AS IT RAN (no re-render: a frame nobody reads may start at the 16^3 tiles and cull with looser bounds than
 * the reference): per stage the tiles evaluated and the tiles left ambiguous, then the smallest tiles handed to the float pass */
int mpr_ctx_frame_tiles(mpr_context* c, int64_t out[7])
{
    if (!c || !out) return mpr::set_error(MPR_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int i = 0; i < 3; ++i) {
        out[i] = c->last.tiles_in[i];
        out[3 + i] = c->last.tiles_active[i];
    }
    out[6] = c->last.voxel_tiles;
    return MPR_OK;
}

#ifdef MPR_TEST_HOOKS
/* development (MPR_DEBUG_WALKED=1 when the context was made): tiles the last frame's float pass on the root tape's code walked; -1: not counted */
extern "C" long long mpr_debug_tiles_walked(mpr_context* c)
{
    if (!c || !c->walked_dev) return -1;
    if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return -1;
    int host[1024];
    if (hipMemcpy(host, c->walked_dev, sizeof host, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    long long n = 0;
    for (int i = 0; i < 1024; ++i) n += host[i];
    return n;
}
#endif  /* MPR_TEST_HOOKS */

int mpr_ctx_sync(mpr_context* c)
{
    if (!c) return mpr::set_error(MPR_ERR_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->frame_pending = false;
    return MPR_OK;
}

int mpr_render2d(mpr_context* c, const mpr_tape* t, const float m[9], float z)
{
    return render_checked(c, t, 2, m, z, true);
}
int mpr_render3d(mpr_context* c, const mpr_tape* t, const float m[16])
{
    return render_checked(c, t, 3, m, 0.0f, true);
}
/* Context::render2D_heatmap / render3D_heatmap (inc/context.hpp:51-58, src/context.cu:1984-2339): a
 * normal frame whose tile and pixel kernels also accumulate the words they walk, spread over the
 * pixels they cover; the sum is divided by the tape's clause count on the host as upstream does. */
static int render_heatmap(mpr_context* c, const mpr_tape* t, int dim, const float* m, float z, float* out)
{
    if (!c || !t || !out) return mpr::set_error(MPR_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    const size_t n = (size_t)c->S * c->S;
    if (!c->heat && hipMalloc(&c->heat, n * sizeof(float)) != hipSuccess) {
        c->heat = nullptr;
        return mpr::set_error(MPR_ERR_ALLOC, "heatmap allocation failed");
    }
    c->heat_frame = true;
    const int rc = render_frame(c, t, dim, m, z, nullptr, 0, false, true);
    c->heat_frame = false;
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out, c->heat, n * sizeof(float), hipMemcpyDeviceToHost));
    const float clauses = (float)((long long)t->clauses.size() - 2);       /* src/context.cu:2142, :2336 (integer divisor) */
    for (size_t i = 0; i < n; ++i) out[i] /= clauses;
    return MPR_OK;
}
int mpr_render2d_heatmap(mpr_context* c, const mpr_tape* t, const float m[9], float z, float* out)
{
    return render_heatmap(c, t, 2, m, z, out);
}
int mpr_render3d_heatmap(mpr_context* c, const mpr_tape* t, const float m[16], float* out)
{
    return render_heatmap(c, t, 3, m, 0.0f, out);
}
int mpr_render2d_brute(mpr_context* c, const mpr_tape* t, const float m[9], float z)
{
    return render_frame(c, t, 2, m, z, nullptr, 0, true, true);
}
int mpr_render2d_async(mpr_context* c, const mpr_tape* t, const float m[9], float z)
{
    return render_checked(c, t, 2, m, z, false);
}
int mpr_render3d_async(mpr_context* c, const mpr_tape* t, const float m[16])
{
    return render_checked(c, t, 3, m, 0.0f, false);
}
/* Per 64 x 64 column the number of first-stage tiles the interval evaluation leaves ambiguous: the work proxy the column deal of
 * SURVEY.md 8(e) uses (every rank runs the cheap 64 px stage over the whole frame, identically, and deals the columns by
 * longest-processing-time-first on these counts: no communication, no frame rendered in advance). */
int mpr_column_weights(mpr_context* c, const mpr_tape* t, int32_t dim, const float* mat, float z, float* weights)
{
    if (!c || !t || !mat || !weights || (dim != 2 && dim != 3)) return mpr::set_error(MPR_ERR_INVALID, "bad argument");
    c->force_reference = true;
    c->stage0_only = true;
    const int rc = render_frame(c, t, dim, mat, z, nullptr, 0, false, true);
    c->force_reference = false;
    c->stage0_only = false;
    /* the previous frame is gone (its images were cleared, its first tile list overwritten): readers get what is there — this
     * call's first stage — until the next frame is rendered */
    c->last_frame_fast = false;
    c->last_frame_lean = false;
    c->last_tape.reset();
    for (int i = 1; i < 4; ++i) c->tiles_n[i] = 0;
    if (rc) return rc;
    const int t0 = c->S / 64, cols = t0 * t0;
    const size_t n = (size_t)cols * (dim == 3 ? t0 : 1);
    std::vector<mpr_tile_node> tiles(n);
    HIP_TRY(hipMemcpy(tiles.data(), c->tiles[0], n * sizeof(mpr_tile_node), hipMemcpyDeviceToHost));
    for (int i = 0; i < cols; ++i) weights[i] = 0.0f;
    for (size_t k = 0; k < n; ++k)
        if (tiles[k].position != -1) weights[k % (size_t)cols] += 1.0f;
    return MPR_OK;
}
int mpr_render3d_part(mpr_context* c, const mpr_tape* t, const float m[16], const int32_t* owner, int32_t rank)
{
    if (!owner) return mpr::set_error(MPR_ERR_INVALID, "null owner table");
    return render_frame(c, t, 3, m, 0.0f, owner, rank, false, true);
}
int mpr_render3d_part_async(mpr_context* c, const mpr_tape* t, const float m[16], const int32_t* owner, int32_t rank)
{
    if (!owner) return mpr::set_error(MPR_ERR_INVALID, "null owner table");
    return render_frame(c, t, 3, m, 0.0f, owner, rank, false, false);
}
int mpr_render2d_part_async(mpr_context* c, const mpr_tape* t, const float m[9], float z, const int32_t* owner, int32_t rank)
{
    if (!owner) return mpr::set_error(MPR_ERR_INVALID, "null owner table");
    return render_frame(c, t, 2, m, z, owner, rank, false, false);
}
int mpr_render2d_part(mpr_context* c, const mpr_tape* t, const float m[9], float z, const int32_t* owner, int32_t rank)
{
    if (!owner) return mpr::set_error(MPR_ERR_INVALID, "null owner table");
    return render_frame(c, t, 2, m, z, owner, rank, false, true);
}

static int column_list(mpr_context* c, const int32_t* owner, int rank, int capacity, int* ncols)
{
    const int cols = (c->S / 64) * (c->S / 64);
    std::vector<int> list;
    for (int i = 0; i < cols; ++i) if (owner[i] == rank) list.push_back(i);
    if ((int)list.size() > capacity) return mpr::set_error(MPR_ERR_INVALID, "capacity_cols too small for this rank");
    *ncols = (int)list.size();
    c->my_cols_rank = -1;              /* col_list_dev is about to hold this list instead of the normals pass's */
    if (!list.empty()) {
        HIP_TRY(hipMemcpyAsync(c->col_list_dev, list.data(), list.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return MPR_OK;
}

int mpr_pack_columns(mpr_context* c, const int32_t* owner, int32_t rank, int32_t capacity_cols,
                     int32_t with_normals, void* dev_out)
{
    if (!c || !owner || !dev_out) return mpr::set_error(MPR_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    int ncols = 0;
    int rc = column_list(c, owner, rank, capacity_cols, &ncols);
    if (rc) return rc;
    mprk::launch_pack(c->stream, c->filled[3], c->normals, c->S, c->col_list_dev, ncols, capacity_cols, with_normals, (int*)dev_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MPR_OK;
}
int mpr_unpack_columns(mpr_context* c, const int32_t* owner, int32_t rank, int32_t capacity_cols,
                       int32_t with_normals, const void* dev_in)
{
    if (!c || !owner || !dev_in) return mpr::set_error(MPR_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    int ncols = 0;
    int rc = column_list(c, owner, rank, capacity_cols, &ncols);
    if (rc) return rc;
    mprk::launch_unpack(c->stream, c->filled[3], c->normals, c->S, c->col_list_dev, ncols, capacity_cols, with_normals, (const int*)dev_in);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MPR_OK;
}

/* ---- one-plan gather: no per-frame uploads or host synchronisation ---- */
int mpr_gather_plan(mpr_context* c, const int32_t* owner, int32_t rank, int32_t world, int32_t capacity_cols, int32_t with_normals)
{
    if (!c || !owner || world < 1 || rank < 0 || rank >= world) return mpr::set_error(MPR_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    const size_t cols = (size_t)(c->S / 64) * (c->S / 64);
    std::vector<int32_t> slot(cols, 0), fill((size_t)world, 0);
    for (size_t i = 0; i < cols; ++i) {
        if (owner[i] < 0 || owner[i] >= world) return mpr::set_error(MPR_ERR_INVALID, "owner entry out of range");
        slot[i] = fill[(size_t)owner[i]]++;
        if (slot[i] >= capacity_cols) return mpr::set_error(MPR_ERR_INVALID, "capacity_cols too small");
    }
    c->owner_host.assign(owner, owner + cols);
    c->owner_gen++;
    HIP_TRY(hipMemcpyAsync(c->owner_dev, c->owner_host.data(), cols * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->slot_dev, slot.data(), cols * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));        /* the host vectors go out of scope */
    c->plan_rank = rank;
    c->plan_world = world;
    c->plan_capacity = capacity_cols;
    c->plan_normals = with_normals;
    return MPR_OK;
}
int mpr_pack_planned_async(mpr_context* c, void* dev_out)
{
    if (!c || !dev_out || c->plan_rank < 0) return mpr::set_error(MPR_ERR_INVALID, "no gather plan");
    HIP_TRY(hipSetDevice(c->device));
    mprk::launch_pack_planned(c->stream, c->filled[3], c->normals, c->S, c->owner_dev, c->slot_dev, c->plan_rank, c->plan_capacity,
                              c->plan_normals, (int*)dev_out);
    HIP_TRY(hipGetLastError());
    return MPR_OK;
}
int mpr_unpack_planned_async(mpr_context* c, const void* dev_in_all)
{
    if (!c || !dev_in_all || c->plan_rank < 0) return mpr::set_error(MPR_ERR_INVALID, "no gather plan");
    HIP_TRY(hipSetDevice(c->device));
    mprk::launch_unpack_planned(c->stream, c->filled[3], c->normals, c->S, c->owner_dev, c->slot_dev, c->plan_rank, c->plan_capacity,
                                c->plan_normals, (const int*)dev_in_all);
    HIP_TRY(hipGetLastError());
    c->frame_pending = true;
    return MPR_OK;
}

/* ---- results ---- */
int mpr_read_filled(mpr_context* c, int32_t stage, int32_t* host)
{
    if (!c || !host || stage < 0 || stage > 3) return mpr::set_error(MPR_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    /* the images of the tile stages are the reference's only after a frame that ran every stage (a frame that started at the 16^3
     * tiles left stages[0].filled empty); the heightmap / 2-D image is the reference's after any frame */
    if (stage < 3)
        if (const int rc = ensure_full_frame(c)) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(host, c->filled[stage], c->filled_n[stage] * sizeof(int), hipMemcpyDeviceToHost));
    return MPR_OK;
}
int mpr_read_normals(mpr_context* c, uint32_t* host)
{
    if (!c || !host) return mpr::set_error(MPR_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(host, c->normals, (size_t)c->S * c->S * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return MPR_OK;
}
int mpr_read_tiles(mpr_context* c, int32_t stage, mpr_tile_node* host, size_t cap, size_t* n)
{
    if (!c || stage < 0 || stage > 3 || !n) return mpr::set_error(MPR_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    if (const int rc = ensure_full_frame(c)) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    *n = c->tiles_n[stage];
    if (host) {
        const size_t m = std::min(cap, c->tiles_n[stage]);
        if (m) HIP_TRY(hipMemcpy(host, c->tiles[stage], m * sizeof(mpr_tile_node), hipMemcpyDeviceToHost));
    }
    return MPR_OK;
}
int mpr_read_tape_pool(mpr_context* c, uint64_t* host, size_t cap, int32_t* tape_index)
{
    if (!c || !tape_index) return mpr::set_error(MPR_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    if (const int rc = ensure_full_frame(c)) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    int ti = 0;
    unsigned long long ti64 = 0;
    HIP_TRY(hipMemcpy(&ti64, c->tape_index, sizeof(ti64), hipMemcpyDeviceToHost));
    ti = (int)std::min<unsigned long long>(ti64, 0x7FFFFFFFull);
    *tape_index = ti;
    if (host) {
        const size_t m = std::min<size_t>(cap, (size_t)std::min<long long>(std::max(ti, 0), c->pool_cap));
        if (m) HIP_TRY(hipMemcpy(host, c->pool, m * sizeof(uint64_t), hipMemcpyDeviceToHost));
    }
    return MPR_OK;
}
int32_t* mpr_dev_filled(mpr_context* c, int32_t stage) { return (c && stage >= 0 && stage <= 3) ? c->filled[stage] : nullptr; }
uint32_t* mpr_dev_normals(mpr_context* c) { return c ? c->normals : nullptr; }
void* mpr_ctx_stream(mpr_context* c) { return c ? (void*)c->stream : nullptr; }

/* ---- mpr::Effects ------------------------------------------------------------------------ */
struct mpr_effects {
    int device = 0;
    int32_t S = 0;
    int32_t* tmp = nullptr;
    int32_t* image = nullptr;
    void* tables_dev = nullptr;
    float kernel[64 * 3];
    float rvecs[256 * 3];
};
int mpr_effects_create(int32_t device, mpr_effects** out)
{
    if (!out) return mpr::set_error(MPR_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(device));
    mpr_effects* fx = new mpr_effects();
    fx->device = device;
    mpr_effects_tables(fx->kernel, fx->rvecs);
    if (hipMalloc(&fx->tables_dev, mprk::effect_tables_bytes()) != hipSuccess) {
        delete fx;
        return mpr::set_error(MPR_ERR_ALLOC, "hipMalloc failed");
    }
    float both[64 * 3 + 256 * 3];
    std::memcpy(both, fx->kernel, sizeof(fx->kernel));
    std::memcpy(both + 64 * 3, fx->rvecs, sizeof(fx->rvecs));
    HIP_TRY(hipMemcpy(fx->tables_dev, both, sizeof(both), hipMemcpyHostToDevice));
    *out = fx;
    return MPR_OK;
}
void mpr_effects_destroy(mpr_effects* fx)
{
    if (!fx) return;
    (void)hipSetDevice(fx->device);
    if (fx->tmp) (void)hipFree(fx->tmp);
    if (fx->image) (void)hipFree(fx->image);
    if (fx->tables_dev) (void)hipFree(fx->tables_dev);
    delete fx;
}
/* Effects::resizeTo, src/effects.cu:238-244 */
static int effects_resize(mpr_effects* fx, const mpr_context* c)
{
    if (fx->S == c->S) return MPR_OK;
    if (fx->tmp) (void)hipFree(fx->tmp);
    if (fx->image) (void)hipFree(fx->image);
    fx->tmp = fx->image = nullptr;
    fx->S = 0;
    const size_t bytes = (size_t)c->S * c->S * sizeof(int32_t);
    HIP_TRY(hipMalloc((void**)&fx->tmp, bytes));
    HIP_TRY(hipMalloc((void**)&fx->image, bytes));
    fx->S = c->S;
    return MPR_OK;
}
static int effects_draw(mpr_effects* fx, mpr_context* c, bool shaded)
{
    if (!fx || !c) return mpr::set_error(MPR_ERR_INVALID, "null argument");
    if (fx->device != c->device) return mpr::set_error(MPR_ERR_INVALID, "effects and context live on different devices");
    HIP_TRY(hipSetDevice(c->device));
    int rc = effects_resize(fx, c);
    if (rc) return rc;
    hipStream_t s = c->stream;        /* ordered after the frame that produced depth and normals */
    const size_t bytes = (size_t)fx->S * fx->S * sizeof(int32_t);
    HIP_TRY(hipMemsetAsync(fx->tmp, 0, bytes, s));
    HIP_TRY(hipMemsetAsync(fx->image, 0, bytes, s));
    const int32_t* depth = c->filled[3];
    if (!shaded) {                    /* drawSSAO, src/effects.cu:246-263 */
        mprk::launch_draw_ssao(s, depth, c->normals, fx->tables_dev, fx->S, fx->tmp);
        mprk::launch_blur_ssao(s, depth, fx->tmp, fx->S, fx->image);
    } else {                          /* drawShaded, src/effects.cu:265-286 */
        mprk::launch_draw_ssao(s, depth, c->normals, fx->tables_dev, fx->S, fx->image);
        mprk::launch_blur_ssao(s, depth, fx->image, fx->S, fx->tmp);
        mprk::launch_draw_shaded(s, depth, c->normals, fx->tmp, fx->S, fx->image);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));
    return MPR_OK;
}
int mpr_effects_draw_ssao(mpr_effects* fx, mpr_context* c) { return effects_draw(fx, c, false); }
int mpr_effects_draw_shaded(mpr_effects* fx, mpr_context* c) { return effects_draw(fx, c, true); }
int mpr_effects_read_image(mpr_effects* fx, int32_t* host)
{
    if (!fx || !host || !fx->image) return mpr::set_error(MPR_ERR_INVALID, "nothing drawn yet");
    HIP_TRY(hipSetDevice(fx->device));
    HIP_TRY(hipMemcpy(host, fx->image, (size_t)fx->S * fx->S * sizeof(int32_t), hipMemcpyDeviceToHost));
    return MPR_OK;
}
int mpr_effects_read_tmp(mpr_effects* fx, int32_t* host)
{
    if (!fx || !host || !fx->tmp) return mpr::set_error(MPR_ERR_INVALID, "nothing drawn yet");
    HIP_TRY(hipSetDevice(fx->device));
    HIP_TRY(hipMemcpy(host, fx->tmp, (size_t)fx->S * fx->S * sizeof(int32_t), hipMemcpyDeviceToHost));
    return MPR_OK;
}
int32_t* mpr_effects_dev_image(mpr_effects* fx) { return fx ? fx->image : nullptr; }
int mpr_effects_tables_get(const mpr_effects* fx, float* kernel, float* rvecs)
{
    if (!fx) return mpr::set_error(MPR_ERR_INVALID, "null argument");
    if (kernel) std::memcpy(kernel, fx->kernel, sizeof(fx->kernel));
    if (rvecs) std::memcpy(rvecs, fx->rvecs, sizeof(fx->rvecs));
    return MPR_OK;
}

int mpr_get_counters(mpr_context* c, mpr_counters* out)
{
    if (!c || !out) return mpr::set_error(MPR_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    if (const int rc = ensure_full_frame(c)) return rc;        /* tape_index: the pool as the reference leaves it */
    HIP_TRY(hipStreamSynchronize(c->stream));
    int ti = 0;
    unsigned long long tiw[2] = {0, 0};
    HIP_TRY(hipMemcpy(tiw, c->tape_index, sizeof(tiw), hipMemcpyDeviceToHost));
    ti = (int)std::min<unsigned long long>(tiw[0], 0x7FFFFFFFull);
    c->last.tape_index = ti;
    c->last.pool_overflowed = tiw[1] ? 1 : 0;             /* set by any claim that did not fit, counters or not */
    if (c->flags & MPR_CTX_COUNTERS) {
        unsigned long long h[mprk::CNT_COUNT];
        HIP_TRY(hipMemcpy(h, c->counters, sizeof(h), hipMemcpyDeviceToHost));
        c->last.clauses_fwd = (int64_t)h[mprk::CNT_FWD];
        c->last.clauses_bwd = (int64_t)h[mprk::CNT_BWD];
        c->last.clauses_written = (int64_t)h[mprk::CNT_WRITTEN];
        c->last.lane_clauses = (int64_t)h[mprk::CNT_LANE];
        c->last.clauses_fwd_voxels = (int64_t)h[mprk::CNT_FWD_VOX];
        c->last.clauses_fwd_normals = (int64_t)h[mprk::CNT_FWD_NORM];
        c->last.normal_pixels = (int64_t)h[mprk::CNT_NORMAL_PX];
        if (h[mprk::CNT_OVERFLOW]) c->last.pool_overflowed = 1;
        if (c->debug_tiles & 4) {
            /* development: per-stage cycle breakdown of k_eval_tiles (sum over wavefronts) */
            unsigned long long ph[32];
            HIP_TRY(hipMemcpy(ph, c->counters + mprk::CNT_COUNT, sizeof(ph), hipMemcpyDeviceToHost));
            for (int st = 0; st < 3; ++st)
                fprintf(stderr, "stage %d: waves %llu  cycles/wave: prologue %.0f forward %.0f classify+claim %.0f backward %.0f head %.0f\n", st,
                        ph[st * 6 + 5], ph[st * 6 + 0] / (double)std::max(1ull, ph[st * 6 + 5]), ph[st * 6 + 1] / (double)std::max(1ull, ph[st * 6 + 5]),
                        ph[st * 6 + 2] / (double)std::max(1ull, ph[st * 6 + 5]), ph[st * 6 + 3] / (double)std::max(1ull, ph[st * 6 + 5]),
                        ph[st * 6 + 4] / (double)std::max(1ull, ph[st * 6 + 5]));
        }
    }
    if ((long long)ti >= c->pool_cap) c->last.pool_overflowed = 1;
    *out = c->last;
    return MPR_OK;
}

const char* mpr_ctx_float_kernel(const mpr_context* c) { return c ? c->float_kernel : ""; }
const char* mpr_ctx_normals_kernel(const mpr_context* c) { return c ? c->normals_kernel : ""; }

int mpr_get_timings(mpr_context* c, const char** names, float* ms, int32_t cap, int32_t* n)
{
    if (!c || !n) return mpr::set_error(MPR_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    int32_t k = 0;
    for (size_t i = 0; i < c->timings_used && k < cap; ++i, ++k) {
        float t = 0.0f;
        HIP_TRY(hipEventElapsedTime(&t, c->timings[i].start, c->timings[i].stop));
        if (names) names[k] = c->timings[i].name;
        if (ms) ms[k] = t;
    }
    *n = k;
    return MPR_OK;
}

#ifdef MPR_TEST_HOOKS
/* ---- primitive self-tests ---- */
namespace {
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 4); }
};
}  // namespace

int mpr_test_interval_op(int32_t device, int32_t op, int32_t n, const float* a_lo, const float* a_hi,
                         const float* b_lo, const float* b_hi, float imm, float* out_lo, float* out_hi,
                         int32_t* out_choice)
{
    if (n <= 0 || !a_lo || !a_hi || !out_lo || !out_hi) return mpr::set_error(MPR_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(device));
    const size_t bytes = (size_t)n * 4;
    DevBuf al, ah, bl, bh, ol, oh, ch;
    HIP_TRY(al.alloc(bytes)); HIP_TRY(ah.alloc(bytes)); HIP_TRY(bl.alloc(bytes)); HIP_TRY(bh.alloc(bytes));
    HIP_TRY(ol.alloc(bytes)); HIP_TRY(oh.alloc(bytes)); HIP_TRY(ch.alloc(bytes));
    HIP_TRY(hipMemcpy(al.p, a_lo, bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ah.p, a_hi, bytes, hipMemcpyHostToDevice));
    if (b_lo) HIP_TRY(hipMemcpy(bl.p, b_lo, bytes, hipMemcpyHostToDevice));
    if (b_hi) HIP_TRY(hipMemcpy(bh.p, b_hi, bytes, hipMemcpyHostToDevice));
    mprk::launch_test_interval(nullptr, op, n, (float*)al.p, (float*)ah.p, b_lo ? (float*)bl.p : nullptr,
                               b_hi ? (float*)bh.p : nullptr, imm, (float*)ol.p, (float*)oh.p, (int*)ch.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out_lo, ol.p, bytes, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_hi, oh.p, bytes, hipMemcpyDeviceToHost));
    if (out_choice) HIP_TRY(hipMemcpy(out_choice, ch.p, bytes, hipMemcpyDeviceToHost));
    return MPR_OK;
}
int mpr_test_interval_op_asm(int32_t device, int32_t op, int32_t variant, int32_t n, const float* a_lo, const float* a_hi,
                             const float* b_lo, const float* b_hi, float imm, float* out_lo, float* out_hi,
                             int32_t* out_choice)
{
    if (n <= 0 || !a_lo || !a_hi || !out_lo || !out_hi) return mpr::set_error(MPR_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(device));
    const size_t bytes = (size_t)n * 4;
    DevBuf al, ah, bl, bh, ol, oh, ch, dt;
    HIP_TRY(al.alloc(bytes)); HIP_TRY(ah.alloc(bytes)); HIP_TRY(bl.alloc(bytes)); HIP_TRY(bh.alloc(bytes));
    HIP_TRY(ol.alloc(bytes)); HIP_TRY(oh.alloc(bytes)); HIP_TRY(ch.alloc(bytes)); HIP_TRY(dt.alloc(64 * 8));
    HIP_TRY(hipMemcpy(al.p, a_lo, bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ah.p, a_hi, bytes, hipMemcpyHostToDevice));
    if (b_lo) HIP_TRY(hipMemcpy(bl.p, b_lo, bytes, hipMemcpyHostToDevice));
    if (b_hi) HIP_TRY(hipMemcpy(bh.p, b_hi, bytes, hipMemcpyHostToDevice));
    uint32_t immbits;
    memcpy(&immbits, &imm, 4);
    /* variant 1 / 2: a copy in front makes lhs / rhs "the previous clause's result" (operand forwarding) */
    const bool has_b = b_lo && b_hi;
    const uint32_t lhs = variant == 1 ? 5 : 1, rhs = has_b ? (variant == 2 ? 5 : 2) : 0;
    uint64_t tape[64] = {mpr_cl_make(0, 1, 2, 3, 0),
                         variant == 2 ? mpr_cl_make(MPR_OP_COPY_RHS, 5, 0, 2, 0) : mpr_cl_make(MPR_OP_COPY_LHS, 5, 1, 0, 0),
                         mpr_cl_make((uint32_t)op, 4, lhs, rhs, immbits), mpr_cl_make(0, 4, 0, 0, 0)};
    HIP_TRY(hipMemcpy(dt.p, tape, sizeof(tape), hipMemcpyHostToDevice));
    mprk::launch_test_interval_asm(nullptr, (const uint64_t*)dt.p, n, (float*)al.p, (float*)ah.p, b_lo ? (float*)bl.p : nullptr,
                                   b_hi ? (float*)bh.p : nullptr, (float*)ol.p, (float*)oh.p, (int*)ch.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out_lo, ol.p, bytes, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_hi, oh.p, bytes, hipMemcpyDeviceToHost));
    if (out_choice) HIP_TRY(hipMemcpy(out_choice, ch.p, bytes, hipMemcpyDeviceToHost));
    return MPR_OK;
}
/* ---- one clause through the scheduled interval code on the device (include/mpr_amd.h: mpr_test_interval_gen_op / mpr_test_loose_gen) ---- */
namespace {
struct OneClauseCode {
    uint32_t* code = nullptr;
    ~OneClauseCode() { free_executable(code); }
    /* tape {head (1, 2, 3), clause (out 4, lhs 1, rhs 2 where the opcode has one), end (4)} -> installed code */
    int make(int device, int op, float imm, bool loose, bool tight = false)
    {
        uint32_t immbits;
        memcpy(&immbits, &imm, 4);
        const bool has_l = !(op == MPR_OP_SUB_IMM_RHS || op == MPR_OP_DIV_IMM_RHS || op == MPR_OP_COPY_IMM || op == MPR_OP_COPY_RHS);
        const bool has_r = op == MPR_OP_ADD_LHS_RHS || op == MPR_OP_MUL_LHS_RHS || op == MPR_OP_MIN_LHS_RHS || op == MPR_OP_MAX_LHS_RHS ||
                           op == MPR_OP_SUB_IMM_RHS || op == MPR_OP_SUB_LHS_RHS || op == MPR_OP_DIV_IMM_RHS || op == MPR_OP_DIV_LHS_RHS || op == MPR_OP_COPY_RHS;
        const uint64_t tape[3] = {mpr_cl_make(0, 1, 2, 3, 0), mpr_cl_make((uint32_t)op, 4, has_l ? 1 : 0, has_r ? 2 : 0, immbits), mpr_cl_make(0, 4, 0, 0, 0)};
        const mpr::IntervalCode g = mpr::interval_gen_build(tape, 3, mpr::IW_FIRST, loose, 0, 3, false, tight ? mpr::IGEN_TIGHT_VGPRS : loose ? mpr::IGEN_LEAN_VGPRS : 0, loose, tight);
        if (!g.ok) return mpr::set_error(MPR_ERR_UNSUPPORTED, "the generator does not take this clause");
        const size_t n = (g.words.size() + 127) & ~(size_t)63;
        code = static_cast<uint32_t*>(alloc_executable(device, n * sizeof(uint32_t)));
        if (!code) return mpr::set_error(MPR_ERR_NO_DEVICE, "no executable memory");
        DevBuf stage;
        HIP_TRY(stage.alloc(n * sizeof(uint32_t)));
        std::vector<uint32_t> all(n, 0xBF800000u);
        std::copy(g.words.begin(), g.words.end(), all.begin());
        HIP_TRY(hipMemcpy(stage.p, all.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        mprk::launch_install_code(nullptr, code, (const uint32_t*)stage.p, n, std::max(prop.multiProcessorCount, 1));
        HIP_TRY(hipDeviceSynchronize());
        return MPR_OK;
    }
};
}  // namespace
extern "C" int mpr_test_interval_gen_op(int32_t device, int32_t op, int32_t loose, int32_t n, const float* a_lo, const float* a_hi, const float* b_lo,
                                        const float* b_hi, float imm, float* out_lo, float* out_hi, int32_t* out_choice, int32_t* out_asks_exact)
{
    if (n <= 0 || !a_lo || !a_hi || !out_lo || !out_hi || !out_choice || !out_asks_exact) return mpr::set_error(MPR_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(device));
    OneClauseCode oc;
    const int rc = oc.make(device, op, imm, loose != 0);
    if (rc) return rc;
    const size_t bytes = (size_t)n * 4;
    DevBuf al, ah, bl, bh, ol, oh, ch, ax;
    HIP_TRY(al.alloc(bytes)); HIP_TRY(ah.alloc(bytes)); HIP_TRY(bl.alloc(bytes)); HIP_TRY(bh.alloc(bytes));
    HIP_TRY(ol.alloc(bytes)); HIP_TRY(oh.alloc(bytes)); HIP_TRY(ch.alloc(bytes)); HIP_TRY(ax.alloc(bytes));
    HIP_TRY(hipMemcpy(al.p, a_lo, bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ah.p, a_hi, bytes, hipMemcpyHostToDevice));
    if (b_lo) HIP_TRY(hipMemcpy(bl.p, b_lo, bytes, hipMemcpyHostToDevice));
    if (b_hi) HIP_TRY(hipMemcpy(bh.p, b_hi, bytes, hipMemcpyHostToDevice));
    mprk::launch_test_interval_gen(nullptr, oc.code, loose != 0, n, (float*)al.p, (float*)ah.p, b_lo ? (float*)bl.p : nullptr, b_hi ? (float*)bh.p : nullptr,
                                   (float*)ol.p, (float*)oh.p, (int*)ch.p, (int*)ax.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out_lo, ol.p, bytes, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_hi, oh.p, bytes, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_choice, ch.p, bytes, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out_asks_exact, ax.p, bytes, hipMemcpyDeviceToHost));
    return MPR_OK;
}
extern "C" int mpr_test_loose_gen(int32_t device, int32_t op, float imm, float other_lo, float other_hi, int32_t x_is_rhs, uint64_t first, uint64_t count,
                                  uint64_t out[5])
{
    if (!out) return mpr::set_error(MPR_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(device));
    OneClauseCode oc;
    const int rc = oc.make(device, op, imm, true);
    if (rc) return rc;
    DevBuf d;
    HIP_TRY(d.alloc(5 * 8));
    HIP_TRY(hipMemset(d.p, 0, 5 * 8));
    mprk::launch_test_loose_gen(nullptr, oc.code, op, imm, other_lo, other_hi, x_is_rhs, first, count, (unsigned long long*)d.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, d.p, 5 * 8, hipMemcpyDeviceToHost));
    return MPR_OK;
}
extern "C" int mpr_test_tight_trig(int32_t device, int32_t is_sin, uint64_t first, uint64_t count, uint64_t out[7])
{
    if (!out) return mpr::set_error(MPR_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(device));
    OneClauseCode oc;
    const int rc = oc.make(device, is_sin ? MPR_OP_SIN_LHS : MPR_OP_COS_LHS, 0.0f, true, true);
    if (rc) return rc;
    DevBuf d;
    HIP_TRY(d.alloc(7 * 8));
    HIP_TRY(hipMemset(d.p, 0, 7 * 8));
    mprk::launch_test_tight_trig(nullptr, oc.code, is_sin, first, count, (unsigned long long*)d.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, d.p, 7 * 8, hipMemcpyDeviceToHost));
    return MPR_OK;
}
extern "C" int mpr_test_float_in_enclosure(int32_t device, int32_t op, float imm, uint64_t first, uint64_t count, uint64_t out[6])
{
    if (!out) return mpr::set_error(MPR_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(device));
    DevBuf d;
    HIP_TRY(d.alloc(6 * 8));
    HIP_TRY(hipMemset(d.p, 0, 6 * 8));
    mprk::launch_test_float_in_enclosure(nullptr, op, imm, first, count, (unsigned long long*)d.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, d.p, 6 * 8, hipMemcpyDeviceToHost));
    return MPR_OK;
}
/* development aid (scripts/walk_cycles.py): mean cycles per scheduled forward walk (interval_gen.hpp) of `clauses`, per wavefront, with
 * `waves` wavefronts in flight (one per workgroup, four per SIMD at most); empty != 0: the harness alone (code that returns at once) */
extern "C" int mpr_debug_walk_cycles(int32_t device, const uint64_t* clauses, int32_t length, int32_t kind, int32_t loose, int32_t window,
                                     int32_t empty, int32_t reps, int32_t waves, long long* out, uint32_t* redone, int32_t* info)
{
    HIP_TRY(hipSetDevice(device));
    std::vector<uint32_t> words, exact;
    if (empty) {
        words.push_back(0xBE801D26u);                        /* s_setpc_b64 s[38:39] */
    } else {
        const mpr::IntervalCode g = mpr::interval_gen_build(clauses, length, kind, loose != 0, window);
        if (!g.ok) return mpr::set_error(MPR_ERR_UNSUPPORTED, "the generator does not take this tape");
        words = g.words;
        if (info) { info[0] = g.instructions; info[1] = g.window; info[2] = g.max_vgprs; info[3] = g.est_cycles; }
        if (loose) {
            const mpr::IntervalCode x = mpr::interval_gen_build(clauses, length, kind, false, window);
            if (!x.ok) return mpr::set_error(MPR_ERR_UNSUPPORTED, "the generator does not take this tape");
            exact = x.words;
        }
    }
    const size_t n1 = (words.size() + 63) & ~(size_t)63, n2 = (exact.size() + 63) & ~(size_t)63;
    uint32_t* code = static_cast<uint32_t*>(alloc_executable(device, (n1 + n2 + 64) * sizeof(uint32_t)));
    if (!code) return mpr::set_error(MPR_ERR_NO_DEVICE, "no executable memory");
    DevBuf stage, dout, dred;
    HIP_TRY(stage.alloc((n1 + n2 + 64) * sizeof(uint32_t)));
    HIP_TRY(dout.alloc((size_t)waves * 8));
    HIP_TRY(dred.alloc(8));
    HIP_TRY(hipMemset(dred.p, 0, 8));
    std::vector<uint32_t> all(n1 + n2 + 64, 0xBF800000u);
    std::copy(words.begin(), words.end(), all.begin());
    std::copy(exact.begin(), exact.end(), all.begin() + (long)n1);
    HIP_TRY(hipMemcpy(stage.p, all.data(), all.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    {
        /* (one workgroup per compute unit: each invalidates its own instruction cache) */
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        mprk::launch_install_code(nullptr, code, (const uint32_t*)stage.p, all.size(), std::max(prop.multiProcessorCount, 1));
    }
    HIP_TRY(hipDeviceSynchronize());
    mprk::launch_debug_walk_cycles(nullptr, code, exact.empty() ? nullptr : code + n1, reps, (long long*)dout.p, (unsigned int*)dred.p, waves);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dout.p, (size_t)waves * 8, hipMemcpyDeviceToHost));
    if (redone) HIP_TRY(hipMemcpy(redone, dred.p, 4, hipMemcpyDeviceToHost));
    free_executable(code);
    return MPR_OK;
}
/* development aid (scripts/interp_cycles.py): cycles per forward walk of `clauses` (head, body, end) */
extern "C" int mpr_debug_interp_cycles(int32_t device, const uint64_t* clauses, int32_t length, int32_t reps, int32_t waves, long long* out)
{
    HIP_TRY(hipSetDevice(device));
    DevBuf dt, dout;
    HIP_TRY(dt.alloc((size_t)(length + 128) * 8));
    HIP_TRY(dout.alloc((size_t)reps * 8));
    HIP_TRY(hipMemset(dt.p, 0, (size_t)(length + 128) * 8));
    HIP_TRY(hipMemcpy(dt.p, clauses, (size_t)length * 8, hipMemcpyHostToDevice));
    mprk::launch_debug_interp_cycles(nullptr, (const uint64_t*)dt.p, reps, (long long*)dout.p, waves);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dout.p, (size_t)reps * 8, hipMemcpyDeviceToHost));
    return MPR_OK;
}
int mpr_test_float_op(int32_t device, int32_t op, int32_t n, const float* a, const float* b, float imm, float* out)
{
    if (n <= 0 || !a || !out) return mpr::set_error(MPR_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(device));
    const size_t bytes = (size_t)n * 4;
    DevBuf da, db, dout;
    HIP_TRY(da.alloc(bytes)); HIP_TRY(db.alloc(bytes)); HIP_TRY(dout.alloc(bytes));
    HIP_TRY(hipMemcpy(da.p, a, bytes, hipMemcpyHostToDevice));
    if (b) HIP_TRY(hipMemcpy(db.p, b, bytes, hipMemcpyHostToDevice));
    mprk::launch_test_float(nullptr, op, n, (float*)da.p, b ? (float*)db.p : nullptr, imm, (float*)dout.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dout.p, bytes, hipMemcpyDeviceToHost));
    return MPR_OK;
}
int mpr_test_float_op_asm(int32_t device, int32_t op, int32_t variant, int32_t n, const float* a, const float* b, float imm, float* out)
{
    if (n <= 0 || !a || !out) return mpr::set_error(MPR_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(device));
    const size_t bytes = (size_t)n * 4;
    DevBuf da, db, dout, dt;
    HIP_TRY(da.alloc(bytes)); HIP_TRY(db.alloc(bytes)); HIP_TRY(dout.alloc(bytes)); HIP_TRY(dt.alloc(64 * 8));
    HIP_TRY(hipMemcpy(da.p, a, bytes, hipMemcpyHostToDevice));
    if (b) HIP_TRY(hipMemcpy(db.p, b, bytes, hipMemcpyHostToDevice));
    uint32_t immbits;
    memcpy(&immbits, &imm, 4);
    /* 64 clauses so that the interpreter's 63-clause block fetch stays inside the buffer */
    /* variant 1 / 2: a copy in front makes lhs / rhs "the previous clause's result" (operand forwarding:
     * handler tables 1 / 2); 3..5: the same three with two negations of the result behind the clause, so that
     * its result "dies in the next clause" (tables 3..5: no store, no address); -(-x) is x bit for bit */
    const bool jit = variant >= 6;             /* 6..8: variants 0..2 as generated machine code */
    if (jit) variant -= 6;
    const bool dies = variant >= 3;
    if (dies) variant -= 3;
    const uint32_t lhs = variant == 1 ? 5 : 1, rhs = b ? (variant == 2 ? 5 : 2) : 0;
    uint64_t tape3[64] = {mpr_cl_make(0, 1, 2, 3, 0),
                          variant == 2 ? mpr_cl_make(MPR_OP_COPY_RHS, 5, 0, 2, 0) : mpr_cl_make(MPR_OP_COPY_LHS, 5, 1, 0, 0),
                          mpr_cl_make((uint32_t)op, 4, lhs, rhs, immbits), mpr_cl_make(0, 4, 0, 0, 0)};
    if (dies) {
        tape3[3] = mpr_cl_make(MPR_OP_NEG_LHS, 4, 4, 0, 0);
        tape3[4] = mpr_cl_make(MPR_OP_NEG_LHS, 4, 4, 0, 0);
        tape3[5] = mpr_cl_make(0, 4, 0, 0, 0);
    }
    HIP_TRY(hipMemcpy(dt.p, tape3, sizeof(tape3), hipMemcpyHostToDevice));
    if (jit) {
        /* the same tape through the translator and the generated code (kernels_voxel_jit.hip) */
        const uint32_t region = 512;
        const size_t blocks = ((size_t)n + 63) / 64;
        uint32_t* code = static_cast<uint32_t*>(alloc_executable(device, blocks * region * sizeof(uint32_t)));
        if (!code) return mpr::set_error(MPR_ERR_UNSUPPORTED, "no executable device memory");
        mprk::launch_test_float_jit(nullptr, (const uint64_t*)dt.p, code, region, n, (float*)da.p, b ? (float*)db.p : nullptr, (float*)dout.p);
        const hipError_t e1 = hipGetLastError(), e2 = hipDeviceSynchronize();
        free_executable(code);
        HIP_TRY(e1);
        HIP_TRY(e2);
        HIP_TRY(hipMemcpy(out, dout.p, bytes, hipMemcpyDeviceToHost));
        return MPR_OK;
    }
    mprk::launch_test_float_asm(nullptr, (const uint64_t*)dt.p, n, (float*)da.p, b ? (float*)db.p : nullptr, (float*)dout.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dout.p, bytes, hipMemcpyDeviceToHost));
    return MPR_OK;
}
/* One clause through the host-generated float walk (voxel_gen.hpp; mpr_test_voxel_gen shows the code): a, b arrive in slots 1, 2.
 * variant 0: out = slot 4; 1: a copy of a in slot 5 is the lhs AND the out slot (result written over its operand); 2: the same for
 * the rhs.  dl / dr: the tile's decisions, bit 0 = this clause (a decided min / max is a copy of that operand). */
int mpr_test_float_op_gen(int32_t device, int32_t op, int32_t variant, uint64_t dl, uint64_t dr, int32_t n, const float* a, const float* b,
                          float imm, float* out)
{
    if (n <= 0 || !a || !out || variant < 0 || variant > 2) return mpr::set_error(MPR_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(device));
    uint32_t immbits;
    memcpy(&immbits, &imm, 4);
    const uint32_t lhs = variant == 1 ? 5 : 1, rhs = b ? (variant == 2 ? 5 : 2) : 0, outs = variant == 0 ? 4 : 5;
    const uint64_t tape[4] = {mpr_cl_make(0, 1, 2, 3, 0),
                              variant == 2 ? mpr_cl_make(MPR_OP_COPY_RHS, 5, 0, 2, 0) : mpr_cl_make(MPR_OP_COPY_LHS, 5, 1, 0, 0),
                              mpr_cl_make((uint32_t)op, outs, lhs, rhs, immbits), mpr_cl_make(0, outs, 0, 0, 0)};
    const mpr::VoxelGen g = mpr::voxel_gen_build(tape, 4, 1);
    if (!g.ok) return mpr::set_error(MPR_ERR_UNSUPPORTED, "no generated code for this clause");
    const size_t bytes = (size_t)n * 4, cbytes = (g.code.size() + 64) * sizeof(uint32_t);
    DevBuf da, db, dout, dc;
    HIP_TRY(da.alloc(bytes)); HIP_TRY(db.alloc(bytes)); HIP_TRY(dout.alloc(bytes)); HIP_TRY(dc.alloc(cbytes));
    HIP_TRY(hipMemcpy(da.p, a, bytes, hipMemcpyHostToDevice));
    if (b) HIP_TRY(hipMemcpy(db.p, b, bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dc.p, g.code.data(), g.code.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    uint32_t* code = static_cast<uint32_t*>(alloc_executable(device, cbytes));
    if (!code) return mpr::set_error(MPR_ERR_UNSUPPORTED, "no executable device memory");
    int cus = 1;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    mprk::launch_install_code(nullptr, code, (const uint32_t*)dc.p, g.code.size(), std::max(cus, 1));
    mprk::launch_test_float_gen(nullptr, code, n, (float*)da.p, b ? (float*)db.p : nullptr, (float*)dout.p, dl, dr);
    const hipError_t e1 = hipGetLastError(), e2 = hipDeviceSynchronize();
    free_executable(code);
    HIP_TRY(e1);
    HIP_TRY(e2);
    HIP_TRY(hipMemcpy(out, dout.p, bytes, hipMemcpyDeviceToHost));
    return MPR_OK;
}
/* a clause with one operand (x in the lhs, and in the rhs where the opcode has one) through the host-generated float walk on every bit
 * pattern of [first, first + count) against the float pass's definition: out = {tested, differing, an input that differs} */
int mpr_test_float_gen_all(int32_t device, int32_t op, float imm, uint64_t first, uint64_t count, uint64_t out[3])
{
    if (!out) return mpr::set_error(MPR_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(device));
    uint32_t immbits;
    memcpy(&immbits, &imm, 4);
    const bool has_r = op == MPR_OP_ADD_LHS_RHS || op == MPR_OP_MUL_LHS_RHS || op == MPR_OP_MIN_LHS_RHS || op == MPR_OP_MAX_LHS_RHS || op == MPR_OP_SUB_IMM_RHS ||
                       op == MPR_OP_SUB_LHS_RHS || op == MPR_OP_DIV_IMM_RHS || op == MPR_OP_DIV_LHS_RHS;
    const bool has_l = !(op == MPR_OP_SUB_IMM_RHS || op == MPR_OP_DIV_IMM_RHS);
    const uint64_t tape[3] = {mpr_cl_make(0, 1, 2, 3, 0), mpr_cl_make((uint32_t)op, 4, has_l ? 1 : 0, has_r ? 2 : 0, immbits), mpr_cl_make(0, 4, 0, 0, 0)};
    const mpr::VoxelGen g = mpr::voxel_gen_build(tape, 3, 1);
    if (!g.ok) return mpr::set_error(MPR_ERR_UNSUPPORTED, "no generated code for this clause");
    const size_t cbytes = (g.code.size() + 64) * sizeof(uint32_t);
    DevBuf dc, d;
    HIP_TRY(dc.alloc(cbytes));
    HIP_TRY(d.alloc(3 * 8));
    HIP_TRY(hipMemset(d.p, 0, 3 * 8));
    HIP_TRY(hipMemcpy(dc.p, g.code.data(), g.code.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    uint32_t* code = static_cast<uint32_t*>(alloc_executable(device, cbytes));
    if (!code) return mpr::set_error(MPR_ERR_UNSUPPORTED, "no executable device memory");
    int cus = 1;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    mprk::launch_install_code(nullptr, code, (const uint32_t*)dc.p, g.code.size(), std::max(cus, 1));
    mprk::launch_test_float_gen_all(nullptr, code, op, imm, first, count, (unsigned long long*)d.p);
    const hipError_t e1 = hipGetLastError(), e2 = hipDeviceSynchronize();
    free_executable(code);
    HIP_TRY(e1);
    HIP_TRY(e2);
    HIP_TRY(hipMemcpy(out, d.p, 3 * 8, hipMemcpyDeviceToHost));
    return MPR_OK;
}
/* the square-root routine of the float interpreters / generated code on the bit patterns [first, first + count): number of results
 * that differ from the correctly rounded root, and one such input */
int mpr_test_sqrt_all(int32_t device, uint64_t first, uint64_t count, uint64_t* mismatches, uint32_t* example)
{
    if (!mismatches) return mpr::set_error(MPR_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(device));
    DevBuf d;
    HIP_TRY(d.alloc(16));
    HIP_TRY(hipMemset(d.p, 0, 16));
    mprk::launch_test_sqrt_all(nullptr, first, count, (unsigned long long*)d.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long h[2] = {0, 0};
    HIP_TRY(hipMemcpy(h, d.p, sizeof(h), hipMemcpyDeviceToHost));
    *mismatches = h[0];
    if (example) *example = (uint32_t)h[1];
    return MPR_OK;
}
int mpr_test_deriv_op(int32_t device, int32_t op, int32_t n, const float* a4, const float* b4, float imm, float* out4)
{
    if (n <= 0 || !a4 || !out4) return mpr::set_error(MPR_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(device));
    const size_t bytes = (size_t)n * 16;
    DevBuf da, db, dout;
    HIP_TRY(da.alloc(bytes)); HIP_TRY(db.alloc(bytes)); HIP_TRY(dout.alloc(bytes));
    HIP_TRY(hipMemcpy(da.p, a4, bytes, hipMemcpyHostToDevice));
    if (b4) HIP_TRY(hipMemcpy(db.p, b4, bytes, hipMemcpyHostToDevice));
    mprk::launch_test_deriv(nullptr, op, n, (float*)da.p, b4 ? (float*)db.p : nullptr, imm, (float*)dout.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out4, dout.p, bytes, hipMemcpyDeviceToHost));
    return MPR_OK;
}

#endif  /* MPR_TEST_HOOKS */
}  // extern "C"
