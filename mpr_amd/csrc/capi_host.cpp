/*
 * capi_host.cpp — host-only part of the C ABI (include/mpr_amd.h): expression trees,
 * .frep archives, tape building, column partitioning.  No GPU needed.
 */
#include <algorithm>
#include <cstring>
#include <future>
#include <new>
#include <numeric>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mpr_amd.h"
#include "../../include/mpr_amd_test.h"
#include "internal.hpp"
#include "interval_gen.hpp"
#include "tile_gen.hpp"
#include "frame_domain.hpp"
#include "tape_builder.hpp"
#include "tree.hpp"

using mpr::front::Tree;

struct mpr_tree {
    Tree t;
};

namespace mpr {
thread_local std::string g_last_error;
int set_error(int code, const std::string& msg)
{
    g_last_error = msg;
    return code;
}
}  // namespace mpr

extern "C" {

const char* mpr_last_error(void) { return mpr::g_last_error.c_str(); }
const char* mpr_version(void) { return "mpr_amd 0.1 (gfx950)"; }

const char* mpr_op_str(uint8_t op)
{
    /* names as in src/gpu_opcode.cu:17-58 */
    static const char* names[MPR_OP_COUNT] = {
        "INVALID", "JUMP", "SQUARE_LHS", "SQRT_LHS", "NEG_LHS", "SIN_LHS", "COS_LHS", "ASIN_LHS",
        "ACOS_LHS", "ATAN_LHS", "EXP_LHS", "ABS_LHS", "LOG_LHS", "ADD_LHS_IMM", "ADD_LHS_RHS",
        "MUL_LHS_IMM", "MUL_LHS_RHS", "MIN_LHS_IMM", "MIN_LHS_RHS", "MAX_LHS_IMM", "MAX_LHS_RHS",
        "SUB_LHS_IMM", "SUB_IMM_RHS", "SUB_LHS_RHS", "DIV_LHS_IMM", "DIV_IMM_RHS", "DIV_LHS_RHS",
        "COPY_IMM", "COPY_LHS", "COPY_RHS"};
    return op < MPR_OP_COUNT ? names[op] : "INVALID";
}

#define MPR_TRY(body)                                                        \
    try { body }                                                             \
    catch (const std::bad_alloc&) { return mpr::set_error(MPR_ERR_ALLOC, "out of memory"); } \
    catch (const std::exception& e) { return mpr::set_error(MPR_ERR_INVALID, e.what()); }

static int wrap(Tree t, mpr_tree** out)
{
    if (!out) return mpr::set_error(MPR_ERR_INVALID, "null output pointer");
    *out = new mpr_tree{std::move(t)};
    return MPR_OK;
}

int mpr_tree_x(mpr_tree** out) { MPR_TRY(return wrap(Tree::X(), out);) }
int mpr_tree_y(mpr_tree** out) { MPR_TRY(return wrap(Tree::Y(), out);) }
int mpr_tree_z(mpr_tree** out) { MPR_TRY(return wrap(Tree::Z(), out);) }
int mpr_tree_const(float v, mpr_tree** out) { MPR_TRY(return wrap(Tree(v), out);) }
int mpr_tree_unary(int op, const mpr_tree* a, mpr_tree** out)
{
    if (!a) return mpr::set_error(MPR_ERR_INVALID, "null operand");
    MPR_TRY(return wrap(Tree::unary((mpr::front::Op)op, a->t), out);)
}
int mpr_tree_binary(int op, const mpr_tree* a, const mpr_tree* b, mpr_tree** out)
{
    if (!a || !b) return mpr::set_error(MPR_ERR_INVALID, "null operand");
    MPR_TRY(return wrap(Tree::binary((mpr::front::Op)op, a->t, b->t), out);)
}
int mpr_tree_remap(const mpr_tree* t, const mpr_tree* x, const mpr_tree* y, const mpr_tree* z,
                   mpr_tree** out)
{
    if (!t || !x || !y || !z) return mpr::set_error(MPR_ERR_INVALID, "null operand");
    MPR_TRY(return wrap(t->t.remap(x->t, y->t, z->t), out);)
}
int mpr_tree_from_frep(const void* bytes, size_t n, mpr_tree** out)
{
    if (!bytes || !out) return mpr::set_error(MPR_ERR_INVALID, "null argument");
    try {
        return wrap(mpr::front::deserialize_frep((const uint8_t*)bytes, n), out);
    } catch (const std::exception& e) {
        return mpr::set_error(MPR_ERR_PARSE, e.what());
    }
}
int mpr_tree_from_frep_file(const char* path, mpr_tree** out)
{
    if (!path || !out) return mpr::set_error(MPR_ERR_INVALID, "null argument");
    try {
        return wrap(mpr::front::load_frep(path), out);
    } catch (const std::exception& e) {
        return mpr::set_error(MPR_ERR_PARSE, e.what());
    }
}
int mpr_tree_to_frep(const mpr_tree* t, void* bytes, size_t cap, size_t* n)
{
    if (!t || !n) return mpr::set_error(MPR_ERR_INVALID, "null argument");
    MPR_TRY(
        auto v = mpr::front::serialize_frep(t->t);
        *n = v.size();
        if (bytes && cap >= v.size()) std::memcpy(bytes, v.data(), v.size());
        return MPR_OK;)
}
int mpr_tree_size(const mpr_tree* t, size_t* nodes)
{
    if (!t || !nodes) return mpr::set_error(MPR_ERR_INVALID, "null argument");
    MPR_TRY(*nodes = t->t.size(); return MPR_OK;)
}
void mpr_tree_free(mpr_tree* t) { delete t; }

/* ---- tape ---- */
static void finish_tape(mpr_tape* t)
{
    int max_slot = 0, choices = 0;
    for (size_t i = 0; i < t->clauses.size(); ++i) {
        const uint64_t c = t->clauses[i];
        max_slot = std::max<int>(max_slot, std::max<int>(mpr_cl_out(c), std::max<int>(mpr_cl_lhs(c), mpr_cl_rhs(c))));
        if (i > 0 && i + 1 < t->clauses.size() && mpr_op_is_minmax(mpr_cl_op(c))) choices++;
        if (mpr_cl_op(c) == MPR_OP_ASIN_LHS || mpr_cl_op(c) == MPR_OP_ACOS_LHS) t->loose_ok = false;
        if (mpr_cl_op(c) == MPR_OP_DIV_LHS_IMM) {
            const uint32_t mag = mpr_cl_immbits(c) & 0x7FFFFFFFu;
            if (mag < 0x0D800000u || mag > 0x71800000u) t->loose_ok = false;
        }
    }
    t->num_slots = max_slot + 1;
    t->num_choices = choices;
    t->schedule = mpr::build_schedule(t->clauses.data(), (int32_t)t->clauses.size());
    /* the tape's walks as machine code, here and not in the first frame that renders it (bear: 3 ms on a GPU box's host — nine
     * scheduled forward walks side by side, tile_gen.cpp: build_tape_code —, 30 on one slow core; round 6 began at 70: interval_gen.cpp:
     * schedule_region; scripts/tape_times.py) */
    t->code = mpr::build_tape_code(t->clauses.data(), (int)t->clauses.size(), mpr::TAPE_CODE_DEFAULT_MIN_RUN);
    if (!t->code && t->num_slots > mpr::TILE_GEN_MAX_SLOTS && t->num_slots <= 94) {
        /* (prospero: 32 000 + 48 000 instructions, 25 of the 36 ms its tape takes to make; architecture: 6800 + 19 000, 4 of 7 ms) */
        /* the backward walk: any such tape (it reads whatever forward walk's choices); the loose forward walk: tapes the loose arithmetic takes */
        /* (side by side: the forward walk is a third of the two) */
        std::future<mpr::IntervalCode> fwd;
        if (t->loose_ok)
            fwd = std::async(std::launch::async, [t]() { return mpr::interval_gen_build(t->clauses.data(), (int)t->clauses.size(), mpr::IW_FIRST_MASKS, true); });
        std::vector<uint32_t> bw = mpr::tile_gen_build_big_backward(t->clauses.data(), (int)t->clauses.size());
        if (!bw.empty()) {
            t->big_bwd = std::make_shared<const std::vector<uint32_t>>(std::move(bw));
            for (size_t i = 1; i < t->clauses.size(); ++i)
                if (mpr_cl_op(t->clauses[i]) == MPR_OP_INVALID) { t->big_end = (int32_t)i; break; }
        }
        if (t->loose_ok) {
            const mpr::IntervalCode ic = fwd.get();
            if (ic.ok) {
                t->big_fwd = std::make_shared<const std::vector<uint32_t>>(ic.words);
                t->big_end = ic.walk_words;
            }
        }
    }
    static std::atomic<uint64_t> serial{1};
    t->serial = serial++;
}

int mpr_tape_from_tree(const mpr_tree* tr, mpr_tape** out)
{
    if (!tr || !out) return mpr::set_error(MPR_ERR_INVALID, "null argument");
    MPR_TRY(
        mpr::front::TapeBuild tb = mpr::front::build_tape(tr->t);
        if (!tb.error.empty()) return mpr::set_error(MPR_ERR_INVALID, tb.error);
        /* The reference prints "Ran out of slots!" and carries on with slot 0 (src/tape.cpp:79-81), which
         * silently evaluates something else: slot 0 means "no operand, take the immediate" to every
         * interpreter (src/context.cu:418-424).  Refuse the expression instead. */
        if (tb.slots_exhausted)
            return mpr::set_error(MPR_ERR_UNSUPPORTED, "expression needs more than 254 live values at once (src/tape.cpp:79: \"Ran out of slots!\")");
        auto* t = new mpr_tape();
        t->clauses = std::move(tb.clauses);
        t->flags = tb.unsupported ? 2 : 0;          /* bit 0 (slots exhausted) is never set: such an expression is refused above */
        finish_tape(t);
        *out = t;
        return MPR_OK;)
}
/* dependency-level schedule of the tape (tape_schedule.hpp): number of levels, widest level, and
 * for every body clause its level (levels may be null) */
int mpr_tape_schedule_info(const mpr_tape* t, int32_t* nlevels, int32_t* max_width, int32_t* levels)
{
    if (!t) return mpr::set_error(MPR_ERR_INVALID, "null tape");
    const mpr::TapeSchedule& sc = t->schedule;
    if (!sc.ok) return mpr::set_error(MPR_ERR_UNSUPPORTED, "tape has no level schedule (too long, or jumps inside)");
    const int32_t nl = (int32_t)sc.level_start.size() - 1;
    if (nlevels) *nlevels = nl;
    int32_t mw = 0;
    for (int32_t l = 0; l < nl; ++l) {
        const int32_t w = sc.level_start[(size_t)l + 1] - sc.level_start[(size_t)l];
        mw = std::max(mw, w);
        if (levels) for (int32_t k = sc.level_start[(size_t)l]; k < sc.level_start[(size_t)l + 1]; ++k) levels[sc.recs[(size_t)k].idx] = l;
    }
    if (max_width) *max_width = mw;
    return MPR_OK;
}
int mpr_tape_from_clauses(const uint64_t* clauses, int32_t length, mpr_tape** out)
{
    if (!clauses || !out || length < 2) return mpr::set_error(MPR_ERR_INVALID, "bad tape");
    if (mpr_cl_op(clauses[0]) != 0 || mpr_cl_op(clauses[length - 1]) != 0)
        return mpr::set_error(MPR_ERR_INVALID, "tape must start with a head clause and end with an end clause");
    for (int32_t i = 1; i + 1 < length; ++i) {
        const uint32_t op = mpr_cl_op(clauses[i]);
        if (op < MPR_OP_SQUARE_LHS || op >= MPR_OP_COUNT)
            return mpr::set_error(MPR_ERR_INVALID, "invalid opcode in tape at clause " + std::to_string(i));
        /* Slot 0 is "no operand": the interpreters read `slot ? value : immediate` (as the reference's tape
         * pushing does, src/context.cu:418-424), so a register operand numbered 0 would be taken for an
         * immediate.  mpr::Tape never produces one (src/tape.cpp:72-87 starts at slot 1). */
        const bool needs_lhs = op != MPR_OP_SUB_IMM_RHS && op != MPR_OP_DIV_IMM_RHS && op != MPR_OP_COPY_IMM && op != MPR_OP_COPY_RHS;
        const bool needs_rhs = op == MPR_OP_ADD_LHS_RHS || op == MPR_OP_MUL_LHS_RHS || op == MPR_OP_MIN_LHS_RHS || op == MPR_OP_MAX_LHS_RHS ||
                               op == MPR_OP_SUB_LHS_RHS || op == MPR_OP_DIV_LHS_RHS || op == MPR_OP_SUB_IMM_RHS || op == MPR_OP_DIV_IMM_RHS ||
                               op == MPR_OP_COPY_RHS;
        if ((needs_lhs && mpr_cl_lhs(clauses[i]) == 0) || (needs_rhs && mpr_cl_rhs(clauses[i]) == 0) || mpr_cl_out(clauses[i]) == 0)
            return mpr::set_error(MPR_ERR_INVALID, "clause " + std::to_string(i) + " names slot 0 as a register (slot 0 means: no operand)");
    }
    MPR_TRY(
        auto* t = new mpr_tape();
        t->clauses.assign(clauses, clauses + length);
        finish_tape(t);
        *out = t;
        return MPR_OK;)
}
int32_t mpr_tape_length(const mpr_tape* t) { return t ? (int32_t)t->clauses.size() : 0; }
const uint64_t* mpr_tape_data(const mpr_tape* t) { return t ? t->clauses.data() : nullptr; }
int32_t mpr_tape_num_slots(const mpr_tape* t) { return t ? t->num_slots : 0; }
int32_t mpr_tape_num_choices(const mpr_tape* t) { return t ? t->num_choices : 0; }
int32_t mpr_tape_flags(const mpr_tape* t) { return t ? t->flags : 0; }
/* frame_domain.hpp: 1 if every interval operation of the tape stays, over the whole view, where the reference's interval routines are
 * inclusion-isotone (frames of such a view may start at the 16^3 tiles and take the loose enclosures), 0 if not, < 0: an error */
int mpr_tape_frame_is_tame(const mpr_tape* t, int dim, const float* mat, float z, double* trace)
{
    if (!t || !mat || (dim != 2 && dim != 3)) return mpr::set_error(MPR_ERR_INVALID, "null argument or bad dimension");
    return mpr::frame_is_tame(t->clauses.data(), (int)t->clauses.size(), dim, mat, z, trace) ? 1 : 0;
}
void mpr_tape_free(mpr_tape* t) { delete t; }

/* ---- column partition: longest-processing-time-first deal ---- */
int mpr_partition_columns(int32_t columns, const float* weights, int32_t nranks, int32_t* owner)
{
    if (columns <= 0 || nranks <= 0 || !owner) return mpr::set_error(MPR_ERR_INVALID, "bad partition arguments");
    MPR_TRY(
        if (!weights) {
            for (int32_t c = 0; c < columns; ++c) owner[c] = c % nranks;
            return MPR_OK;
        }
        std::vector<int32_t> order(columns);
        std::iota(order.begin(), order.end(), 0);
        /* heaviest first; ties by index so that every rank computes the same deal */
        std::stable_sort(order.begin(), order.end(),
                         [&](int32_t a, int32_t b) { return weights[a] > weights[b]; });
        std::vector<double> load(nranks, 0.0);
        std::vector<int32_t> cnt(nranks, 0);
        for (int32_t c : order) {
            int best = 0;
            for (int r = 1; r < nranks; ++r) {
                if (load[r] < load[best] || (load[r] == load[best] && cnt[r] < cnt[best])) best = r;
            }
            owner[c] = best;
            load[best] += weights[c] > 0 ? weights[c] : 0;
            cnt[best]++;
        }
        return MPR_OK;)
}

}  // extern "C"
