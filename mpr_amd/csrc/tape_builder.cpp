/*
 * tape_builder.cpp — expression DAG -> flat tape of 64-bit clauses (host side of mpr::Tape).
 *
 * Follows the behaviour of the reference's Tape::Tape(const libfive::Tree&),
 * src/tape.cpp:21-228:
 *   - walk the DAG in orderedDfs() order (:25); constants never get a slot, X/Y/Z are
 *     remembered (:31-40); every supported operation records "last use" of its operands
 *     (:42-66)
 *   - slot 0 is reserved; output slots come LIFO from a free list, else a fresh slot, and at
 *     255 slots the builder reports "ran out of slots" and uses slot 0 (:68-87)
 *   - the head clause (op 0) carries the slots bound to X, Y, Z in bytes 1..3 (:89-99)
 *   - opcode variants: commutative ops keep the non-constant operand in lhs (:134-155);
 *     non-commutative ops have IMM_RHS / LHS_IMM / LHS_RHS forms (:157-176)
 *   - operand slots are released at their last use BEFORE the output slot is chosen, so
 *     in-place clauses (out == lhs) occur (:198-212)
 *   - the end clause (op 0) carries the root's slot in byte 1 (:214-221)
 * Unlike the reference this returns error information instead of printing to stderr.
 */
#include "tape_builder.hpp"

#include <cstring>
#include <map>
#include <unordered_map>

#include "../../include/mpr_clause.h"

namespace mpr {
namespace front {

static uint32_t fbits(float f)
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

TapeBuild build_tape(const Tree& tree)
{
    TapeBuild tb;
    if (!tree.valid()) {
        tb.error = "empty tree";
        return tb;
    }
    const std::vector<Tree> ordered = tree.orderedDfs();

    std::vector<const Node*> ordered_fast;
    ordered_fast.reserve(ordered.size());
    std::unordered_map<const Node*, const Node*> last_used;
    const Node* axes_used[3] = {nullptr, nullptr, nullptr};

    for (auto& c : ordered) {
        switch (c->op) {
            case CONSTANT: continue;
            case VAR_X: axes_used[0] = c.id(); break;
            case VAR_Y: axes_used[1] = c.id(); break;
            case VAR_Z: axes_used[2] = c.id(); break;
            case OP_ADD: case OP_MUL: case OP_MIN: case OP_MAX: case OP_SUB: case OP_DIV:
                last_used[c->rhs.get()] = c.id();
                /* FALLTHROUGH */
            case OP_SQUARE: case OP_SQRT: case OP_NEG: case OP_SIN: case OP_COS: case OP_ASIN:
            case OP_ACOS: case OP_ATAN: case OP_EXP: case OP_ABS: case OP_LOG:
                last_used[c->lhs.get()] = c.id();
                ordered_fast.push_back(c.id());
                break;
            default:
                tb.warnings += "unsupported opcode " + std::to_string((int)c->op) + "; ";
                tb.unsupported++;
                break;
        }
    }

    std::vector<uint8_t> free_slots;
    std::unordered_map<const Node*, uint8_t> bound_slots;
    unsigned num_slots = 1;

    auto get_slot = [&](const Node* id) -> uint8_t {
        uint8_t out = 0;
        if (!free_slots.empty()) {
            out = free_slots.back();
            free_slots.pop_back();
        } else if (num_slots == 255) {
            tb.slots_exhausted = true;
        } else {
            out = (uint8_t)num_slots++;
        }
        bound_slots[id] = out;
        return out;
    };
    auto get_reg = [&](const Node* n) -> uint8_t {
        auto it = bound_slots.find(n);
        if (it != bound_slots.end()) return it->second;
        tb.warnings += "unbound operand; ";
        return 0;
    };

    uint32_t axis_slot[3] = {0, 0, 0};
    for (unsigned i = 0; i < 3; ++i)
        if (axes_used[i] != nullptr) axis_slot[i] = get_slot(axes_used[i]);
    tb.clauses.reserve(ordered_fast.size() + 2);
    tb.clauses.push_back(mpr_cl_make(0, axis_slot[0], axis_slot[1], axis_slot[2], 0));

    for (const Node* c : ordered_fast) {
        uint32_t op = 0, lhs = 0, rhs = 0, imm = 0;
        const bool lc = c->lhs && c->lhs->op == CONSTANT;
        const bool rc = c->rhs && c->rhs->op == CONSTANT;
        auto unary = [&](uint32_t o) { op = o; lhs = get_reg(c->lhs.get()); };
        auto commutative = [&](uint32_t o_imm, uint32_t o_rhs) {
            if (lc) { op = o_imm; lhs = get_reg(c->rhs.get()); imm = fbits(c->lhs->value); }
            else if (rc) { op = o_imm; lhs = get_reg(c->lhs.get()); imm = fbits(c->rhs->value); }
            else { op = o_rhs; lhs = get_reg(c->lhs.get()); rhs = get_reg(c->rhs.get()); }
        };
        auto noncommutative = [&](uint32_t o_lhs_imm, uint32_t o_imm_rhs, uint32_t o_lhs_rhs) {
            if (lc) { op = o_imm_rhs; rhs = get_reg(c->rhs.get()); imm = fbits(c->lhs->value); }
            else if (rc) { op = o_lhs_imm; lhs = get_reg(c->lhs.get()); imm = fbits(c->rhs->value); }
            else { op = o_lhs_rhs; lhs = get_reg(c->lhs.get()); rhs = get_reg(c->rhs.get()); }
        };
        switch (c->op) {
            case OP_SQUARE: unary(MPR_OP_SQUARE_LHS); break;
            case OP_SQRT: unary(MPR_OP_SQRT_LHS); break;
            case OP_NEG: unary(MPR_OP_NEG_LHS); break;
            case OP_SIN: unary(MPR_OP_SIN_LHS); break;
            case OP_COS: unary(MPR_OP_COS_LHS); break;
            case OP_ASIN: unary(MPR_OP_ASIN_LHS); break;
            case OP_ACOS: unary(MPR_OP_ACOS_LHS); break;
            case OP_ATAN: unary(MPR_OP_ATAN_LHS); break;
            case OP_EXP: unary(MPR_OP_EXP_LHS); break;
            case OP_ABS: unary(MPR_OP_ABS_LHS); break;
            case OP_LOG: unary(MPR_OP_LOG_LHS); break;
            case OP_ADD: commutative(MPR_OP_ADD_LHS_IMM, MPR_OP_ADD_LHS_RHS); break;
            case OP_MUL: commutative(MPR_OP_MUL_LHS_IMM, MPR_OP_MUL_LHS_RHS); break;
            case OP_MIN: commutative(MPR_OP_MIN_LHS_IMM, MPR_OP_MIN_LHS_RHS); break;
            case OP_MAX: commutative(MPR_OP_MAX_LHS_IMM, MPR_OP_MAX_LHS_RHS); break;
            case OP_SUB:
                noncommutative(MPR_OP_SUB_LHS_IMM, MPR_OP_SUB_IMM_RHS, MPR_OP_SUB_LHS_RHS);
                break;
            case OP_DIV:
                noncommutative(MPR_OP_DIV_LHS_IMM, MPR_OP_DIV_IMM_RHS, MPR_OP_DIV_LHS_RHS);
                break;
            default: break;
        }

        /* release operand slots at their last use, before choosing the output slot */
        for (const Node* h : {c->lhs.get(), c->rhs.get()}) {
            if (h != nullptr && h->op != CONSTANT) {
                auto lu = last_used.find(h);
                if (lu != last_used.end() && lu->second == c) {
                    auto it = bound_slots.find(h);
                    if (it != bound_slots.end()) {
                        free_slots.push_back(it->second);
                        bound_slots.erase(it);
                    }
                }
            }
        }
        const uint32_t out = get_slot(c);
        tb.clauses.push_back(mpr_cl_make(op, out, lhs, rhs, imm));
    }

    /* end clause: result slot of the root.  A root that is a bare constant or axis has no
     * clause of its own; an axis reads its bound slot, a constant has none (slot 0). */
    const Node* root = ordered.back().id();
    uint32_t root_slot = 0;
    if (root->op != CONSTANT) root_slot = get_reg(root);
    tb.clauses.push_back(mpr_cl_make(0, root_slot, 0, 0, 0));
    tb.num_slots = (int)num_slots;
    return tb;
}

}  // namespace front
}  // namespace mpr
