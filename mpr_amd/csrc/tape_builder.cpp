/*
 * tape_builder.cpp — expression DAG -> flat tape of 64-bit clauses (host side of mpr::Tape).
 *
 * Produces the tape the reference's Tape::Tape(const libfive::Tree&) produces for the same DAG
 * (src/tape.cpp:21-228) — the slot numbers are part of that, so the allocation ORDER is the
 * reference's: slot 0 reserved; X, Y, Z (those that occur) take the first slots (:89-99, the head
 * clause names them); operations are emitted in dependency order (:25); an operand's slot goes
 * back on a LIFO free list at its last reader and BEFORE the reader's own output slot is taken,
 * so in-place clauses occur (:198-212); a fresh slot only when the list is empty; at 255 slots
 * the tape is marked "out of slots" and slot 0 is handed out (:68-87); constants never get a slot
 * but become the immediate of the IMM form of their reader (:134-176); the end clause names the
 * root's slot (:214-221).
 *
 * Shape: the DAG is numbered in dependency order once, and everything else is arrays over those
 * numbers (operand numbers, last reader, slot) plus one table that maps a tree operation to its
 * clause opcodes.  Errors come back in TapeBuild instead of going to stderr.
 */
#include "tape_builder.hpp"

#include <cstring>
#include <unordered_map>

#include "../../include/mpr_clause.h"

namespace mpr {
namespace front {
namespace {

/* how a tree operation is spelt in clauses */
enum Shape : uint8_t { NOT_EMITTED, UNARY, COMMUTATIVE, ORDERED };
struct Spelling {
    Shape shape = NOT_EMITTED;
    uint8_t plain = 0;        /* UNARY: the opcode; binary: both operands in slots */
    uint8_t const_right = 0;  /* binary: lhs in a slot, rhs constant */
    uint8_t const_left = 0;   /* ORDERED only: lhs constant, rhs in a slot */
};
struct SpellingTable {
    Spelling of[LAST_OP];
    constexpr SpellingTable() : of{}
    {
        constexpr struct { Op op; uint8_t code; } unary[] = {
            {OP_SQUARE, MPR_OP_SQUARE_LHS}, {OP_SQRT, MPR_OP_SQRT_LHS}, {OP_NEG, MPR_OP_NEG_LHS}, {OP_SIN, MPR_OP_SIN_LHS},
            {OP_COS, MPR_OP_COS_LHS}, {OP_ASIN, MPR_OP_ASIN_LHS}, {OP_ACOS, MPR_OP_ACOS_LHS}, {OP_ATAN, MPR_OP_ATAN_LHS},
            {OP_EXP, MPR_OP_EXP_LHS}, {OP_ABS, MPR_OP_ABS_LHS}, {OP_LOG, MPR_OP_LOG_LHS}};
        for (const auto& u : unary) of[u.op] = Spelling{UNARY, u.code, 0, 0};
        /* a commutative operation with a constant keeps the other operand in lhs whichever side it was on */
        of[OP_ADD] = Spelling{COMMUTATIVE, MPR_OP_ADD_LHS_RHS, MPR_OP_ADD_LHS_IMM, MPR_OP_ADD_LHS_IMM};
        of[OP_MUL] = Spelling{COMMUTATIVE, MPR_OP_MUL_LHS_RHS, MPR_OP_MUL_LHS_IMM, MPR_OP_MUL_LHS_IMM};
        of[OP_MIN] = Spelling{COMMUTATIVE, MPR_OP_MIN_LHS_RHS, MPR_OP_MIN_LHS_IMM, MPR_OP_MIN_LHS_IMM};
        of[OP_MAX] = Spelling{COMMUTATIVE, MPR_OP_MAX_LHS_RHS, MPR_OP_MAX_LHS_IMM, MPR_OP_MAX_LHS_IMM};
        of[OP_SUB] = Spelling{ORDERED, MPR_OP_SUB_LHS_RHS, MPR_OP_SUB_LHS_IMM, MPR_OP_SUB_IMM_RHS};
        of[OP_DIV] = Spelling{ORDERED, MPR_OP_DIV_LHS_RHS, MPR_OP_DIV_LHS_IMM, MPR_OP_DIV_IMM_RHS};
    }
};
constexpr SpellingTable SPELL;

uint32_t bits_of(float f)
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

constexpr int NONE = -1;

}  // namespace

TapeBuild build_tape(const Tree& tree)
{
    TapeBuild tb;
    if (!tree.valid()) {
        tb.error = "empty tree";
        return tb;
    }

    /* ---- number the DAG in dependency order ---- */
    const std::vector<Tree> order = tree.orderedDfs();
    const int n = (int)order.size();
    std::unordered_map<const Node*, int> number;
    number.reserve((size_t)n * 2);
    for (int i = 0; i < n; ++i) number.emplace(order[(size_t)i].id(), i);
    auto number_of = [&](const NodePtr& p) {
        if (!p) return NONE;
        const auto it = number.find(p.get());
        return it == number.end() ? NONE : it->second;
    };
    std::vector<int> left((size_t)n, NONE), right((size_t)n, NONE);
    std::vector<int> last_reader((size_t)n, NONE);      /* the last emitted operation that reads this node */
    std::vector<int> emitted;
    int axis_node[3] = {NONE, NONE, NONE};
    for (int i = 0; i < n; ++i) {
        const Node* const node = order[(size_t)i].id();
        if (node->op == CONSTANT) continue;
        if (node->op >= VAR_X && node->op <= VAR_Z) {
            axis_node[node->op - VAR_X] = i;
            continue;
        }
        const Shape shape = node->op < LAST_OP ? SPELL.of[node->op].shape : NOT_EMITTED;
        if (shape == NOT_EMITTED) {
            tb.warnings += "unsupported opcode " + std::to_string((int)node->op) + "; ";
            tb.unsupported++;
            continue;
        }
        left[(size_t)i] = number_of(node->lhs);
        right[(size_t)i] = shape == UNARY ? NONE : number_of(node->rhs);
        /* (the rhs is recorded first: when both operands are one node the result is the same either way) */
        if (right[(size_t)i] != NONE) last_reader[(size_t)right[(size_t)i]] = i;
        if (left[(size_t)i] != NONE) last_reader[(size_t)left[(size_t)i]] = i;
        emitted.push_back(i);
    }

    /* ---- slots ---- */
    std::vector<int> slot_of((size_t)n, NONE);
    std::vector<uint8_t> returned;                     /* LIFO */
    unsigned slots_made = 1;                           /* slot 0 is nobody's */
    auto take_slot = [&](int node) {
        unsigned s = 0;
        if (!returned.empty()) {
            s = returned.back();
            returned.pop_back();
        } else if (slots_made == 255) {
            tb.slots_exhausted = true;
        } else {
            s = slots_made++;
        }
        slot_of[(size_t)node] = (int)s;
        return s;
    };
    auto slot_read = [&](int node) -> uint32_t {
        if (node != NONE && slot_of[(size_t)node] != NONE) return (uint32_t)slot_of[(size_t)node];
        tb.warnings += "unbound operand; ";
        return 0;
    };
    auto is_constant = [&](int node) { return node != NONE && order[(size_t)node]->op == CONSTANT; };
    auto constant_bits = [&](int node) { return bits_of(order[(size_t)node]->value); };

    uint32_t axis_slot[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k)
        if (axis_node[k] != NONE) axis_slot[k] = take_slot(axis_node[k]);
    tb.clauses.reserve(emitted.size() + 2);
    tb.clauses.push_back(mpr_cl_make(0, axis_slot[0], axis_slot[1], axis_slot[2], 0));

    /* ---- one clause per emitted operation ---- */
    for (const int i : emitted) {
        const Spelling& sp = SPELL.of[order[(size_t)i]->op];
        const int a = left[(size_t)i], b = right[(size_t)i];
        uint32_t opcode = sp.plain, lhs = 0, rhs = 0, imm = 0;
        if (sp.shape == UNARY) {
            lhs = slot_read(a);
        } else if (is_constant(a)) {
            opcode = sp.const_left;
            imm = constant_bits(a);
            if (sp.shape == COMMUTATIVE) lhs = slot_read(b);
            else rhs = slot_read(b);
        } else if (is_constant(b)) {
            opcode = sp.const_right;
            imm = constant_bits(b);
            lhs = slot_read(a);
        } else {
            lhs = slot_read(a);
            rhs = slot_read(b);
        }
        /* operands read here for the last time give their slots back first: the output may take one of them */
        for (const int operand : {a, b}) {
            if (operand == NONE || is_constant(operand)) continue;
            if (last_reader[(size_t)operand] == i && slot_of[(size_t)operand] != NONE) {
                returned.push_back((uint8_t)slot_of[(size_t)operand]);
                slot_of[(size_t)operand] = NONE;
            }
        }
        const uint32_t out = take_slot(i);
        tb.clauses.push_back(mpr_cl_make(opcode, out, lhs, rhs, imm));
    }

    /* end clause: the root's slot.  A root that is a bare axis reads the axis' slot; a bare constant has none (0). */
    const int root = n - 1;
    const uint32_t root_slot = is_constant(root) ? 0u : slot_read(root);
    tb.clauses.push_back(mpr_cl_make(0, root_slot, 0, 0, 0));
    tb.num_slots = (int)slots_made;
    return tb;
}

}  // namespace front
}  // namespace mpr
