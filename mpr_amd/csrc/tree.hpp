/*
 * tree.hpp — minimal expression-DAG front end: the part of libfive that mpr's hot path
 * consumes (a `libfive::Tree` goes into `mpr::Tape`, src/tape.cpp:21-66).
 *
 * libfive is an un-vendored, empty submodule in the reference tree (.gitmodules:1-3), so
 * this is NOT a restatement of libfive: it is a small stand-in offering the operations the
 * reference's call sites use (benchmark/circle.cpp:22-24, render_2d_table.cpp:41-45,
 * print_tape_table.cpp:29, render_effects.cpp:36) plus a reader for the `.frep` archives in
 * benchmark/files (format reverse-engineered from the six files).
 *
 * Like libfive's Cache, nodes are hash-consed (structurally identical sub-expressions are
 * one node) and a few algebraic identities are applied on construction.  libfive's exact
 * simplification / re-balancing rules are not reproduced; parity is therefore stated "on
 * the same tape" (DESIGN.md).
 */
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace mpr {
namespace front {

/* libfive Opcode numbering with LIBFIVE_PACKED_OPCODES (CMakeLists.txt:6-8), as found in
 * the .frep files */
enum Op : uint8_t {
    INVALID = 0, CONSTANT = 1, VAR_X = 2, VAR_Y = 3, VAR_Z = 4, VAR_FREE = 5, CONST_VAR = 6,
    OP_SQUARE = 7, OP_SQRT = 8, OP_NEG = 9, OP_SIN = 10, OP_COS = 11, OP_TAN = 12,
    OP_ASIN = 13, OP_ACOS = 14, OP_ATAN = 15, OP_EXP = 16, OP_ABS = 17, OP_LOG = 18,
    OP_RECIP = 19,
    OP_ADD = 20, OP_MUL = 21, OP_MIN = 22, OP_MAX = 23, OP_SUB = 24, OP_DIV = 25,
    OP_ATAN2 = 26, OP_POW = 27, OP_NTH_ROOT = 28, OP_MOD = 29, OP_NANFILL = 30,
    OP_COMPARE = 31, LAST_OP = 32
};
int op_args(Op op);   /* 0, 1 or 2; -1 if unknown */

struct Node {
    Op op = INVALID;
    float value = 0.0f;
    std::shared_ptr<const Node> lhs, rhs;
};
using NodePtr = std::shared_ptr<const Node>;

class Tree {
public:
    Tree() = default;
    Tree(float v);
    Tree(double v) : Tree((float)v) {}
    Tree(int v) : Tree((float)v) {}
    explicit Tree(NodePtr p) : ptr(std::move(p)) {}

    static Tree X();
    static Tree Y();
    static Tree Z();
    static Tree unary(Op op, const Tree& a);
    static Tree binary(Op op, const Tree& a, const Tree& b);

    /* raw constructor used by the .frep reader: hash-consing only, no simplification */
    static Tree raw(Op op, float value, const Tree& a, const Tree& b);

    const Node* operator->() const { return ptr.get(); }
    const Node* id() const { return ptr.get(); }
    bool valid() const { return (bool)ptr; }

    /* dependency order, depth first, lhs before rhs, every node once (what
     * libfive::Tree::orderedDfs() yields at src/tape.cpp:25; the lhs-first post-order is
     * what benchmark/brute.cu:39-61, a dump of such a walk, shows) */
    std::vector<Tree> orderedDfs() const;

    /* substitute X, Y, Z (benchmark/render_effects.cpp:36) */
    Tree remap(const Tree& x, const Tree& y, const Tree& z) const;

    size_t size() const;   /* number of distinct nodes */

    NodePtr ptr;
};

Tree operator+(const Tree& a, const Tree& b);
Tree operator-(const Tree& a, const Tree& b);
Tree operator*(const Tree& a, const Tree& b);
Tree operator/(const Tree& a, const Tree& b);
Tree operator-(const Tree& a);
Tree min(const Tree& a, const Tree& b);
Tree max(const Tree& a, const Tree& b);
Tree sqrt(const Tree& a);
Tree square(const Tree& a);
Tree abs(const Tree& a);
Tree sin(const Tree& a);
Tree cos(const Tree& a);
Tree asin(const Tree& a);
Tree acos(const Tree& a);
Tree atan(const Tree& a);
Tree exp(const Tree& a);
Tree log(const Tree& a);

/* Parse a libfive Archive (`.frep`): returns the tree of the first shape.  Throws
 * std::runtime_error with a description on malformed input. */
Tree deserialize_frep(const uint8_t* bytes, size_t n);
Tree load_frep(const std::string& path);
/* Serialize in the same format (gui/main.cpp:394-403 "Save shape.frep"). */
std::vector<uint8_t> serialize_frep(const Tree& t);

}  // namespace front
}  // namespace mpr
