/*
 * tree.cpp — see tree.hpp.  Host-only C++17.
 */
#include "tree.hpp"

#include <cmath>
#include <cstring>
#include <fstream>
#include <mutex>
#include <stdexcept>
#include <unordered_map>
#include <unordered_set>

namespace mpr {
namespace front {

int op_args(Op op)
{
    if (op == CONSTANT || op == VAR_X || op == VAR_Y || op == VAR_Z || op == VAR_FREE) return 0;
    if (op == CONST_VAR) return 1;
    if (op >= OP_SQUARE && op <= OP_RECIP) return 1;
    if (op >= OP_ADD && op <= OP_COMPARE) return 2;
    return -1;
}

namespace {

struct Key {
    uint8_t op;
    uint32_t bits;
    const Node* l;
    const Node* r;
    bool operator==(const Key& o) const { return op == o.op && bits == o.bits && l == o.l && r == o.r; }
};
struct KeyHash {
    size_t operator()(const Key& k) const
    {
        size_t h = k.op * 0x9E3779B97F4A7C15ull;
        h ^= (size_t)k.bits + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
        h ^= (size_t)(uintptr_t)k.l + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
        h ^= (size_t)(uintptr_t)k.r + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
        return h;
    }
};

/* The hash-consing table (the role of libfive::Cache::instance(), src/tape.cpp:23).  Weak
 * references, so nodes die with the last Tree that uses them. */
struct Cache {
    std::mutex mut;
    std::unordered_map<Key, std::weak_ptr<const Node>, KeyHash> map;
    size_t since_sweep = 0;

    NodePtr get(Op op, float value, const NodePtr& l, const NodePtr& r)
    {
        uint32_t bits;
        std::memcpy(&bits, &value, 4);
        if (op != CONSTANT) bits = 0;
        Key k{(uint8_t)op, bits, l.get(), r.get()};
        std::lock_guard<std::mutex> lock(mut);
        auto it = map.find(k);
        if (it != map.end()) {
            if (auto sp = it->second.lock()) return sp;
        }
        auto n = std::make_shared<Node>();
        n->op = op;
        n->value = (op == CONSTANT) ? value : 0.0f;
        n->lhs = l;
        n->rhs = r;
        map[k] = n;
        if (++since_sweep > 65536) {
            since_sweep = 0;
            for (auto i = map.begin(); i != map.end();) {
                if (i->second.expired()) i = map.erase(i); else ++i;
            }
        }
        return n;
    }
};
Cache& cache()
{
    static Cache c;
    return c;
}

bool is_const(const Tree& t) { return t->op == CONSTANT; }
bool is_const(const Tree& t, float v) { return t->op == CONSTANT && t->value == v; }

}  // namespace

Tree::Tree(float v) : ptr(cache().get(CONSTANT, v, nullptr, nullptr)) {}
Tree Tree::X() { return Tree(cache().get(VAR_X, 0, nullptr, nullptr)); }
Tree Tree::Y() { return Tree(cache().get(VAR_Y, 0, nullptr, nullptr)); }
Tree Tree::Z() { return Tree(cache().get(VAR_Z, 0, nullptr, nullptr)); }

Tree Tree::raw(Op op, float value, const Tree& a, const Tree& b)
{
    return Tree(cache().get(op, value, a.ptr, b.ptr));
}

Tree Tree::unary(Op op, const Tree& a)
{
    if (op_args(op) != 1 || !a.valid()) throw std::runtime_error("Tree::unary: bad opcode/operand");
    if (is_const(a)) {
        const float v = a->value;
        switch (op) {
            case OP_NEG: return Tree(-v);
            case OP_SQUARE: return Tree(v * v);
            case OP_SQRT: return Tree(std::sqrt(v));
            case OP_ABS: return Tree(std::fabs(v));
            default: break;
        }
    }
    return raw(op, 0, a, Tree());
}

Tree Tree::binary(Op op, const Tree& a, const Tree& b)
{
    if (op_args(op) != 2 || !a.valid() || !b.valid())
        throw std::runtime_error("Tree::binary: bad opcode/operand");
    if (is_const(a) && is_const(b)) {
        const float x = a->value, y = b->value;
        switch (op) {
            case OP_ADD: return Tree(x + y);
            case OP_SUB: return Tree(x - y);
            case OP_MUL: return Tree(x * y);
            case OP_DIV: return Tree(x / y);
            case OP_MIN: return Tree(std::fmin(x, y));
            case OP_MAX: return Tree(std::fmax(x, y));
            default: break;
        }
    }
    switch (op) {
        case OP_ADD:
            if (is_const(a, 0.0f)) return b;
            if (is_const(b, 0.0f)) return a;
            break;
        case OP_SUB:
            if (is_const(b, 0.0f)) return a;
            if (is_const(a, 0.0f)) return unary(OP_NEG, b);
            break;
        case OP_MUL:
            if (is_const(a, 0.0f)) return a;
            if (is_const(b, 0.0f)) return b;
            if (is_const(a, 1.0f)) return b;
            if (is_const(b, 1.0f)) return a;
            if (is_const(a, -1.0f)) return unary(OP_NEG, b);
            if (is_const(b, -1.0f)) return unary(OP_NEG, a);
            if (a.id() == b.id()) return unary(OP_SQUARE, a);
            break;
        case OP_DIV:
            /* x / 1 is NOT folded: bear.frep, written by libfive, contains six such divisions, so libfive's
             * own simplifier keeps them (every other identity rule here never fires on the six archives of
             * the reference — scripts: tests/test_host_api.py::test_archives_are_canonical) */
            break;
        case OP_MIN:
        case OP_MAX:
            if (a.id() == b.id()) return a;
            break;
        default: break;
    }
    return raw(op, 0, a, b);
}

std::vector<Tree> Tree::orderedDfs() const
{
    std::vector<Tree> out;
    if (!ptr) return out;
    std::unordered_set<const Node*> seen;
    struct Item { NodePtr n; int state; };
    std::vector<Item> stack;
    stack.push_back({ptr, 0});
    while (!stack.empty()) {
        Item& top = stack.back();
        if (top.state == 0) {
            if (seen.count(top.n.get())) { stack.pop_back(); continue; }
            top.state = 1;
            if (top.n->lhs && !seen.count(top.n->lhs.get())) {
                NodePtr c = top.n->lhs;
                stack.push_back({c, 0});
            }
        } else if (top.state == 1) {
            top.state = 2;
            if (top.n->rhs && !seen.count(top.n->rhs.get())) {
                NodePtr c = top.n->rhs;
                stack.push_back({c, 0});
            }
        } else {
            if (!seen.count(top.n.get())) {
                seen.insert(top.n.get());
                out.emplace_back(top.n);
            }
            stack.pop_back();
        }
    }
    return out;
}

size_t Tree::size() const { return orderedDfs().size(); }

Tree Tree::remap(const Tree& x, const Tree& y, const Tree& z) const
{
    std::unordered_map<const Node*, Tree> done;
    for (auto& t : orderedDfs()) {
        Tree r;
        switch (t->op) {
            case VAR_X: r = x; break;
            case VAR_Y: r = y; break;
            case VAR_Z: r = z; break;
            default: {
                const int n = op_args(t->op);
                if (n == 0) r = t;
                else if (n == 1) r = unary(t->op, done.at(t->lhs.get()));
                else r = binary(t->op, done.at(t->lhs.get()), done.at(t->rhs.get()));
            }
        }
        done.emplace(t.id(), r);
    }
    return done.at(id());
}

Tree operator+(const Tree& a, const Tree& b) { return Tree::binary(OP_ADD, a, b); }
Tree operator-(const Tree& a, const Tree& b) { return Tree::binary(OP_SUB, a, b); }
Tree operator*(const Tree& a, const Tree& b) { return Tree::binary(OP_MUL, a, b); }
Tree operator/(const Tree& a, const Tree& b) { return Tree::binary(OP_DIV, a, b); }
Tree operator-(const Tree& a) { return Tree::unary(OP_NEG, a); }
Tree min(const Tree& a, const Tree& b) { return Tree::binary(OP_MIN, a, b); }
Tree max(const Tree& a, const Tree& b) { return Tree::binary(OP_MAX, a, b); }
Tree sqrt(const Tree& a) { return Tree::unary(OP_SQRT, a); }
Tree square(const Tree& a) { return Tree::unary(OP_SQUARE, a); }
Tree abs(const Tree& a) { return Tree::unary(OP_ABS, a); }
Tree sin(const Tree& a) { return Tree::unary(OP_SIN, a); }
Tree cos(const Tree& a) { return Tree::unary(OP_COS, a); }
Tree asin(const Tree& a) { return Tree::unary(OP_ASIN, a); }
Tree acos(const Tree& a) { return Tree::unary(OP_ACOS, a); }
Tree atan(const Tree& a) { return Tree::unary(OP_ATAN, a); }
Tree exp(const Tree& a) { return Tree::unary(OP_EXP, a); }
Tree log(const Tree& a) { return Tree::unary(OP_LOG, a); }

/* ------------------------------------------------------------------------------------
 * .frep archive:  'T' "name" "doc"  node*  0xFF   ...  0xFF
 *   node := u8 opcode, then  CONSTANT: f32 LE | VAR_*: nothing | unary: u32 operand |
 *           binary: u32 RIGHT operand, u32 LEFT operand      (indices into the node list)
 * ------------------------------------------------------------------------------------ */
namespace {
struct Reader {
    const uint8_t* p;
    size_t n, i = 0;
    uint8_t u8()
    {
        if (i >= n) throw std::runtime_error("frep: unexpected end of data");
        return p[i++];
    }
    uint32_t u32()
    {
        if (i + 4 > n) throw std::runtime_error("frep: unexpected end of data");
        uint32_t v = (uint32_t)p[i] | ((uint32_t)p[i + 1] << 8) | ((uint32_t)p[i + 2] << 16) |
                     ((uint32_t)p[i + 3] << 24);
        i += 4;
        return v;
    }
    std::string str()
    {
        if (u8() != '"') throw std::runtime_error("frep: expected string");
        std::string s;
        for (;;) {
            uint8_t c = u8();
            if (c == '"') break;
            if (c == '\\') c = u8();
            s.push_back((char)c);
        }
        return s;
    }
};
}  // namespace

/* Nodes are rebuilt through Tree::unary / Tree::binary — the same deduplication, identity and constant
 * folding an expression written with the operators gets (libfive's deserializer reconstructs nodes through
 * its Tree constructors and Cache::operation too) — so an archive and the same expression built in code
 * give one tape. */
Tree deserialize_frep(const uint8_t* bytes, size_t n)
{
    Reader r{bytes, n};
    /* optional shape tags before the tree; the files in benchmark/files start with 'T' */
    if (r.u8() != 'T') throw std::runtime_error("frep: missing 'T' shape tag");
    (void)r.str();
    (void)r.str();
    std::vector<Tree> nodes;
    for (;;) {
        const uint8_t opb = r.u8();
        if (opb == 0xFF) break;
        const Op op = (Op)opb;
        const int args = op_args(op);
        if (args < 0 || op == VAR_FREE || op == CONST_VAR)
            throw std::runtime_error("frep: unsupported opcode " + std::to_string(opb));
        if (op == CONSTANT) {
            uint32_t bits = r.u32();
            float v;
            std::memcpy(&v, &bits, 4);
            nodes.push_back(Tree(v));
        } else if (args == 0) {
            nodes.push_back(Tree::raw(op, 0, Tree(), Tree()));
        } else if (args == 1) {
            const uint32_t a = r.u32();
            if (a >= nodes.size()) throw std::runtime_error("frep: forward reference");
            nodes.push_back(Tree::unary(op, nodes[a]));
        } else {
            const uint32_t rhs = r.u32();
            const uint32_t lhs = r.u32();
            if (rhs >= nodes.size() || lhs >= nodes.size())
                throw std::runtime_error("frep: forward reference");
            nodes.push_back(Tree::binary(op, nodes[lhs], nodes[rhs]));
        }
    }
    if (nodes.empty()) throw std::runtime_error("frep: empty tree");
    return nodes.back();
}

Tree load_frep(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("frep: cannot open " + path);
    std::vector<uint8_t> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return deserialize_frep(buf.data(), buf.size());
}

std::vector<uint8_t> serialize_frep(const Tree& t)
{
    std::vector<uint8_t> out = {'T', '"', '"', '"', '"'};
    auto put32 = [&](uint32_t v) {
        for (int k = 0; k < 4; ++k) out.push_back((uint8_t)(v >> (8 * k)));
    };
    std::unordered_map<const Node*, uint32_t> index;
    for (auto& n : t.orderedDfs()) {
        out.push_back((uint8_t)n->op);
        const int args = op_args(n->op);
        if (n->op == CONSTANT) {
            uint32_t bits;
            std::memcpy(&bits, &n->value, 4);
            put32(bits);
        } else if (args == 1) {
            put32(index.at(n->lhs.get()));
        } else if (args == 2) {
            put32(index.at(n->rhs.get()));
            put32(index.at(n->lhs.get()));
        }
        const uint32_t id = (uint32_t)index.size();
        index.emplace(n.id(), id);
    }
    out.push_back(0xFF);
    out.push_back(0xFF);
    return out;
}

}  // namespace front
}  // namespace mpr
