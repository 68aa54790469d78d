/*
 * mpr.hpp — header-only C++ façade over the C ABI (mpr_amd.h) that restores the API the
 * reference's callers are written against:
 *
 *     auto tape = mpr::Tape(tree);                    inc/tape.hpp:24-30
 *     auto ctx  = mpr::Context(size);                 inc/context.hpp:38-39
 *     ctx.render2D(tape, Matrix3f::Identity());       inc/context.hpp:41-42
 *     ctx.render3D(tape, T);                          inc/context.hpp:40
 *     ctx.stages[3].filled[i], ctx.normals[i], ctx.stages[k].tiles[i],
 *     ctx.stages[k].tile_array_size, ctx.tape_data[j], *ctx.tape_index
 *
 * The reference exposes managed-memory pointers (Ptr<T[]>, inc/util.hpp:44-50) that callers index
 * on the host after a render (benchmark/circle.cpp:43-71, tape_shortening.cpp:57-98,
 * render_3d_heatmap.cpp:64); here the same members — stages[k].filled / .tiles / .tile_array_size,
 * normals, tape_data, tape_index — are host mirrors with the same spelling at the call site
 * (operator[], get(), operator* for tape_index, conversion to size_t for tile_array_size), fetched
 * from the device on first access after a render, so a timing loop that only renders pays for
 * nothing else.  the programs under tests/facade/ are this repository's own programs written in the access
 * patterns of those three reference files; tests/test_host_api.py compiles them.  Eigen is not required: Matrix3f /
 * Matrix4f below are minimal column-major matrices with Eigen's (row, col) indexing; an Eigen
 * matrix's .data() can be passed to the *_raw overloads directly.
 *
 * libfive is not part of this repository; libfive::Tree below is the small expression front
 * end of libmpr_amd (operators, sqrt/min/max/..., .frep archives) under libfive's names.  What a
 * reference main still has to change: `libfive::Archive::deserialize(in).shapes.front().tree`
 * becomes `libfive::Tree::load(path)`, Eigen::Matrix{3,4}f becomes mpr::Matrix{3,4}f (or pass
 * Eigen's .data()), and libfive::Heightmap / savePNG are not provided (benchmark/render_table.cpp
 * here writes PGM).
 */
#pragma once
#include <cmath>
#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "mpr_amd.h"

namespace mpr {

inline void check(int rc)
{
    if (rc != MPR_OK) throw std::runtime_error(std::string("mpr: ") + mpr_last_error());
}

template <int N>
struct MatrixNf {
    float m[N * N];
    static MatrixNf Identity()
    {
        MatrixNf r;
        for (int i = 0; i < N * N; ++i) r.m[i] = 0.0f;
        for (int i = 0; i < N; ++i) r.m[i + i * N] = 1.0f;
        return r;
    }
    float& operator()(int row, int col) { return m[row + col * N]; }
    float operator()(int row, int col) const { return m[row + col * N]; }
    const float* data() const { return m; }
};
using Matrix3f = MatrixNf<3>;
using Matrix4f = MatrixNf<4>;

}  // namespace mpr

namespace libfive {

/* stand-in for libfive::Tree (see tree.hpp in the library sources) */
class Tree {
public:
    Tree() = default;
    Tree(float v) { mpr_tree* t = nullptr; mpr::check(mpr_tree_const(v, &t)); reset(t); }
    Tree(double v) : Tree((float)v) {}
    Tree(int v) : Tree((float)v) {}
    static Tree X() { mpr_tree* t = nullptr; mpr::check(mpr_tree_x(&t)); return Tree(t); }
    static Tree Y() { mpr_tree* t = nullptr; mpr::check(mpr_tree_y(&t)); return Tree(t); }
    static Tree Z() { mpr_tree* t = nullptr; mpr::check(mpr_tree_z(&t)); return Tree(t); }
    static Tree unary(int op, const Tree& a)
    {
        mpr_tree* t = nullptr;
        mpr::check(mpr_tree_unary(op, a.get(), &t));
        return Tree(t);
    }
    static Tree binary(int op, const Tree& a, const Tree& b)
    {
        mpr_tree* t = nullptr;
        mpr::check(mpr_tree_binary(op, a.get(), b.get(), &t));
        return Tree(t);
    }
    Tree remap(const Tree& x, const Tree& y, const Tree& z) const
    {
        mpr_tree* t = nullptr;
        mpr::check(mpr_tree_remap(get(), x.get(), y.get(), z.get(), &t));
        return Tree(t);
    }
    /* Archive::deserialize(in).shapes.front().tree */
    static Tree load(const std::string& frep_path)
    {
        mpr_tree* t = nullptr;
        mpr::check(mpr_tree_from_frep_file(frep_path.c_str(), &t));
        return Tree(t);
    }
    const mpr_tree* get() const { return p.get(); }

private:
    explicit Tree(mpr_tree* t) { reset(t); }
    void reset(mpr_tree* t) { p = std::shared_ptr<mpr_tree>(t, mpr_tree_free); }
    std::shared_ptr<mpr_tree> p;
};
inline Tree operator+(const Tree& a, const Tree& b) { return Tree::binary(MPR_T_ADD, a, b); }
inline Tree operator-(const Tree& a, const Tree& b) { return Tree::binary(MPR_T_SUB, a, b); }
inline Tree operator*(const Tree& a, const Tree& b) { return Tree::binary(MPR_T_MUL, a, b); }
inline Tree operator/(const Tree& a, const Tree& b) { return Tree::binary(MPR_T_DIV, a, b); }
inline Tree operator-(const Tree& a) { return Tree::unary(MPR_T_NEG, a); }
inline Tree min(const Tree& a, const Tree& b) { return Tree::binary(MPR_T_MIN, a, b); }
inline Tree max(const Tree& a, const Tree& b) { return Tree::binary(MPR_T_MAX, a, b); }
inline Tree sqrt(const Tree& a) { return Tree::unary(MPR_T_SQRT, a); }
inline Tree square(const Tree& a) { return Tree::unary(MPR_T_SQUARE, a); }
inline Tree abs(const Tree& a) { return Tree::unary(MPR_T_ABS, a); }
inline Tree sin(const Tree& a) { return Tree::unary(MPR_T_SIN, a); }
inline Tree cos(const Tree& a) { return Tree::unary(MPR_T_COS, a); }
inline Tree asin(const Tree& a) { return Tree::unary(MPR_T_ASIN, a); }
inline Tree acos(const Tree& a) { return Tree::unary(MPR_T_ACOS, a); }
inline Tree atan(const Tree& a) { return Tree::unary(MPR_T_ATAN, a); }
inline Tree exp(const Tree& a) { return Tree::unary(MPR_T_EXP, a); }
inline Tree log(const Tree& a) { return Tree::unary(MPR_T_LOG, a); }

}  // namespace libfive

namespace mpr {

/* Host mirror of a device array that the reference exposes as a managed-memory Ptr<T[]>
 * (inc/util.hpp): indexable, .get()-able, fetched from the device on first use after a render. */
template <typename T>
struct Mirror {
    const T& operator[](size_t i) const { return fetch()[i]; }
    const T* get() const { return fetch().data(); }
    const T* data() const { return fetch().data(); }
    size_t size() const { return fetch().size(); }
    typename std::vector<T>::const_iterator begin() const { return fetch().begin(); }
    typename std::vector<T>::const_iterator end() const { return fetch().end(); }
    operator const std::vector<T>&() const { return fetch(); }
    void invalidate() { valid = false; }
    std::function<void(std::vector<T>&)> loader;

private:
    const std::vector<T>& fetch() const
    {
        if (!valid && loader) {
            loader(cache);
            valid = true;
        }
        return cache;
    }
    mutable std::vector<T> cache;
    mutable bool valid = false;
};

struct Tape {
    /* Unlike the reference's constructor (src/tape.cpp:79-81, :106, :195: a warning on stderr, then a tape that evaluates
     * something else), this one THROWS std::runtime_error for an expression that needs more than 254 live values at once
     * or names an opcode the evaluators lack: a main written against the reference may want a try block. */
    explicit Tape(const libfive::Tree& tree)
    {
        mpr_tape* t = nullptr;
        check(mpr_tape_from_tree(tree.get(), &t));
        handle = std::shared_ptr<mpr_tape>(t, mpr_tape_free);
        data = mpr_tape_data(t);
        length = mpr_tape_length(t);
    }
    const uint64_t* data = nullptr;   /* host copy; the device copy lives in the Context's pool */
    int32_t length = 0;
    std::shared_ptr<mpr_tape> handle;
};

using TileNode = mpr_tile_node;

/* a single value behind a Ptr<T> in the reference (Context::tape_index) or a plain member that is
 * only known after the frame (Tiles::tile_array_size): fetched on first use after a render */
template <typename T>
struct LazyValue {
    T operator*() const { return fetch(); }
    operator T() const { return fetch(); }
    void invalidate() { valid = false; }
    std::function<T()> loader;

private:
    T fetch() const
    {
        if (!valid && loader) {
            cache = loader();
            valid = true;
        }
        return cache;
    }
    mutable T cache = T();
    mutable bool valid = false;
};

struct Tiles {                        /* inc/context.hpp:29-36 */
    Mirror<int32_t> filled;
    Mirror<TileNode> tiles;
    LazyValue<size_t> tile_array_size;
};

struct Context {
    explicit Context(int32_t image_size_px, int32_t device = 0) : image_size_px(image_size_px)
    {
        mpr_context* c = nullptr;
        check(mpr_ctx_create(device, image_size_px, &c));
        handle = std::shared_ptr<mpr_context>(c, mpr_ctx_destroy);
        mpr_context* const h = handle.get();          /* the loaders hold the C handle, never `this` */
        for (int i = 0; i < 4; ++i) {
            stages[i].filled.loader = [h, i](std::vector<int32_t>& v) {
                const int32_t S = mpr_ctx_image_size(h);
                const int32_t side = (i == 3) ? S : (i == 2 ? S / 4 : (i == 1 ? S / 16 : S / 64));
                v.resize((size_t)side * side);
                check(mpr_read_filled(h, i, v.data()));
            };
            stages[i].tiles.loader = [h, i](std::vector<TileNode>& v) {
                size_t n = 0;
                check(mpr_read_tiles(h, i, nullptr, 0, &n));
                v.resize(n);
                if (n) check(mpr_read_tiles(h, i, v.data(), n, &n));
            };
            stages[i].tile_array_size.loader = [h, i]() {
                size_t n = 0;
                check(mpr_read_tiles(h, i, nullptr, 0, &n));
                return n;
            };
        }
        normals.loader = [h](std::vector<uint32_t>& v) {
            const int32_t S = mpr_ctx_image_size(h);
            v.resize((size_t)S * S);
            check(mpr_read_normals(h, v.data()));
        };
        tape_data.loader = [h](std::vector<uint64_t>& v) {
            int32_t ti = 0;
            check(mpr_read_tape_pool(h, nullptr, 0, &ti));
            v.resize(ti > 0 ? (size_t)ti : 0);
            if (!v.empty()) check(mpr_read_tape_pool(h, v.data(), v.size(), &ti));
        };
        tape_index.loader = [h]() {
            int32_t ti = 0;
            check(mpr_read_tape_pool(h, nullptr, 0, &ti));
            return ti;
        };
    }
    /* one owner per device context, like the reference's unique_ptr members: movable, not copyable */
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    Context(Context&&) = default;
    Context& operator=(Context&&) = default;

    void render2D(const Tape& tape, const Matrix3f& mat, const float z = 0.0f)
    {
        check(mpr_render2d(handle.get(), tape.handle.get(), mat.data(), z));
        refresh(false);
    }
    void render3D(const Tape& tape, const Matrix4f& mat)
    {
        check(mpr_render3d(handle.get(), tape.handle.get(), mat.data()));
        refresh(true);
    }
    void render2D_brute(const Tape& tape, const Matrix3f& mat, const float z = 0.0f)
    {
        check(mpr_render2d_brute(handle.get(), tape.handle.get(), mat.data(), z));
        refresh(false);
    }
    /* inc/context.hpp:51-58.  Upstream returns a managed float array; here a host vector. */
    std::vector<float> render2D_heatmap(const Tape& tape, const Matrix3f& mat, const float z = 0.0f)
    {
        std::vector<float> heat((size_t)image_size_px * image_size_px);
        check(mpr_render2d_heatmap(handle.get(), tape.handle.get(), mat.data(), z, heat.data()));
        refresh(false);
        return heat;
    }
    std::vector<float> render3D_heatmap(const Tape& tape, const Matrix4f& mat)
    {
        std::vector<float> heat((size_t)image_size_px * image_size_px);
        check(mpr_render3d_heatmap(handle.get(), tape.handle.get(), mat.data(), heat.data()));
        refresh(true);
        return heat;
    }
    int32_t image_size_px;
    Mirror<uint64_t> tape_data;       /* tape_data[j]: the pool up to *tape_index (benchmark/tape_shortening.cpp:56-72) */
    LazyValue<int32_t> tape_index;    /* *tape_index (benchmark/render_3d_heatmap.cpp:64) */
    Tiles stages[4];
    Mirror<uint32_t> normals;
    std::shared_ptr<mpr_context> handle;

private:
    void refresh(bool)
    {
        for (auto& s : stages) {
            s.filled.invalidate();
            s.tiles.invalidate();
            s.tile_array_size.invalidate();
        }
        normals.invalidate();
        tape_data.invalidate();
        tape_index.invalidate();
    }
};

/* inc/effects.hpp:21-37 — SSAO and shading over the context's last render3D */
struct Effects {
    explicit Effects(int32_t device = 0)
    {
        mpr_effects* e = nullptr;
        check(mpr_effects_create(device, &e));
        handle = std::shared_ptr<mpr_effects>(e, mpr_effects_destroy);
    }
    void drawSSAO(const Context& ctx)
    {
        check(mpr_effects_draw_ssao(handle.get(), ctx.handle.get()));
        refresh(ctx.image_size_px);
    }
    void drawShaded(const Context& ctx)
    {
        check(mpr_effects_draw_shaded(handle.get(), ctx.handle.get()));
        refresh(ctx.image_size_px);
    }
    Mirror<int32_t> image;            /* Effects::image: fetched on first use after a draw */
    std::shared_ptr<mpr_effects> handle;

private:
    void refresh(int32_t size)
    {
        mpr_effects* const h = handle.get();
        image.loader = [h, size](std::vector<int32_t>& v) {
            v.resize((size_t)size * size);
            check(mpr_effects_read_image(h, v.data()));
        };
        image.invalidate();
    }
};

}  // namespace mpr
