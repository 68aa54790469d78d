// Written against include/mpr.hpp in the access pattern of the reference's
// benchmark/tape_shortening.cpp:56-118: the length of every surviving tile's shortened tape, read by
// walking ctx.tape_data from tile.tape + 1 and following JUMP clauses.  Prints a summary.
#include <cmath>
#include <cstdio>

#include "mpr.hpp"
#include "mpr_clause.h"

static unsigned walk(const mpr::Context& ctx, int32_t head)
{
    unsigned len = 0;
    for (auto j = head + 1; mpr_cl_op(ctx.tape_data[j]); ++j) {
        const auto d = ctx.tape_data[j];
        if (mpr_cl_op(d) == MPR_OP_JUMP) {
            j += mpr_cl_jump(d);
        } else {
            len++;
        }
    }
    return len;
}

int main(int argc, char** argv)
{
    const unsigned size = argc > 2 ? std::atoi(argv[2]) : 256;
    auto X = libfive::Tree::X(), Y = libfive::Tree::Y(), Z = libfive::Tree::Z();
    auto t = argc > 1 ? libfive::Tree::load(argv[1])            // default: benchmark/render_2d_table.cpp:44-45
                      : min(sqrt((X + 0.5) * (X + 0.5) + Y * Y + Z * Z) - 0.25,
                            sqrt((X - 0.5) * (X - 0.5) + Y * Y + Z * Z) - 0.25);
    auto tape = mpr::Tape(t);
    auto ctx = mpr::Context(size);
    ctx.render2D(tape, mpr::Matrix3f::Identity());
    std::printf("Initial clauses: %d\n", tape.length);

    unsigned long long sum64 = 0, n64 = 0, sum8 = 0, n8 = 0;
    for (unsigned i = 0; i < std::pow(size / 64, 2); ++i) {
        const auto tile = ctx.stages[0].tiles[i];
        if (tile.position == -1) {
            continue;
        }
        sum64 += walk(ctx, tile.tape);
        n64++;
    }
    for (unsigned i = 0; i < ctx.stages[2].tile_array_size; ++i) {
        const auto tile = ctx.stages[2].tiles[i];
        if (tile.position == -1) {
            continue;
        }
        sum8 += walk(ctx, tile.tape);
        n8++;
    }
    if (*ctx.tape_index >= (long long)MPR_NUM_SUBTAPES_BIG * MPR_SUBTAPE_CHUNK) {      // render_3d_heatmap.cpp:64-67
        std::fprintf(stderr, "Tape overflowed and wasn't pruned\n");
        return 1;
    }
    std::printf("64px tiles %llu mean length %.1f; 8px tiles %llu mean length %.1f\n", n64, n64 ? (double)sum64 / n64 : 0.0, n8,
                n8 ? (double)sum8 / n8 : 0.0);
    return 0;
}
