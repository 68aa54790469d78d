// Written against include/mpr.hpp in the access pattern of the reference's benchmark/circle.cpp:42-103
// (which 64^2 tiles were decided at the first stage, which 8^2 tiles at the second, and what the
// final image says about the rest).  Prints counts instead of writing PNGs.  Compiled by
// tests/test_host_api.py (syntax only without a GPU; run by the gpu test).
#include <cmath>
#include <cstdio>
#include <vector>

#include "mpr.hpp"

int main(int argc, char** argv)
{
    auto X = libfive::Tree::X();
    auto Y = libfive::Tree::Y();
    auto t = sqrt((X + 1) * (X + 1) + (Y + 1) * (Y + 1)) - 1.8;
    const unsigned size = argc > 1 ? std::atoi(argv[1]) : 128;

    auto tape = mpr::Tape(t);
    auto ctx = mpr::Context(size);
    ctx.render2D(tape, mpr::Matrix3f::Identity());

    unsigned decided64 = 0;
    for (unsigned i = 0; i < std::pow(size / 64, 2); ++i) {
        const auto tile = ctx.stages[0].tiles[i];
        if (tile.position != -1) {
            continue;
        }
        decided64++;
    }

    // every surviving 64^2 tile was given a compacted id (`next`) and its 64 children sit at [64 * id, 64 * id + 64) of the
    // next stage's list: one table from id to the parent's position, built once
    std::vector<int> parent_of(ctx.stages[2].tile_array_size / 64 + 1, -1);
    for (unsigned j = 0; j < std::pow(size / 64, 2); ++j) {
        const auto top = ctx.stages[0].tiles[j];
        if (top.next >= 0 && (size_t)top.next < parent_of.size()) {
            parent_of[top.next] = top.position;
        }
    }
    unsigned decided8 = 0, filled8 = 0;
    for (unsigned i = 0; i < ctx.stages[2].tile_array_size; ++i) {
        if (ctx.stages[2].tiles[i].position != -1) {
            continue;
        }
        const unsigned parent = parent_of[i / 64], child = i % 64, per_side = size / 64;
        const unsigned x = (parent % per_side) * 8 + child % 8, y = (parent / per_side) * 8 + child / 8;
        decided8++;
        filled8 += ctx.stages[3].filled[x * 8 + y * 8 * size] != 0;
    }

    unsigned inside = 0;
    for (unsigned i = 0; i < size; ++i) {
        for (unsigned j = 0; j < size; ++j) {
            inside += ctx.stages[3].filled[i + j * size] != 0;
        }
    }
    std::printf("size %u decided64 %u decided8 %u filled8 %u inside %u tape_index %d\n", size, decided64, decided8, filled8,
                inside, (int)*ctx.tape_index);
    return 0;
}
