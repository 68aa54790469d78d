/*
 * render_heatmap.cpp — benchmark/render_2d_heatmap.cpp and render_3d_heatmap.cpp of the reference
 * written against include/mpr.hpp: one heatmap frame (Context::render2D_heatmap /
 * render3D_heatmap, inc/context.hpp:51-58), saved with the reference's colour packing — the pixel
 * is 0xFF000000 | unsigned(heat * 100000), i.e. low byte in red (render_3d_heatmap.cpp:72-79) —
 * as a binary PPM, plus a one-line summary.
 *
 *   build:  hipcc -O2 -std=c++17 -Iinclude benchmark/render_heatmap.cpp -Lmpr_amd -lmpr_amd \
 *                 -Wl,-rpath,$PWD/mpr_amd -o render_heatmap
 *   run:    ./render_heatmap 2|3 fixtures/models/prospero.frep [size]
 */
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "mpr.hpp"

int main(int argc, char** argv)
{
    if (argc < 3 || (argv[1][0] != '2' && argv[1][0] != '3')) {
        std::cerr << "usage: render_heatmap 2|3 model.frep [size]\n";
        return 1;
    }
    const int dim = argv[1][0] - '0';
    const int size = argc > 3 ? std::atoi(argv[3]) : 512;       /* the reference's default resolution */
    const libfive::Tree t = libfive::Tree::load(argv[2]);
    auto tape = mpr::Tape(t);
    auto ctx = mpr::Context(size);
    std::vector<float> heat;
    if (dim == 3) {
        mpr::Matrix4f T = mpr::Matrix4f::Identity();
        T(3, 2) = 0.3f;
        heat = ctx.render3D_heatmap(tape, T);
    } else {
        heat = ctx.render2D_heatmap(tape, mpr::Matrix3f::Identity(), 0.0f);
    }
    if (*ctx.tape_index >= (int64_t)MPR_NUM_SUBTAPES_BIG * MPR_SUBTAPE_CHUNK) {
        std::cerr << "Tape overflowed and wasn't pruned\n";
        return 1;
    }

    double sum = 0.0;
    float peak = 0.0f;
    const std::string name = "out_heatmap_" + std::to_string(dim) + "d.ppm";
    std::ofstream f(name, std::ios::binary);
    f << "P6\n" << size << " " << size << "\n255\n";
    for (int y = size - 1; y >= 0; --y) {
        for (int x = 0; x < size; ++x) {
            const float v = heat[(size_t)x + (size_t)y * size];
            sum += v;
            if (v > peak) peak = v;
            const unsigned h = (unsigned)(v * 100000.0f);
            if (h > 0xFFFFFF) {
                std::cerr << "toooo big" << h << "\n";
                return 1;
            }
            f.put((char)(h & 0xFF));
            f.put((char)((h >> 8) & 0xFF));
            f.put((char)((h >> 16) & 0xFF));
        }
    }
    /* mean heat = tape evaluations per pixel, in units of the whole root tape: brute force is 1.0 */
    std::cout << "heatmap " << dim << "D " << size << " mean " << sum / ((double)size * size) << " peak " << peak << " -> "
              << name << "\n";
    return 0;
}
