/*
 * render_table.cpp — the reference's table benchmarks written against include/mpr.hpp.
 *
 * Same protocol and output as benchmark/render_2d_table.cpp:27-66 and
 * benchmark/render_3d_table.cpp:27-76 of the reference: per size one line
 * "size mean_ms stdev_ms", 20 warm-up + 100 timed blocking calls (benchmark/stats.cpp:19-47),
 * 2-D view = identity, 3-D view = identity with T(3,2) = 0.3, 3-D stops once a frame takes
 * more than 750 ms.  Images are written as binary PGM instead of PNG (no libpng here).
 *
 *   build:  hipcc -O2 -std=c++17 -Iinclude benchmark/render_table.cpp -Lmpr_amd -lmpr_amd \
 *                 -Wl,-rpath,$PWD/mpr_amd -o render_table
 *   run:    ./render_table 2 fixtures/models/prospero.frep
 *           ./render_table 3 fixtures/models/bear.frep
 */
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <iostream>
#include <vector>

#include "mpr.hpp"

static double get_stats(const std::function<void()>& f, int warmup = 20, int count = 100)
{
    for (int i = 0; i < warmup; ++i) f();
    std::vector<double> ms;
    for (int i = 0; i < count; ++i) {
        const auto a = std::chrono::steady_clock::now();
        f();
        const auto b = std::chrono::steady_clock::now();
        ms.push_back(std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count() / 1e6);
    }
    double mean = 0;
    for (double v : ms) mean += v;
    mean /= ms.size();
    double sd = 0;
    for (double v : ms) sd += (v - mean) * (v - mean);
    sd = std::sqrt(sd / (ms.size() - 1));
    std::cout << mean << " " << sd << "\n";
    return mean;
}

static void save_pgm(const std::string& path, const std::vector<int32_t>& img, int size, int maxv)
{
    std::ofstream f(path, std::ios::binary);
    f << "P5\n" << size << " " << size << "\n255\n";
    for (int y = size - 1; y >= 0; --y)
        for (int x = 0; x < size; ++x)
            f.put((char)(maxv ? std::min(255, img[x + y * size] * 255 / maxv) : 0));
}

/* the normals image: the three packed components as RGB (libfive::Heightmap::saveNormalPNG's picture, as a PPM) */
static void save_ppm_normals(const std::string& path, const std::vector<uint32_t>& n, int size)
{
    std::ofstream f(path, std::ios::binary);
    f << "P6\n" << size << " " << size << "\n255\n";
    for (int y = size - 1; y >= 0; --y)
        for (int x = 0; x < size; ++x) {
            const uint32_t v = n[(size_t)x + (size_t)y * size];
            f.put((char)(v & 0xFF));
            f.put((char)((v >> 8) & 0xFF));
            f.put((char)((v >> 16) & 0xFF));
        }
}

int main(int argc, char** argv)
{
    const int dim = argc > 1 ? std::atoi(argv[1]) : 2;
    libfive::Tree t = libfive::Tree::X();
    if (argc > 2) {
        t = libfive::Tree::load(argv[2]);
    } else {
        /* default model of both reference programs */
        auto X = libfive::Tree::X(), Y = libfive::Tree::Y(), Z = libfive::Tree::Z();
        t = min(sqrt((X + 0.5) * (X + 0.5) + Y * Y + Z * Z) - 0.25,
                sqrt((X - 0.5) * (X - 0.5) + Y * Y + Z * Z) - 0.25);
    }
    if (dim == 2) {
        for (int size : {256, 512, 1024, 2048, 3072, 4096}) {
            if (size % 64) continue;
            auto tape = mpr::Tape(t);
            auto c = mpr::Context(size);
            std::cout << size << " ";
            get_stats([&]() { c.render2D(tape, mpr::Matrix3f::Identity()); });
            save_pgm("out_gpu_" + std::to_string(size) + ".pgm", c.stages[3].filled, size, 1);
        }
    } else {
        mpr::Matrix4f T = mpr::Matrix4f::Identity();
        T(3, 2) = 0.3f;
        for (int size : {256, 512, 1024, 1536, 2048}) {
            auto tape = mpr::Tape(t);
            auto c = mpr::Context(size);
            std::cout << size << " ";
            const double mean = get_stats([&]() { c.render3D(tape, T); });
            save_pgm("out_gpu_depth_" + std::to_string(size) + ".pgm", c.stages[3].filled, size, size);
            save_ppm_normals("out_gpu_norm_" + std::to_string(size) + ".ppm", c.normals, size);     /* (render_3d_table.cpp:64-69: saveNormalPNG) */
            if (mean > 750) break;
        }
    }
    return 0;
}
