/*
 * render_effects.cpp — the reference's benchmark/render_effects.cpp written against
 * include/mpr.hpp: render a model in 3-D, then time Effects::drawSSAO and Effects::drawShaded
 * (20 warm-up + 100 timed calls each, benchmark/stats.cpp:19-47) and save the shaded image.
 *
 *   build:  hipcc -O2 -std=c++17 -Iinclude benchmark/render_effects.cpp -Lmpr_amd -lmpr_amd \
 *                 -Wl,-rpath,$PWD/mpr_amd -o render_effects
 *   run:    ./render_effects fixtures/models/bear.frep [size]
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

static void stats(const char* what, const std::function<void()>& f, int warmup = 20, int count = 100)
{
    for (int i = 0; i < warmup; ++i) f();
    std::vector<double> ms;
    for (int i = 0; i < count; ++i) {
        const auto a = std::chrono::steady_clock::now();
        f();
        const auto b = std::chrono::steady_clock::now();
        ms.push_back(std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count() / 1e6);
    }
    double mean = 0, sd = 0;
    for (double v : ms) mean += v;
    mean /= ms.size();
    for (double v : ms) sd += (v - mean) * (v - mean);
    std::cout << what << " " << mean << " " << std::sqrt(sd / (ms.size() - 1)) << "\n";
}

int main(int argc, char** argv)
{
    if (argc < 2) {
        std::cerr << "usage: render_effects model.frep [size]\n";
        return 1;
    }
    const int size = argc > 2 ? std::atoi(argv[2]) : 1024;
    const libfive::Tree t = libfive::Tree::load(argv[1]);
    mpr::Matrix4f T = mpr::Matrix4f::Identity();
    T(3, 2) = 0.3f;
    auto tape = mpr::Tape(t);
    auto ctx = mpr::Context(size);
    auto fx = mpr::Effects();
    stats("render3D", [&]() { ctx.render3D(tape, T); });
    stats("drawSSAO", [&]() { fx.drawSSAO(ctx); });
    stats("drawShaded", [&]() { fx.drawShaded(ctx); });

    std::ofstream f("out_shaded_" + std::to_string(size) + ".pgm", std::ios::binary);
    f << "P5\n" << size << " " << size << "\n255\n";
    for (int y = size - 1; y >= 0; --y)
        for (int x = 0; x < size; ++x) f.put((char)(fx.image[(size_t)x + (size_t)y * size] & 0xFF));
    return 0;
}
