/*
 * render_table_multi.cpp — the reference's 3-D table benchmark (benchmark/render_3d_table.cpp:27-76) on N GPUs of one node,
 * in C++ against the C ABI (include/mpr_amd.h): the tile-parallel loop of SURVEY.md 8(e) without Python.
 *
 * One process, one HOST THREAD PER GPU (a frame's driver waits for the survivor counts between its stages, so ranks must not share
 * a thread), one context per device.  The 64 x 64 pixel columns are dealt to the ranks by the first tile stage's own verdict
 * (mpr_column_weights -> mpr_partition_columns: identical on every rank, no communication).  Per frame and rank, all on the
 * rank's stream:  mpr_render3d_part_async -> mpr_pack_planned_async -> [every peer's pack copied over xGMI into this rank's
 * receive buffer: hipMemcpyPeerAsync, ordered behind the peer's pack by an event] -> mpr_unpack_planned_async -> sync.
 * No reduction: every rank ends with the complete heightmap and normals, bit-identical to a single-GPU frame (--verify checks it).
 * (mpr_amd/multigpu.py + bench.py --gpus N is the same loop with one PROCESS per GPU and the gather as one RCCL all-gather.)
 *
 *   build:  hipcc -O2 -std=c++17 -Iinclude benchmark/render_table_multi.cpp -Lmpr_amd -lmpr_amd -Wl,-rpath,$PWD/mpr_amd \
 *                 -pthread -o render_table_multi
 *   run:    ./render_table_multi fixtures/models/bear.frep --gpus 8 [--sizes 1024,2048] [--verify] [--frames 100]
 *           --share-device: all ranks on device 0 (a development check of the loop on a box with one GPU; not a measurement)
 * Output per size: "size gpus mean_ms stdev_ms" (the slowest rank's frame: all ranks start a frame together).
 */
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "mpr_amd.h"

#define CHECK_HIP(x)                                                                                    \
    do {                                                                                                \
        hipError_t e_ = (x);                                                                            \
        if (e_ != hipSuccess) {                                                                         \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                \
            std::exit(2);                                                                               \
        }                                                                                               \
    } while (0)
#define CHECK_MPR(x)                                                                                    \
    do {                                                                                                \
        if ((x) != MPR_OK) {                                                                            \
            std::fprintf(stderr, "%s: %s\n", #x, mpr_last_error());                                     \
            std::exit(2);                                                                               \
        }                                                                                               \
    } while (0)

namespace {
struct Barrier {                       /* (std::barrier is C++20) */
    std::mutex m;
    std::condition_variable cv;
    int n, waiting = 0, phase = 0;
    explicit Barrier(int n_) : n(n_) {}
    void wait()
    {
        std::unique_lock<std::mutex> l(m);
        const int p = phase;
        if (++waiting == n) {
            waiting = 0;
            ++phase;
            cv.notify_all();
        } else {
            cv.wait(l, [&] { return phase != p; });
        }
    }
};

struct Rank {
    int device = 0;
    mpr_context* ctx = nullptr;
    hipStream_t stream = nullptr;
    int* send = nullptr;               /* this rank's pack: capacity columns x (heights + normals) */
    int* recv = nullptr;               /* every rank's pack, rank r at r * per_rank */
    hipEvent_t packed = nullptr;       /* recorded behind this rank's pack on its stream */
};
}  // namespace

int main(int argc, char** argv)
{
    std::string model;
    int gpus = 1, frames = 100, warmup = 20;
    bool share = false, verify = false;
    std::vector<int> sizes = {256, 512, 1024, 1536, 2048};
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--gpus" && i + 1 < argc) gpus = std::atoi(argv[++i]);
        else if (a == "--frames" && i + 1 < argc) frames = std::atoi(argv[++i]);
        else if (a == "--warmup" && i + 1 < argc) warmup = std::atoi(argv[++i]);
        else if (a == "--share-device") share = true;
        else if (a == "--verify") verify = true;
        else if (a == "--sizes" && i + 1 < argc) {
            sizes.clear();
            for (char* tok = std::strtok(argv[++i], ","); tok; tok = std::strtok(nullptr, ",")) sizes.push_back(std::atoi(tok));
        } else model = a;
    }
    if (model.empty() || gpus < 1) {
        std::fprintf(stderr, "usage: render_table_multi <model.frep> --gpus N [--sizes a,b] [--frames n] [--verify] [--share-device]\n");
        return 1;
    }
    int ndev = 0;
    CHECK_HIP(hipGetDeviceCount(&ndev));
    if (!share && gpus > ndev) {
        std::fprintf(stderr, "%d GPUs asked for, %d present (--share-device puts every rank on device 0)\n", gpus, ndev);
        return 1;
    }
    mpr_tree* tree = nullptr;
    CHECK_MPR(mpr_tree_from_frep_file(model.c_str(), &tree));
    mpr_tape* tape = nullptr;
    CHECK_MPR(mpr_tape_from_tree(tree, &tape));
    float T[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0.3f, 0, 0, 0, 1};      /* column-major identity with T(3,2) = 0.3 */
    if (!share)
        for (int a = 0; a < gpus; ++a) {
            CHECK_HIP(hipSetDevice(a));
            for (int b = 0; b < gpus; ++b)
                if (a != b) (void)hipDeviceEnablePeerAccess(b, 0);      /* xGMI: direct peer copies (an error here means: staged) */
        }

    for (int S : sizes) {
        if (S % 64) continue;
        const int cols = (S / 64) * (S / 64);
        std::vector<Rank> ranks((size_t)gpus);
        for (int r = 0; r < gpus; ++r) {
            ranks[(size_t)r].device = share ? 0 : r;
            CHECK_HIP(hipSetDevice(ranks[(size_t)r].device));
            CHECK_MPR(mpr_ctx_create(ranks[(size_t)r].device, S, &ranks[(size_t)r].ctx));
            ranks[(size_t)r].stream = static_cast<hipStream_t>(mpr_ctx_stream(ranks[(size_t)r].ctx));
            CHECK_HIP(hipEventCreateWithFlags(&ranks[(size_t)r].packed, hipEventDisableTiming));
        }
        /* the deal: every rank would compute the same table; one does */
        std::vector<float> w((size_t)cols);
        std::vector<int32_t> owner((size_t)cols, 0);
        CHECK_HIP(hipSetDevice(ranks[0].device));
        CHECK_MPR(mpr_column_weights(ranks[0].ctx, tape, 3, T, 0.0f, w.data()));
        CHECK_MPR(mpr_partition_columns(cols, w.data(), gpus, owner.data()));
        int capacity = 0;
        {
            std::vector<int> cnt((size_t)gpus, 0);
            for (int32_t o : owner) capacity = std::max(capacity, ++cnt[(size_t)o]);
        }
        const size_t per_rank = (size_t)capacity * 4096 * 2;      /* ints: heights, then normals, of `capacity` columns */
        for (int r = 0; r < gpus; ++r) {
            CHECK_HIP(hipSetDevice(ranks[(size_t)r].device));
            CHECK_HIP(hipMalloc((void**)&ranks[(size_t)r].send, per_rank * sizeof(int)));
            CHECK_HIP(hipMalloc((void**)&ranks[(size_t)r].recv, per_rank * sizeof(int) * (size_t)gpus));
            CHECK_HIP(hipMemset(ranks[(size_t)r].send, 0, per_rank * sizeof(int)));
            if (gpus > 1) CHECK_MPR(mpr_gather_plan(ranks[(size_t)r].ctx, owner.data(), r, gpus, capacity, 1));
        }

        Barrier bar(gpus);
        std::vector<std::vector<double>> ms((size_t)gpus);
        auto frame = [&](int r) {
            Rank& me = ranks[(size_t)r];
            if (gpus == 1) {
                CHECK_MPR(mpr_render3d(me.ctx, tape, T));
                return;
            }
            CHECK_MPR(mpr_render3d_part_async(me.ctx, tape, T, owner.data(), r));
            CHECK_MPR(mpr_pack_planned_async(me.ctx, me.send));
            CHECK_HIP(hipEventRecord(me.packed, me.stream));
            bar.wait();                                    /* every rank's pack is enqueued: its event can be waited for */
            for (int p = 0; p < gpus; ++p) {
                if (p == r) continue;
                const Rank& peer = ranks[(size_t)p];
                CHECK_HIP(hipStreamWaitEvent(me.stream, peer.packed, 0));
                CHECK_HIP(hipMemcpyPeerAsync(me.recv + (size_t)p * per_rank, me.device, peer.send, peer.device, per_rank * sizeof(int), me.stream));
            }
            CHECK_MPR(mpr_unpack_planned_async(me.ctx, me.recv));
            CHECK_MPR(mpr_ctx_sync(me.ctx));
            bar.wait();                                    /* nobody re-records `packed` or rewrites `send` while a peer still copies */
        };
        std::vector<std::thread> threads;
        for (int r = 0; r < gpus; ++r)
            threads.emplace_back([&, r] {
                CHECK_HIP(hipSetDevice(ranks[(size_t)r].device));
                for (int i = 0; i < warmup; ++i) frame(r);
                for (int i = 0; i < frames; ++i) {
                    bar.wait();
                    const auto a = std::chrono::steady_clock::now();
                    frame(r);
                    const auto b = std::chrono::steady_clock::now();
                    ms[(size_t)r].push_back(std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count() / 1e6);
                }
            });
        for (auto& t : threads) t.join();
        double mean = 0, sd = 0;
        std::vector<double> worst((size_t)frames, 0.0);
        for (int i = 0; i < frames; ++i) {
            for (int r = 0; r < gpus; ++r) worst[(size_t)i] = std::max(worst[(size_t)i], ms[(size_t)r][(size_t)i]);
            mean += worst[(size_t)i];
        }
        mean /= frames;
        for (double v : worst) sd += (v - mean) * (v - mean);
        sd = frames > 1 ? std::sqrt(sd / (frames - 1)) : 0.0;
        std::printf("%d %d %.4f %.4f\n", S, gpus, mean, sd);
        std::fflush(stdout);

        if (verify) {
            /* every rank's gathered frame against a single-GPU frame */
            mpr_context* one = nullptr;
            CHECK_HIP(hipSetDevice(ranks[0].device));
            CHECK_MPR(mpr_ctx_create(ranks[0].device, S, &one));
            CHECK_MPR(mpr_render3d(one, tape, T));
            std::vector<int32_t> h0((size_t)S * S), h((size_t)S * S);
            std::vector<uint32_t> n0((size_t)S * S), n((size_t)S * S);
            CHECK_MPR(mpr_read_filled(one, 3, h0.data()));
            CHECK_MPR(mpr_read_normals(one, n0.data()));
            for (int r = 0; r < gpus; ++r) {
                CHECK_HIP(hipSetDevice(ranks[(size_t)r].device));
                CHECK_MPR(mpr_read_filled(ranks[(size_t)r].ctx, 3, h.data()));
                CHECK_MPR(mpr_read_normals(ranks[(size_t)r].ctx, n.data()));
                if (std::memcmp(h.data(), h0.data(), h.size() * 4) != 0 || std::memcmp(n.data(), n0.data(), n.size() * 4) != 0) {
                    std::fprintf(stderr, "size %d: rank %d of %d holds a frame that differs from the single-GPU frame\n", S, r, gpus);
                    return 3;
                }
            }
            mpr_ctx_destroy(one);
            std::printf("  verified: every rank's gathered frame equals the single-GPU frame\n");
        }
        for (int r = 0; r < gpus; ++r) {
            CHECK_HIP(hipSetDevice(ranks[(size_t)r].device));
            mpr_ctx_destroy(ranks[(size_t)r].ctx);
            CHECK_HIP(hipFree(ranks[(size_t)r].send));
            CHECK_HIP(hipFree(ranks[(size_t)r].recv));
            CHECK_HIP(hipEventDestroy(ranks[(size_t)r].packed));
        }
        if (mean > 750) break;
    }
    mpr_tape_free(tape);
    mpr_tree_free(tree);
    return 0;
}
