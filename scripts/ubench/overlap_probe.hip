// Do two kernels with complementary limiters share the chip when they are launched on two streams?  (Design input for
// overlapping the last tile stage — many short waves, each a dependent chain: latency-bound — with the float pass —
// persistent workgroups, VALU-bound.)  A = `na` one-wave workgroups each running a dependent v_fma chain with 122 VGPRs
// reserved (4 waves per SIMD, like k_eval_tiles<3, true, 24>); B = persistent 256-thread workgroups running 8 independent
// chains with 80 VGPRs (6 per SIMD, like k_eval_voxels_jit_groups<3, 24>) until a work counter runs out.
// Reports A alone, B alone, A then B on one stream, A and B on two streams, for several sizes of B's grid.
// Build: hipcc --offload-arch=gfx950 -O3 overlap_probe.hip -o overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

__global__ void __launch_bounds__(64, 4) k_latency(int iters, float* sink)
{
    float a = threadIdx.x;
    const float x = 0.999f, y = 1e-3f;
    // keep the register footprint of the tile stage: 122 VGPRs
    asm volatile("v_mov_b32 v120, 0" ::: "v120");
    for (int i = 0; i < iters; ++i) asm volatile(REP16("v_fma_f32 %0, %0, %1, %2\n s_nop 4\n") : "+v"(a) : "v"(x), "v"(y));
    if (a == 12345.f) sink[0] = a;
}
__global__ void __launch_bounds__(256, 6) k_issue(int* counter, int items, int iters, float* sink)
{
    float a = threadIdx.x, b = 1, c = 2, d = 3, e = 4, f = 5, g = 6, h = 7;
    const float x = 0.999f, y = 1e-3f;
    asm volatile("v_mov_b32 v78, 0" ::: "v78");
    __shared__ int item;
    for (;;) {
        if (threadIdx.x == 0) item = atomicAdd(counter, 1);
        __syncthreads();
        const int it = item;
        __syncthreads();
        if (it >= items) break;
        for (int i = 0; i < iters; ++i)
            asm volatile(REP4("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y));
    }
    if (a == 12345.f) sink[0] = a + b + c + d + e + f + g + h;
}

int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    float* sink; int* counter;
    CK(hipMalloc(&sink, 64)); CK(hipMalloc(&counter, 64));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1, ea; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&ea));
    const int na = 26000, ia = 600;             // A: 26 k waves of ~10 k dependent instructions' latency
    const int items = 26000, ib = 60;           // B: 26 k items of 4 waves x 1920 VALU instructions
    auto time_ms = [&](auto&& f) {
        f(); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, s1); f(); (void)hipEventRecord(e1, s1); (void)hipStreamSynchronize(s2); (void)hipEventSynchronize(e1);
        (void)hipDeviceSynchronize();
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
    };
    auto A = [&](hipStream_t s) { hipLaunchKernelGGL(k_latency, dim3(na), dim3(64), 0, s, ia, sink); };
    auto B = [&](hipStream_t s, int wg_per_cu) {
        (void)hipMemsetAsync(counter, 0, 4, s);
        hipLaunchKernelGGL(k_issue, dim3(cus * wg_per_cu), dim3(256), 0, s, counter, items, ib, sink);
    };
    printf("device %s, %d CUs\n", p.gcnArchName, cus);
    const float ta = time_ms([&] { A(s1); });
    printf("A alone (latency-bound, %d waves):                    %.3f ms\n", na, ta);
    for (int wg : {6, 5, 4, 3}) {
        const float tb = time_ms([&] { B(s1, wg); });
        const float tser = time_ms([&] { A(s1); B(s1, wg); });
        // two streams: B on s2 starts when A starts (event), everything is joined back into s1
        const float tpar = time_ms([&] {
            (void)hipEventRecord(ea, s1);
            (void)hipStreamWaitEvent(s2, ea, 0);
            A(s1);
            B(s2, wg);
            (void)hipEventRecord(ea, s2);
            (void)hipStreamWaitEvent(s1, ea, 0);
        });
        // two streams, B's launch first
        const float tpar2 = time_ms([&] {
            (void)hipEventRecord(ea, s1);
            (void)hipStreamWaitEvent(s2, ea, 0);
            B(s2, wg);
            A(s1);
            (void)hipEventRecord(ea, s2);
            (void)hipStreamWaitEvent(s1, ea, 0);
        });
        printf("B with %d workgroups/CU: alone %.3f ms | A then B %.3f ms | A || B %.3f ms | B || A (B launched first) %.3f ms\n", wg, tb, tser, tpar, tpar2);
    }
    return 0;
}
