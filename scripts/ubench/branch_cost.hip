// What does a TAKEN scalar branch cost a wavefront on gfx950?  (The float pass's generated code calls a routine for every
// exp / sqrt / log / division: s_swappc_b64 there, s_setpc_b64 back, and the routines' fast paths jump over their slow paths.)
// Per wave: cycles for 4096 x { 8 dependent v_add_f32 } with, after every group of 8, (a) nothing, (b) s_nop 0, (c) a taken
// s_branch to the next instruction, (d) s_getpc + s_setpc to the next instruction (an indirect jump: what a call / return is).
// Run with 1 and with 6 waves per SIMD (does the rest of the SIMD hide it?).
// Build: hipcc --offload-arch=gfx950 -O3 branch_cost.hip -o branch_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define ADD8 "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
template <int MODE>
__global__ void k(int iters, long long* out, float* sink)
{
    float a = threadIdx.x; const float x = 1e-3f;
    __syncthreads();
    const long long t0 = (long long)__builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) asm volatile(REP16(ADD8) : "+v"(a) : "v"(x));
        if (MODE == 1) asm volatile(REP16(ADD8 "s_nop 0\n") : "+v"(a) : "v"(x));
        if (MODE == 2) asm volatile(REP16(ADD8 "s_branch 0\n") : "+v"(a) : "v"(x));                 /* offset 0: the next instruction */
        if (MODE == 3) asm volatile(REP16(ADD8 "s_getpc_b64 s[40:41]\n s_add_u32 s40, s40, 12\n s_addc_u32 s41, s41, 0\n s_setpc_b64 s[40:41]\n") : "+v"(a) : "v"(x) : "s40", "s41", "scc");
        if (MODE == 4) asm volatile(REP16(ADD8 "s_getpc_b64 s[40:41]\n s_add_u32 s40, s40, 12\n s_addc_u32 s41, s41, 0\n s_nop 0\n") : "+v"(a) : "v"(x) : "s40", "s41", "scc");
    }
    const long long t1 = (long long)__builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    if (a == 12345.f) sink[0] = a;
}
int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    long long* d; float* sink; CK(hipMalloc(&d, 8 * cus * 64)); CK(hipMalloc(&sink, 64));
    const int iters = 256;                       // x 16 groups = 4096 groups of 8 adds
    const char* names[5] = {"8 dependent v_add", "+ s_nop 0", "+ taken s_branch", "+ s_getpc/add/addc/s_setpc (taken)", "+ s_getpc/add/addc/s_nop (not a jump)"};
    for (int wps : {1, 6}) {
        printf("waves/SIMD %d\n", wps);
        double base = 0;
        for (int m = 0; m < 5; ++m) {
            const dim3 g(cus), b(64 * 4 * wps);
            for (int rep = 0; rep < 2; ++rep) {
                if (m == 0) hipLaunchKernelGGL(k<0>, g, b, 0, 0, iters, d, sink);
                if (m == 1) hipLaunchKernelGGL(k<1>, g, b, 0, 0, iters, d, sink);
                if (m == 2) hipLaunchKernelGGL(k<2>, g, b, 0, 0, iters, d, sink);
                if (m == 3) hipLaunchKernelGGL(k<3>, g, b, 0, 0, iters, d, sink);
                if (m == 4) hipLaunchKernelGGL(k<4>, g, b, 0, 0, iters, d, sink);
                CK(hipDeviceSynchronize());
            }
            std::vector<long long> h((size_t)cus * 4 * wps);
            CK(hipMemcpy(h.data(), d, 8 * h.size(), hipMemcpyDeviceToHost));
            double s = 0; for (long long v : h) s += (double)v; s /= h.size();
            const double per_group = s / (iters * 16.0);
            if (m == 0) base = per_group;
            printf("  %-40s %8.1f shader clocks per group of 8 adds per wave  (+%.1f)\n", names[m], per_group, per_group - base);
        }
    }
    return 0;
}
