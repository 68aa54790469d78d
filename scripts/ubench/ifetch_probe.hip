// Is straight-line VALU code bound by instruction FETCH rather than issue on gfx950?  (The float pass's generated code is a few
// thousand instructions per walk, most of them 8 bytes long — literals, VOP3 —, each wave fetching its own stream; its SQ counters
// show 0.51 instruction fetches per CU-cycle.)  Same arithmetic, three encodings, bodies of 2048 independent-ish instructions
// (8 chains) executed by 8 waves per SIMD:  v_add_f32 vD, vA, vB (VOP2, 4 bytes);  v_add_f32 vD, 0x3f8ccccd, vB (VOP2 + literal,
// 8 bytes);  v_add_f32_e64 (VOP3, 8 bytes).  Reports wave-instructions per clock per CU and bytes of code per clock per CU.
// Build: hipcc --offload-arch=gfx950 -O3 ifetch_probe.hip -o ifetch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
#define R256(x) R16(R16(x))
#define B8_VOP2 "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
#define B8_LIT  "v_add_f32 %0, 0x3f8ccccd, %0\n v_add_f32 %1, 0x3f8ccccd, %1\n v_add_f32 %2, 0x3f8ccccd, %2\n v_add_f32 %3, 0x3f8ccccd, %3\n v_add_f32 %4, 0x3f8ccccd, %4\n v_add_f32 %5, 0x3f8ccccd, %5\n v_add_f32 %6, 0x3f8ccccd, %6\n v_add_f32 %7, 0x3f8ccccd, %7\n"
#define B8_VOP3 "v_add_f32_e64 %0, %0, %8\n v_add_f32_e64 %1, %1, %8\n v_add_f32_e64 %2, %2, %8\n v_add_f32_e64 %3, %3, %8\n v_add_f32_e64 %4, %4, %8\n v_add_f32_e64 %5, %5, %8\n v_add_f32_e64 %6, %6, %8\n v_add_f32_e64 %7, %7, %8\n"
#define KERNEL(name, body)                                                                                      \
__global__ void __launch_bounds__(1024) name(int iters, float* sink) {                                          \
    float a = threadIdx.x, b = 1, c = 2, d = 3, e = 4, f = 5, g = 6, h = 7; const float x = 1e-3f;              \
    for (int i = 0; i < iters; ++i)                                                                             \
        asm volatile(R256(body) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x)); \
    if (a == 12345.f) sink[0] = a + b + c + d + e + f + g + h;                                                  \
}
KERNEL(k_vop2, B8_VOP2)
KERNEL(k_lit, B8_LIT)
KERNEL(k_vop3, B8_VOP3)
int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    float* sink; CK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    typedef void (*kfn)(int, float*);
    struct T { const char* name; kfn fn; int bytes; } tests[] = {{"VOP2, 4 bytes", k_vop2, 4}, {"VOP2 + literal, 8 bytes", k_lit, 8}, {"VOP3, 8 bytes", k_vop3, 8}};
    const int iters = 200;                         // x 2048 instructions
    for (int wps : {2, 4, 8}) {
        for (auto& t : tests) {
            const dim3 g(cus * (wps * 4 / 16 > 0 ? wps * 4 / 16 : 1)), b(64 * (wps * 4 >= 16 ? 16 : wps * 4));
            hipLaunchKernelGGL(t.fn, g, b, 0, 0, 4, sink); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(t.fn, g, b, 0, 0, iters, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double instr_per_cu = 2048.0 * iters * wps * 4;
            const double per_clk = instr_per_cu / (ms * 1e-3 * p.clockRate * 1e3);
            printf("waves/SIMD %d  %-26s %7.3f ms  %.3f wave-instr/clk/CU (nominal clock)  = %.1f bytes of code per clk per CU (each wave fetches its own)\n",
                   wps, t.name, ms, per_clk, per_clk * t.bytes);
        }
    }
    return 0;
}
