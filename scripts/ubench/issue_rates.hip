// Micro-benchmarks of instruction issue rates on gfx950 that decide the interpreter design:
// SALU vs VALU throughput per CU, whether they overlap, cost of M0-relative VGPR access,
// and scalar-branch cost.  Build: hipcc --offload-arch=gfx950 -O3 issue_rates.hip -o issue_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

__global__ void k_salu(int iters, int* out) {
    int a = threadIdx.x, b = 1, c = 2, d = 3, e = 4;
    for (int i = 0; i < iters; ++i) {
        asm volatile(REP16("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n")
                     : "+s"(b), "+s"(c), "+s"(d), "+s"(e) : : "scc");
    }
    if (a == 9999) out[0] = b + c + d + e;
}
__global__ void k_valu(int iters, float* out) {
    float a = threadIdx.x, b = 1, c = 2, d = 3;
    for (int i = 0; i < iters; ++i) {
        asm volatile(REP16("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    }
    if (a == 9999.f) out[0] = a + b + c + d;
}
__global__ void k_mixed(int iters, float* out) {   // 1 SALU : 1 VALU
    float a = threadIdx.x, b = 1; int s0 = 1, s1 = 2;
    for (int i = 0; i < iters; ++i) {
        asm volatile(REP16("v_add_f32 %0, %0, %0\n s_add_u32 %2, %2, 1\n v_add_f32 %1, %1, %1\n s_add_u32 %3, %3, 1\n"
                           "v_add_f32 %0, %0, %0\n s_add_u32 %2, %2, 1\n v_add_f32 %1, %1, %1\n s_add_u32 %3, %3, 1\n")
                     : "+v"(a), "+v"(b), "+s"(s0), "+s"(s1) : : "scc");
    }
    if (a == 9999.f) out[0] = a + b + s0 + s1;
}
__global__ void k_mixed31(int iters, float* out) {   // 3 SALU : 1 VALU
    float a = threadIdx.x; int s0 = 1, s1 = 2, s2 = 3;
    for (int i = 0; i < iters; ++i) {
        asm volatile(REP16("v_add_f32 %0, %0, %0\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n")
                     : "+v"(a), "+s"(s0), "+s"(s1), "+s"(s2) : : "scc");
    }
    if (a == 9999.f) out[0] = a + s0 + s1 + s2;
}
__global__ void k_movrel(int iters, float* out) {   // s_set_gpr_idx_on + v_mov + s_set_gpr_idx_off per indexed access (5 instr / iteration)
    float r[8]; for (int i = 0; i < 8; ++i) r[i] = threadIdx.x + i;
    float acc = 0; int idx = 1;
    for (int i = 0; i < iters; ++i) {
        asm volatile(REP16("s_and_b32 %2, %2, 3\n s_set_gpr_idx_on %2, gpr_idx(SRC0)\n v_mov_b32 %0, %3\n s_set_gpr_idx_off\n v_add_f32 %1, %1, %0\n")
                     : "=&v"(r[7]), "+v"(acc), "+s"(idx) : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]) : "scc");
    }
    if (acc == 9999.f) out[0] = acc + r[7];
}
__global__ void k_branch(int iters, int* out) {   // s_cmp + s_cbranch (not taken) pairs
    int a = threadIdx.x, s = 5;
    for (int i = 0; i < iters; ++i) {
        asm volatile(REP16("s_cmp_eq_u32 %0, 77\n s_cbranch_scc1 1f\n s_cmp_eq_u32 %0, 78\n s_cbranch_scc1 1f\n"
                           "s_cmp_eq_u32 %0, 79\n s_cbranch_scc1 1f\n s_cmp_eq_u32 %0, 80\n s_cbranch_scc1 1f\n")
                     "1:\n" : "+s"(s) : : "scc");
    }
    if (a == 9999) out[0] = s;
}
__global__ void k_lds(int iters, float* out) {   // ds_read_b32 x2 + ds_write_b32 per "clause"
    __shared__ float lds[64 * 64 * 4];
    float* p = lds + (threadIdx.x >> 6) * 64 * 64 + (threadIdx.x & 63);
    float acc = threadIdx.x;
    for (int i = 0; i < 64; ++i) p[i * 64] = acc;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float a = p[((k * 7) & 63) * 64], b = p[((k * 11 + 3) & 63) * 64];
            p[((k * 5 + 1) & 63) * 64] = a + b;
        }
    }
    out[threadIdx.x + blockIdx.x * blockDim.x] = p[0];
}

template <typename F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    int* di; float* df; CK(hipMalloc(&di, 1 << 22)); CK(hipMalloc(&df, 1 << 22));
    const int iters = 2000;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount; const double ghz = p.clockRate / 1e6;
    printf("CUs %d clock %.2f GHz\n", cus, ghz);
    for (int wpc : {4, 8, 16, 32}) {           // waves per CU
        dim3 g(cus * wpc / 4), b(256);
        auto rep = [&](const char* name, float ms, double instr_per_wave) {
            double waves = (double)cus * wpc;
            double per_cu_per_clk = instr_per_wave * waves / cus / (ms * 1e-3 * ghz * 1e9);
            printf("  %-10s waves/CU %2d: %8.3f ms  -> %.3f wave-instr/clk/CU (nominal clock)\n", name, wpc, ms, per_cu_per_clk);
        };
        rep("salu", timeit([&] { hipLaunchKernelGGL(k_salu, g, b, 0, 0, iters, di); }), iters * 64.0);
        rep("valu", timeit([&] { hipLaunchKernelGGL(k_valu, g, b, 0, 0, iters, df); }), iters * 64.0);
        rep("mixed1:1", timeit([&] { hipLaunchKernelGGL(k_mixed, g, b, 0, 0, iters, df); }), iters * 128.0);
        rep("mixed3:1", timeit([&] { hipLaunchKernelGGL(k_mixed31, g, b, 0, 0, iters, df); }), iters * 64.0);
        rep("gpridx(5i)", timeit([&] { hipLaunchKernelGGL(k_movrel, g, b, 0, 0, iters, df); }), iters * 80.0);
        rep("cmp+br", timeit([&] { hipLaunchKernelGGL(k_branch, g, b, 0, 0, iters, di); }), iters * 128.0);
        rep("lds 2r1w", timeit([&] { hipLaunchKernelGGL(k_lds, g, b, 0, 0, iters / 10, df); }), iters / 10 * 16.0 * 3);
    }
    return 0;
}
