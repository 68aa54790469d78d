// Feasibility probe for device-side code generation on gfx950 (development aid, not product code):
//   1. executable device memory through HSA (hsa_amd_memory_pool_allocate + HSA_AMD_MEMORY_POOL_EXECUTABLE_FLAG);
//   2. a wavefront writes machine code with ordinary vector stores and then jumps into it: which
//      fences does the instruction fetch need (nothing / s_icache_inv / L2 write-back + s_icache_inv)
//      when the same addresses are rewritten with different code round after round;
//   3. throughput of straight-line code that is unique per wavefront (streams through the
//      instruction cache) against the same instructions in a loop (instruction cache hits).
// Build: hipcc --offload-arch=gfx950 -O3 jit_probe.hip -o jit_probe -lhsa-runtime64
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

static hsa_agent_t g_gpu; static int g_have_gpu = 0;
static hsa_amd_memory_pool_t g_pool; static int g_have_pool = 0;
static hsa_status_t agent_cb(hsa_agent_t a, void*) {
    hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) { g_gpu = a; g_have_gpu = 1; }
    return HSA_STATUS_SUCCESS;
}
static hsa_status_t pool_cb(hsa_amd_memory_pool_t p, void*) {
    hsa_amd_segment_t seg; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    uint32_t flags = 0; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    bool alloc = false; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
    size_t size = 0; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SIZE, &size);
    printf("  pool: flags 0x%x alloc %d size %.1f GB\n", flags, (int)alloc, size / 1e9);
    if (alloc && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !g_have_pool) { g_pool = p; g_have_pool = 1; }
    return HSA_STATUS_SUCCESS;
}

constexpr uint32_t ENC_ADD = 0x02505328u;      // v_add_f32 v40, v40, v41
constexpr uint32_t ENC_MUL = 0x0a505328u;      // v_mul_f32 v40, v40, v41
constexpr uint32_t ENC_RET = 0xbe801d1eu;      // s_setpc_b64 s[30:31]
constexpr int REGION_DWORDS = 8192;            // 32 KB per wavefront

// fence: 0 = s_waitcnt vmcnt(0) only, 1 = + s_icache_inv, 2 = + buffer_wbl2 sc1 + s_icache_inv
template <int FENCE>
__global__ void __launch_bounds__(64) k_jit(uint32_t* code, int n_instr, int rounds, int stride_regions,
                                            unsigned* bad, unsigned long long* cycles, float* out)
{
    const int lane = threadIdx.x;
    uint32_t* my = code + (size_t)blockIdx.x * REGION_DWORDS;
    float acc = 0.0f;
    float expect = 0.0f;
    unsigned nbad = 0;
    unsigned long long t_call = 0;
    for (int r = 0; r < rounds; ++r) {
        // optionally move to a fresh region every round (stride_regions > 0): never-reused addresses
        uint32_t* dst = my + (size_t)r * stride_regions * REGION_DWORDS * gridDim.x;
        const uint32_t word = (r & 1) ? ENC_MUL : ENC_ADD;
        for (int i = lane; i < n_instr; i += 64) dst[i] = word;
        if (lane == 0) dst[n_instr] = ENC_RET;
        const unsigned long long addr = (unsigned long long)dst;
        const uint32_t alo = __builtin_amdgcn_readfirstlane((uint32_t)addr), ahi = __builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32));
        unsigned long long t0, t1;
        asm volatile(
            "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
            ".if %[fence] == 2\n buffer_wbl2 sc1\n s_waitcnt vmcnt(0)\n .endif\n"
            ".if %[fence] >= 1\n s_icache_inv\n .endif\n"
            "s_nop 7\n s_nop 7\n"
            "s_mov_b32 s40, %[alo]\n s_mov_b32 s41, %[ahi]\n"
            "v_mov_b32 v40, %[acc]\n v_mov_b32 v41, 1.0\n"
            "s_memtime %[t0]\n s_waitcnt lgkmcnt(0)\n"
            "s_swappc_b64 s[30:31], s[40:41]\n"
            "s_memtime %[t1]\n s_waitcnt lgkmcnt(0)\n"
            "v_mov_b32 %[acc], v40\n"
            : [acc] "+v"(acc), [t0] "=&s"(t0), [t1] "=&s"(t1)
            : [alo] "s"(alo), [ahi] "s"(ahi), [fence] "n"(FENCE)
            : "memory", "s30", "s31", "s40", "s41", "v40", "v41", "scc");
        t_call += t1 - t0;
        if (!(r & 1)) expect += (float)n_instr;      // add rounds add n * 1.0; mul rounds multiply by 1.0
        if (acc != expect) { ++nbad; acc = expect; }
    }
    if (lane == 0) {
        atomicAdd(bad, nbad);
        atomicAdd(cycles, t_call);
        out[blockIdx.x] = acc;
    }
}

// the same number of v_add_f32 from a loop that stays in the instruction cache
__global__ void __launch_bounds__(64) k_loop(int n_instr, int rounds, float* out, unsigned long long* cycles)
{
    float acc = 0.0f;
    unsigned long long t_call = 0;
    for (int r = 0; r < rounds; ++r) {
        unsigned long long t0, t1;
        int n = __builtin_amdgcn_readfirstlane(n_instr / 64);
        asm volatile(
            "v_mov_b32 v40, %[acc]\n v_mov_b32 v41, 1.0\n"
            "s_memtime %[t0]\n s_waitcnt lgkmcnt(0)\n"
            "1:\n"
            ".rept 64\n v_add_f32 v40, v40, v41\n .endr\n"
            "s_sub_u32 %[n], %[n], 1\n s_cmp_lg_u32 %[n], 0\n s_cbranch_scc1 1b\n"
            "s_memtime %[t1]\n s_waitcnt lgkmcnt(0)\n"
            "v_mov_b32 %[acc], v40\n"
            : [acc] "+v"(acc), [t0] "=&s"(t0), [t1] "=&s"(t1), [n] "+s"(n)
            :
            : "memory", "v40", "v41", "scc");
        t_call += t1 - t0;
    }
    if (threadIdx.x == 0) { atomicAdd(cycles, t_call); out[blockIdx.x] = acc; }
}

template <typename F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

int main(int argc, char** argv)
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    CK(hipSetDevice(0));
    void* warm; CK(hipMalloc(&warm, 4096));                 // makes sure HIP has initialised HSA
    if (hsa_init() != HSA_STATUS_SUCCESS) { printf("hsa_init failed\n"); return 2; }
    hsa_iterate_agents(agent_cb, nullptr);
    if (!g_have_gpu) { printf("no HSA GPU agent\n"); return 2; }
    hsa_amd_agent_iterate_memory_pools(g_gpu, pool_cb, nullptr);
    if (!g_have_pool) { printf("no coarse-grained pool\n"); return 2; }
    const int blocks_max = 256 * 32;                        // one region per wavefront of the largest grid
    const size_t region_bytes = (size_t)REGION_DWORDS * 4;
    const int fresh_rounds = 8;
    const size_t bytes = region_bytes * blocks_max * (fresh_rounds + 1);
    void* code = nullptr;
    hsa_status_t st = hsa_amd_memory_pool_allocate(g_pool, bytes, HSA_AMD_MEMORY_POOL_EXECUTABLE_FLAG, &code);
    printf("executable allocation of %.1f MB: status %d ptr %p\n", bytes / 1e6, (int)st, code);
    if (st != HSA_STATUS_SUCCESS) return 3;
    CK(hipMemset(code, 0, bytes));
    unsigned* bad; unsigned long long* cyc; float* out;
    CK(hipMalloc(&bad, 4)); CK(hipMalloc(&cyc, 8)); CK(hipMalloc(&out, blocks_max * 4));
    const int cus = p.multiProcessorCount;
    printf("CUs %d clock %.2f GHz\n", cus, p.clockRate / 1e6);

    auto run = [&](int fence, int blocks, int n_instr, int rounds, int stride) {
        hipMemset(bad, 0, 4); hipMemset(cyc, 0, 8);
        float ms = timeit([&] {
            if (fence == 0) hipLaunchKernelGGL(k_jit<0>, dim3(blocks), dim3(64), 0, 0, (uint32_t*)code, n_instr, rounds, stride, bad, cyc, out);
            if (fence == 1) hipLaunchKernelGGL(k_jit<1>, dim3(blocks), dim3(64), 0, 0, (uint32_t*)code, n_instr, rounds, stride, bad, cyc, out);
            if (fence == 2) hipLaunchKernelGGL(k_jit<2>, dim3(blocks), dim3(64), 0, 0, (uint32_t*)code, n_instr, rounds, stride, bad, cyc, out);
        });
        hipError_t e = hipDeviceSynchronize();
        unsigned hb = 0; unsigned long long hc = 0;
        hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
        printf("  fence %d %s blocks %5d n %5d rounds %2d: %s  bad rounds %u / %d  %8.3f ms  %.2f memtime-ticks/instr/wave\n",
               fence, stride ? "fresh  " : "rewrite", blocks, n_instr, rounds, e == hipSuccess ? "ok" : hipGetErrorString(e),
               hb, blocks * rounds, ms, (double)hc / ((double)blocks * rounds * n_instr));
        return e == hipSuccess;
    };
    // 1. does it run at all: one wavefront, one round
    if (!run(1, 1, 64, 1, 0)) return 4;
    if (!run(1, 1, 4096, 2, 0)) return 4;
    // 2. which fence is needed when code is rewritten in place (stale instruction cache lines show as bad rounds)
    for (int fence : {2, 1, 0}) { run(fence, 64, 2048, 8, 0); run(fence, cus * 8, 2048, 8, 0); }
    // fresh addresses every round, no invalidate
    run(0, cus * 8, 2048, fresh_rounds, 1);
    // 3. throughput: unique code per wavefront against a loop
    for (int wpc : {4, 8, 16, 32}) {
        const int blocks = cus * wpc;
        run(1, blocks, 4096, 4, 0);
        run(0, blocks, 4096, 4, 1);
        hipMemset(cyc, 0, 8);
        float ms = timeit([&] { hipLaunchKernelGGL(k_loop, dim3(blocks), dim3(64), 0, 0, 4096, 4, out, cyc); });
        hipDeviceSynchronize();
        unsigned long long hc = 0; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
        printf("  loop                blocks %5d n  4096 rounds  4:                         %8.3f ms  %.2f memtime-ticks/instr/wave\n",
               blocks, ms, (double)hc / ((double)blocks * 4 * 4096));
    }
    hsa_amd_memory_pool_free(code);
    return 0;
}
