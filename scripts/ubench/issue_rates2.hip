// Issue rates of the instruction kinds the tape interpreters are made of, on gfx950 (development aid).
// Reports wave-instructions per clock per CU at 4..32 wavefronts per CU, so that the VALU / SALU / LDS
// ceilings used in DESIGN.md and bench.py (roofline.valu_frac etc.) can be recomputed from a tracked file.
// Build: hipcc --offload-arch=gfx950 -O3 issue_rates2.hip -o issue_rates2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

// 64 instructions per loop iteration, four independent chains
#define VKERNEL(name, body)                                                                  \
__global__ void name(int iters, float* out) {                                               \
    float a = threadIdx.x, b = 1, c = 2, d = 3; float e = 0.5f, f = 1.5f;                     \
    for (int i = 0; i < iters; ++i) {                                                        \
        asm volatile(REP16(body) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f) : "vcc", "scc", "s40", "s41", "s42", "s43", "m0"); \
    }                                                                                        \
    if (a == 9999.f) out[0] = a + b + c + d;                                                 \
}
VKERNEL(k_add,  "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n")
VKERNEL(k_fma,  "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")
VKERNEL(k_dep,  "v_add_f32 %0, %0, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %0, %0, %4\n v_add_f32 %0, %0, %4\n")
VKERNEL(k_mov,  "v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4\n")
VKERNEL(k_perm, "v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5\n")
VKERNEL(k_rdln, "v_readlane_b32 s40, %0, 3\n v_readlane_b32 s41, %1, 5\n v_readlane_b32 s42, %2, 7\n v_readlane_b32 s43, %3, 9\n")
VKERNEL(k_exp,  "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n")
VKERNEL(k_rcp,  "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n")
VKERNEL(k_sqrt, "v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n")
VKERNEL(k_cnd,  "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n")
VKERNEL(k_cnd64, "v_cndmask_b32_e64 %0, %0, %4, s[40:41]\n v_cndmask_b32_e64 %1, %1, %4, s[40:41]\n v_cndmask_b32_e64 %2, %2, %4, s[42:43]\n v_cndmask_b32_e64 %3, %3, %4, s[42:43]\n")
VKERNEL(k_cndind, "v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %4, %5, vcc\n v_cndmask_b32 %2, %5, %4, vcc\n v_cndmask_b32 %3, %5, %4, vcc\n")
VKERNEL(k_cmpcnd, "v_cmp_gt_f32 vcc, %0, %4\n s_nop 1\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_gt_f32 vcc, %2, %4\n s_nop 1\n v_cndmask_b32 %3, %3, %4, vcc\n")
VKERNEL(k_addsgpr, "v_add_f32 %0, s40, %0\n v_add_f32 %1, s41, %1\n v_add_f32 %2, s42, %2\n v_add_f32 %3, s43, %3\n")
VKERNEL(k_addvcc, "v_addc_co_u32 %0, vcc, %0, %4, vcc\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_addc_co_u32 %2, vcc, %2, %4, vcc\n v_addc_co_u32 %3, vcc, %3, %4, vcc\n")
VKERNEL(k_ldexp, "v_ldexp_f32 %0, %0, %4\n v_ldexp_f32 %1, %1, %4\n v_ldexp_f32 %2, %2, %4\n v_ldexp_f32 %3, %3, %4\n")
VKERNEL(k_divscale, "v_div_scale_f32 %0, vcc, %0, %4, %0\n v_div_scale_f32 %1, vcc, %1, %4, %1\n v_div_fmas_f32 %2, %2, %4, %5\n v_div_fmas_f32 %3, %3, %4, %5\n")
VKERNEL(k_cvt, "v_cvt_i32_f32 %0, %0\n v_cvt_f32_i32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_f32_i32 %3, %3\n")
VKERNEL(k_cmp,  "v_cmp_gt_f32 vcc, %0, %4\n v_cmp_gt_f32 vcc, %1, %4\n v_cmp_gt_f32 vcc, %2, %4\n v_cmp_gt_f32 vcc, %3, %4\n")
VKERNEL(k_lit,  "v_add_f32 %0, 0x40490fdb, %0\n v_add_f32 %1, 0x40490fdb, %1\n v_add_f32 %2, 0x40490fdb, %2\n v_add_f32 %3, 0x40490fdb, %3\n")
VKERNEL(k_vop3, "v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_div_fixup_f32 %2, %2, %4, %5\n v_div_fixup_f32 %3, %3, %4, %5\n")
VKERNEL(k_snop, "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n")
VKERNEL(k_addnop, "v_add_f32 %0, %0, %4\n s_nop 0\n v_add_f32 %1, %1, %4\n s_nop 0\n")
VKERNEL(k_addwait, "v_add_f32 %0, %0, %4\n s_waitcnt lgkmcnt(0)\n v_add_f32 %1, %1, %4\n s_waitcnt lgkmcnt(0)\n")
VKERNEL(k_salu, "s_add_u32 s40, s40, 1\n s_add_u32 s41, s41, 1\n s_add_u32 s42, s42, 1\n s_add_u32 s43, s43, 1\n")
VKERNEL(k_m0,   "s_and_b32 m0, s40, 0xff00\n s_lshr_b32 m0, s41, 16\n s_and_b32 m0, s42, 0xff00\n s_lshr_b32 m0, s43, 16\n")
VKERNEL(k_mix11, "v_add_f32 %0, %0, %4\n s_add_u32 s40, s40, 1\n v_add_f32 %1, %1, %4\n s_add_u32 s41, s41, 1\n")
VKERNEL(k_mix13, "v_add_f32 %0, %0, %4\n s_add_u32 s40, s40, 1\n s_add_u32 s41, s41, 1\n s_add_u32 s42, s42, 1\n")
VKERNEL(k_mix31, "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n s_add_u32 s41, s41, 1\n")

__global__ void k_pk(int iters, float* out) {
    double a = threadIdx.x, b = 1, c = 2, d = 3, e = 0.5;
    for (int i = 0; i < iters; ++i) {
        asm volatile(REP16("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));
    }
    if (a == 9999.) out[0] = a + b + c + d;
}
__global__ void k_f64(int iters, float* out) {
    double a = threadIdx.x, b = 1, c = 2, d = 3, e = 0.5;
    for (int i = 0; i < iters; ++i) {
        asm volatile(REP16("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));
    }
    if (a == 9999.) out[0] = a + b + c + d;
}

// LDS: one wavefront per 4 KB slice; reads / writes with a VGPR address, and with M0 + lane * 4
template <int MODE>
__global__ void k_lds(int iters, float* out) {
    extern __shared__ float lds[];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* p = lds + w * 1024;
    for (int i = 0; i < 16; ++i) p[i * 64 + lane] = (float)(i * 64 + lane);
    __syncthreads();
    uint32_t addr = (uint32_t)(uintptr_t)(p + lane);
    uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)p);
    float a = 0, b = 0, c = 0, d = 0;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0)
            asm volatile(REP16("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:256\n ds_read_b32 %2, %4 offset:512\n ds_read_b32 %3, %4 offset:768\n") "s_waitcnt lgkmcnt(0)\n"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(addr) : "memory");
        if (MODE == 1)
            asm volatile("s_mov_b32 m0, %5\n" REP16("ds_read_addtid_b32 %0\n ds_read_addtid_b32 %1 offset:256\n ds_read_addtid_b32 %2 offset:512\n ds_read_addtid_b32 %3 offset:768\n") "s_waitcnt lgkmcnt(0)\n"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(addr), "s"(base) : "memory", "m0");
        if (MODE == 2)
            asm volatile(REP16("ds_write_b32 %4, %0\n ds_write_b32 %4, %1 offset:256\n ds_write_b32 %4, %2 offset:512\n ds_write_b32 %4, %3 offset:768\n") "s_waitcnt lgkmcnt(0)\n"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(addr) : "memory");
        if (MODE == 3)
            asm volatile("s_mov_b32 m0, %5\n" REP16("ds_write_addtid_b32 %0\n ds_write_addtid_b32 %1 offset:256\n ds_write_addtid_b32 %2 offset:512\n ds_write_addtid_b32 %3 offset:768\n") "s_waitcnt lgkmcnt(0)\n"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(addr), "s"(base) : "memory", "m0");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
// what address does ds_read_addtid_b32 form?  out[lane] = value read with M0 = m0val, offset 256 (floats hold their own index)
__global__ void k_addtid_sem(uint32_t m0val, float* out) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (float)i;
    __syncthreads();
    float r;
    asm volatile("s_mov_b32 m0, %1\n s_nop 0\n ds_read_addtid_b32 %0 offset:256\n s_waitcnt lgkmcnt(0)\n" : "=v"(r) : "s"(m0val) : "memory", "m0");
    out[threadIdx.x] = r;
}

template <typename F> float timeit(F f) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    float* df; CK(hipMalloc(&df, 1 << 24));
    const int iters = 1000;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount; const double ghz = p.clockRate / 1e6;
    printf("device %s CUs %d clock %.3f GHz (rates below are per nominal clock)\n", p.gcnArchName, cus, ghz);
    {   // semantics of ds_read_addtid_b32
        float h[64];
        for (uint32_t m0 : {0u, 1024u, 0x10400u}) {
            hipLaunchKernelGGL(k_addtid_sem, dim3(1), dim3(64), 16384, 0, m0, df);
            CK(hipMemcpy(h, df, sizeof(h), hipMemcpyDeviceToHost));
            printf("ds_read_addtid_b32 offset:256, M0 = 0x%x: lane 0 read word %g, lane 1 word %g, lane 63 word %g (M0 + offset + lane*4 would be %u, %u, %u)\n",
                   m0, h[0], h[1], h[63], (m0 & 0xffff) / 4 + 64, (m0 & 0xffff) / 4 + 65, (m0 & 0xffff) / 4 + 127);
        }
    }
    typedef void (*kfn)(int, float*);
    struct T { const char* name; kfn fn; double per_iter; };
    std::vector<T> tests = {
        {"v_add_f32", k_add, 64}, {"v_fma_f32", k_fma, 64}, {"v_add dep.chain", k_dep, 64}, {"v_mov_b32", k_mov, 64},
        {"v_perm_b32", k_perm, 64}, {"v_readlane", k_rdln, 64}, {"v_exp_f32", k_exp, 64}, {"v_rcp_f32", k_rcp, 64},
        {"v_sqrt_f32", k_sqrt, 64}, {"v_cndmask", k_cnd, 64}, {"v_cndmask e64 sgpr", k_cnd64, 64}, {"v_cndmask independent", k_cndind, 64}, {"v_cmp+nop+cndmask (instr)", k_cmpcnd, 96}, {"v_add sgpr src", k_addsgpr, 64}, {"v_addc vcc", k_addvcc, 64}, {"v_ldexp", k_ldexp, 64}, {"v_div_scale/fmas", k_divscale, 64}, {"v_cvt", k_cvt, 64}, {"v_cmp", k_cmp, 64}, {"v_add literal", k_lit, 64},
        {"vop3 fma/fixup", k_vop3, 64}, {"v_pk_fma_f32", k_pk, 64}, {"v_fma_f64", k_f64, 64},
        {"s_nop", k_snop, 64}, {"v_add+s_nop (instr)", k_addnop, 64}, {"v_add+s_waitcnt (instr)", k_addwait, 64},
        {"s_add_u32", k_salu, 64}, {"s_and/lshr m0", k_m0, 64},
        {"mix 1v:1s (instr)", k_mix11, 64}, {"mix 1v:3s (instr)", k_mix13, 64}, {"mix 3v:1s (instr)", k_mix31, 64},
    };
    for (int wpc : {8, 32}) {
        dim3 g(cus * wpc / 4), b(256);
        printf("waves/CU %d\n", wpc);
        for (auto& t : tests) {
            float ms = timeit([&] { hipLaunchKernelGGL(t.fn, g, b, 0, 0, iters, df); });
            double rate = t.per_iter * iters * wpc / (ms * 1e-3 * ghz * 1e9);
            printf("  %-26s %8.3f ms  %.3f wave-instr/clk/CU\n", t.name, ms, rate);
        }
        const char* ln[4] = {"ds_read_b32", "ds_read_addtid_b32", "ds_write_b32", "ds_write_addtid_b32"};
        for (int m = 0; m < 4; ++m) {
            auto f = [&] {
                if (m == 0) hipLaunchKernelGGL(k_lds<0>, g, b, 16384, 0, iters, df);
                if (m == 1) hipLaunchKernelGGL(k_lds<1>, g, b, 16384, 0, iters, df);
                if (m == 2) hipLaunchKernelGGL(k_lds<2>, g, b, 16384, 0, iters, df);
                if (m == 3) hipLaunchKernelGGL(k_lds<3>, g, b, 16384, 0, iters, df);
            };
            float ms = timeit(f);
            printf("  %-26s %8.3f ms  %.3f wave-instr/clk/CU\n", ln[m], ms, 64.0 * iters * wpc / (ms * 1e-3 * ghz * 1e9));
        }
    }
    return 0;
}
