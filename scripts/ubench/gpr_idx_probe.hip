// s_set_gpr_idx_on on gfx950: does a VGPR file indexed by a scalar behave as documented for GFX9, and what does an access cost?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_probe(const int* idx, float* out, long long* cycles)
{
    const int lane = threadIdx.x;
    float r0 = 0, r1 = 0;
    const int i0 = __builtin_amdgcn_readfirstlane(idx[0]), i1 = __builtin_amdgcn_readfirstlane(idx[1]);
    const float lf = (float)lane;
    asm volatile(
        // v100 + k = 100 k + lane
        "v_mov_b32 v100, %[lf]\n v_add_f32 v101, 100.0, v100\n v_add_f32 v102, 100.0, v101\n v_add_f32 v103, 100.0, v102\n"
        "v_add_f32 v104, 100.0, v103\n v_add_f32 v105, 100.0, v104\n v_add_f32 v106, 100.0, v105\n v_add_f32 v107, 100.0, v106\n"
        "s_set_gpr_idx_on %[i0], gpr_idx(SRC0)\n"
        "v_mov_b32 %[r0], v100\n"                       // = v[100 + i0]
        "s_set_gpr_idx_off\n"
        "s_set_gpr_idx_on %[i1], gpr_idx(DST)\n"
        "v_mov_b32 v100, 1.0\n"                         // v[100 + i1] = 1.0
        "s_set_gpr_idx_off\n"
        "s_set_gpr_idx_on %[i1], gpr_idx(SRC0)\n"
        "v_mov_b32 %[r1], v100\n"
        "s_set_gpr_idx_off\n"
        : [r0] "=&v"(r0), [r1] "=&v"(r1) : [lf] "v"(lf), [i0] "s"(i0), [i1] "s"(i1)
        : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "memory");
    out[lane] = r0;
    out[64 + lane] = r1;
    // cost: a dependent chain of indexed read-modify-write accesses
    long long t0 = __builtin_readcyclecounter();
    float acc = lf;
    int j = i0;
    for (int k = 0; k < 256; ++k) {
        j = __builtin_amdgcn_readfirstlane((j * 5 + 1) & 7);
        asm volatile(
            "s_set_gpr_idx_on %[j], gpr_idx(SRC0)\n v_mov_b32 v110, v100\n s_set_gpr_idx_off\n"
            "v_add_f32 %[acc], %[acc], v110\n"
            "s_set_gpr_idx_on %[j], gpr_idx(DST)\n v_mov_b32 v100, %[acc]\n s_set_gpr_idx_off\n"
            : [acc] "+&v"(acc) : [j] "s"(j)
            : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v110", "memory");
    }
    long long t1 = __builtin_readcyclecounter();
    out[128 + lane] = acc;
    if (lane == 0) cycles[0] = t1 - t0;
}
int main()
{
    int h[2] = {3, 5}, *d; float* o; long long* c;
    hipMalloc(&d, 8); hipMalloc(&o, 192 * 4); hipMalloc(&c, 8);
    hipMemcpy(d, h, 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d, o, c);
    std::vector<float> r(192); long long cyc = 0;
    hipMemcpy(r.data(), o, 192 * 4, hipMemcpyDeviceToHost); hipMemcpy(&cyc, c, 8, hipMemcpyDeviceToHost);
    printf("indexed read v[100+3], lanes 0, 1, 63: %g %g %g (expect 300 301 363)\n", r[0], r[1], r[63]);
    printf("indexed write then read v[100+5]: %g %g (expect 1 1)\n", r[64], r[127]);
    printf("256 dependent read-add-write accesses: %lld cycles = %.1f per access pair\n", cyc, cyc / 256.0);
    return 0;
}
