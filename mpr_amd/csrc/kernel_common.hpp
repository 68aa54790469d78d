/* kernel_common.hpp — small device helpers shared by kernels.hip and kernels_float.hip */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <mutex>

#include "device_math.hpp"
#include "kernels.hpp"

namespace mprk {

/* hipFuncSetAttribute belongs to the function ON THE DEVICE THAT IS CURRENT: a process that drives several devices (one host
 * thread per GPU, benchmark/render_table_multi.cpp) opts in once per device, not once per process.  A second thread on the same
 * device waits until the first one has set the attributes (it could launch a kernel that needs the opt-in before). */
struct OncePerDevice {
    std::mutex m;
    unsigned long long done = 0;
    template <class F>
    void run(F&& f)
    {
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long bit = 1ull << (dev & 63);
        std::lock_guard<std::mutex> lock(m);
        if (done & bit) return;
        f();
        done |= bit;
    }
};

struct int4_ { int x, y, z, w; };
/* src/context.cu:23-30 */
DEV int4_ unpack(int pos, int tps)
{
    int4_ r;
    if ((tps & (tps - 1)) == 0) {
        /* power-of-two tile counts (every benchmark size): shifts instead of three integer divisions */
        const int sh = 31 - __builtin_clz((unsigned)tps);
        r.x = pos & (tps - 1);
        r.y = (pos >> sh) & (tps - 1);
        r.z = pos >> (2 * sh);
        r.w = pos & (tps * tps - 1);
        return r;
    }
    r.x = pos % tps;
    r.y = (pos / tps) % tps;
    r.z = (pos / tps) / tps;
    r.w = pos % (tps * tps);
    return r;
}

DEV int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
DEV uint64_t ballot(bool p) { return __ballot(p); }
DEV int rank_in(uint64_t mask, int lane) { return __popcll(mask & ((1ull << lane) - 1ull)); }
DEV uint64_t rfl64(uint64_t v)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
/* The readlane / readfirstlane builtins return a signed int: always go through these so that
 * widening to 64 bits cannot sign-extend (a mask with bit 31 set would otherwise turn its upper
 * half into all ones). */
DEV uint32_t rdlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
DEV uint32_t rdfirst(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
DEV uint64_t rdlane64(uint64_t v, uint32_t l) { return ((uint64_t)rdlane((uint32_t)(v >> 32), l) << 32) | rdlane((uint32_t)v, l); }
DEV float immf(uint64_t d) { return mpr_u2f((uint32_t)(d >> 32)); }


/* Make a wave-uniform value live in a VGPR and opaque to the compiler's uniformity analysis, so
 * that arithmetic on it is issued on the VALU (1.65 wave-instr/clk/CU, four pipes per CU)
 * instead of the single scalar ALU of the CU (0.95 instr/clk/CU), which is the bottleneck of
 * a tape interpreter (scripts/ubench/issue_rates.hip). */
DEV uint32_t to_vgpr(uint32_t x)
{
    asm volatile("" : "+v"(x));
    return x;
}

}  // namespace mprk
