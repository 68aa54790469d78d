/*
 * kernels_effects.hip — mpr::Effects (reference src/effects.cu): screen-space ambient occlusion,
 * its edge-preserving blur, and the single-light shading, over the heightmap + normals a
 * render3D leaves on the device.  Image-space, one thread per pixel; the arithmetic is
 * include/mpr_effects_math.h (shared with the oracle).  The SSAO pass reads depth at 64 scattered
 * sample positions within a 0.1 radius of the pixel: the working set of a 16 x 16 block is a few
 * KiB of the depth image, served by L1/L2; all three passes are bound by HBM traffic of the
 * S x S images (12-20 bytes per pixel) — microseconds next to the frame that produced them.
 */
#include <hip/hip_runtime.h>

#include "../../include/mpr_effects_math.h"
#include "kernels.hpp"

namespace mprk {

struct EffectTables {
    float kernel[64 * 3];
    float rvecs[256 * 3];
};

__global__ void __launch_bounds__(256)
k_draw_ssao(const int32_t* __restrict__ depth, const uint32_t* __restrict__ norm, const EffectTables* __restrict__ tab, int S,
            int32_t* __restrict__ output)
{
    __shared__ float kernel[64 * 3];
    __shared__ float rvecs[256 * 3];
    const int t = threadIdx.x + threadIdx.y * 16;
    if (t < 192) kernel[t] = tab->kernel[t];
    for (int i = t; i < 768; i += 256) rvecs[i] = tab->rvecs[i];
    __syncthreads();
    const int x = threadIdx.x + blockIdx.x * 16, y = threadIdx.y + blockIdx.y * 16;
    if (x >= S || y >= S) return;
    const int32_t o = mpr_fx_ssao_pixel(depth, norm, kernel, rvecs, S, x, y);
    if (o >= 0) output[x + y * S] = o;
}

__global__ void __launch_bounds__(256)
k_blur_ssao(const int32_t* __restrict__ image, const int32_t* __restrict__ ssao, int S, int32_t* __restrict__ output)
{
    const int x = threadIdx.x + blockIdx.x * 16, y = threadIdx.y + blockIdx.y * 16;
    if (x >= S || y >= S) return;
    output[x + y * S] = mpr_fx_blur_pixel(image, ssao, S, x, y);
}

__global__ void __launch_bounds__(256)
k_draw_shaded(const int32_t* __restrict__ depth, const uint32_t* __restrict__ norm, const int32_t* __restrict__ ssao, int S,
              int32_t* __restrict__ output)
{
    const int x = threadIdx.x + blockIdx.x * 16, y = threadIdx.y + blockIdx.y * 16;
    if (x >= S || y >= S) return;
    const uint32_t c = mpr_fx_shade_pixel(depth, norm, ssao, S, x, y);
    if (c) output[x + y * S] = (int32_t)c;
}

size_t effect_tables_bytes() { return sizeof(EffectTables); }
void launch_draw_ssao(hipStream_t s, const int32_t* depth, const uint32_t* norm, const void* tables, int S, int32_t* out)
{
    const unsigned u = (unsigned)(S + 15) / 16;
    hipLaunchKernelGGL(k_draw_ssao, dim3(u, u), dim3(16, 16), 0, s, depth, norm, (const EffectTables*)tables, S, out);
}
void launch_blur_ssao(hipStream_t s, const int32_t* image, const int32_t* ssao, int S, int32_t* out)
{
    const unsigned u = (unsigned)(S + 15) / 16;
    hipLaunchKernelGGL(k_blur_ssao, dim3(u, u), dim3(16, 16), 0, s, image, ssao, S, out);
}
void launch_draw_shaded(hipStream_t s, const int32_t* depth, const uint32_t* norm, const int32_t* ssao, int S, int32_t* out)
{
    const unsigned u = (unsigned)(S + 15) / 16;
    hipLaunchKernelGGL(k_draw_shaded, dim3(u, u), dim3(16, 16), 0, s, depth, norm, ssao, S, out);
}

}  // namespace mprk
