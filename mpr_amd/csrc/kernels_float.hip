/*
 * kernels_float.hip — the round-to-nearest walkers: float voxel / pixel pass and the
 * derivative (normals) pass.  See kernels.hip for the overall design.
 *
 * Like kernels.hip this file is compiled with -mllvm -structurizecfg-skip-uniform-regions=1
 * (see the Makefile): the opcode dispatch of an interpreter is wave-uniform control flow.
 */
#include "kernel_common.hpp"

namespace mprk {

/* ------------------------------------------------------------------------------------ */
/* eval_voxels_f with calculate_voxels / calculate_pixels fused (reference :707-964)     */
/* one wave = the 64 voxels (4x4x4) or pixels (8x8) of one smallest tile                 */
/* ------------------------------------------------------------------------------------ */
/* rare, long opcodes: kept out of line so that the hot loop stays small in the I-cache */
__device__ __noinline__ float rare_unary(uint32_t op, float v)
{
    switch (op) {
        case MPR_OP_SIN_LHS: return mpr_sinf(v);
        case MPR_OP_COS_LHS: return mpr_cosf(v);
        case MPR_OP_ASIN_LHS: return mpr_asinf(v);
        case MPR_OP_ACOS_LHS: return mpr_acosf(v);
        case MPR_OP_ATAN_LHS: return mpr_atanf(v);
        case MPR_OP_EXP_LHS: return mpr_expf(v);
        default: return mpr_logf(v);
    }
}

template <int DIM>
__global__ void __launch_bounds__(256)
k_eval_voxels(VoxelArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    unsigned char* const myslot = smem + (size_t)wave * a.nslots * 256 + lane * 4;   /* slot s: myslot + s*256 */

    const int tile_index = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
    if (tile_index >= a.count) return;
    const uint64_t* __restrict__ const tro = a.tape_ro;
    const int position = __builtin_amdgcn_readfirstlane(a.tiles[tile_index].position);
    const int tape = __builtin_amdgcn_readfirstlane(a.tiles[tile_index].tape);

    constexpr int SUB = (DIM == 3) ? 4 : 8;
    const int S = a.tps * SUB;
    const int4_ pos = unpack(position, a.tps);
    const int4_ sub = unpack(lane, SUB);
    const int px = pos.x * SUB + sub.x;
    const int py = pos.y * SUB + sub.y;
    const int pz = (DIM == 3) ? pos.z * 4 + sub.z : 0;

    bool skip = false;
    if (DIM == 3) {
        /* reference :852-864: the thread owning (pz_low, pz_low + 2) leaves when image >= pz_low + 2 */
        const int pz_low = pos.z * 4 + (sub.z & 1);
        skip = a.image[px + py * S] >= pz_low + 2;
        if (ballot(!skip) == 0) return;
    }

    const float size_recip = 1.0f / (float)(unsigned)(a.tps * SUB);
    const float fx = ((px + 0.5f) * size_recip - 0.5f) * 2.0f;
    const float fy = ((py + 0.5f) * size_recip - 0.5f) * 2.0f;
    float vx, vy, vz;
    if (DIM == 3) {
        const float fz = ((pz + 0.5f) * size_recip - 0.5f) * 2.0f;
        const float fw = a.mat[3] * fx + a.mat[7] * fy + a.mat[11] * fz + a.mat[15];
        vx = (a.mat[0] * fx + a.mat[4] * fy + a.mat[8] * fz + a.mat[12]) / fw;
        vy = (a.mat[1] * fx + a.mat[5] * fy + a.mat[9] * fz + a.mat[13]) / fw;
        vz = (a.mat[2] * fx + a.mat[6] * fy + a.mat[10] * fz + a.mat[14]) / fw;
    } else {
        const float fw = a.mat[2] * fx + a.mat[5] * fy + a.mat[8];
        vx = (a.mat[0] * fx + a.mat[3] * fy + a.mat[6]) / fw;
        vy = (a.mat[1] * fx + a.mat[4] * fy + a.mat[7]) / fw;
        vz = a.z;
    }
    const uint64_t head0 = tro[0];
    *reinterpret_cast<float*>(myslot + ((head0 >> 8) & 0xFF) * 256) = vx;
    *reinterpret_cast<float*>(myslot + ((head0 >> 16) & 0xFF) * 256) = vy;
    *reinterpret_cast<float*>(myslot + ((head0 >> 24) & 0xFF) * 256) = vz;

    /* Clause stream: 64 clauses per coalesced 512-byte load (lane j holds clause j), handed out
     * with v_readlane.  Fields are decoded on the VALU from a VGPR copy of the clause; only the
     * opcode goes to the scalar unit, for the branch. */
    int base = tape + 1;
    uint64_t blk = tro[base + lane];
    int j = 0, words = 0;
    uint32_t dlo = 0, dhi = 0;
    for (;;) {
        if (j == 64) {
            base += 64;
            blk = tro[base + lane];
            j = 0;
        }
        dlo = rdlane((uint32_t)blk, j);
        dhi = rdlane((uint32_t)(blk >> 32), j);
        ++words;
        const uint32_t op = dlo & 0xFF;
        if (op < 2) {
            if (op == 0) break;
            base = base + j + (int32_t)dhi + 1;      /* JUMP */
            blk = tro[base + lane];
            j = 0;
            continue;
        }
        ++j;
        const uint32_t vlo = to_vgpr(dlo);
        const float imm = mpr_u2f(to_vgpr(dhi));
        const uint32_t r8 = vlo >> 24;
        const float A = *reinterpret_cast<const float*>(myslot + ((vlo >> 8) & 0xFF00));
        const float Bs = *reinterpret_cast<const float*>(myslot + (r8 << 8));
        float* const outp = reinterpret_cast<float*>(myslot + (vlo & 0xFF00));
        const float B = r8 ? Bs : imm;              /* immediate forms carry rhs == 0 */
        float out;
        if (op >= MPR_OP_ADD_LHS_IMM) {
            if (op <= MPR_OP_MAX_LHS_RHS) {
                if (op <= MPR_OP_MUL_LHS_RHS) out = (op <= MPR_OP_ADD_LHS_RHS) ? A + B : A * B;
                else out = (op <= MPR_OP_MIN_LHS_RHS) ? mpr_fminf(A, B) : mpr_fmaxf(A, B);
            } else if (op <= MPR_OP_DIV_LHS_RHS) {
                /* B is already the immediate in the LHS_IMM forms; IMM_RHS swaps it in as lhs */
                const float X = (op == MPR_OP_SUB_IMM_RHS || op == MPR_OP_DIV_IMM_RHS) ? imm : A;
                out = (op <= MPR_OP_SUB_LHS_RHS) ? X - B : X / B;
            } else {
                out = (op == MPR_OP_COPY_LHS) ? A : B;      /* COPY_IMM has rhs == 0 */
            }
        } else if (op <= MPR_OP_NEG_LHS) {
            if (op == MPR_OP_SQUARE_LHS) out = A * A;
            else out = (op == MPR_OP_NEG_LHS) ? -A : __builtin_sqrtf(A);
        } else if (op == MPR_OP_ABS_LHS) {
            out = __builtin_fabsf(A);
        } else {
            out = rare_unary(op, A);
        }
        *outp = out;
    }
    const float res = *reinterpret_cast<const float*>(myslot + (dlo & 0xFF00));
    if (!skip && res < 0.0f) {
        if (DIM == 3) {
            int* p = &a.image[px + py * S];
            if (*p < pz) atomicMax(p, pz);
        } else {
            a.image[px + py * S] = 1;
        }
    }
    const unsigned long long visible = (unsigned long long)__popcll(ballot(!skip));
    if (a.counters && lane == 0) {
        atomicAdd((unsigned long long*)&a.counters[CNT_FWD], (unsigned long long)words);
        atomicAdd((unsigned long long*)&a.counters[CNT_FWD_VOX], (unsigned long long)words);
        atomicAdd((unsigned long long*)&a.counters[CNT_LANE], (unsigned long long)(words - 1) * visible);
    }
    if (a.heat) {
        /* heatmap frames (reference eval_voxels_f_heatmap, src/context.cu:1960-1980).  The reference
         * runs 32 threads per tile, two voxels each: in 3-D every thread that was not skipped adds its
         * walk to its (px, py) — the lanes with sub.z < 2 stand for those threads here — and in 2-D a
         * thread adds half of it to each of its two pixels. */
        const float work = (float)(unsigned)(words - 1);
        if (DIM == 3) {
            if (!skip && sub.z < 2) atomicAdd(&a.heat[px + (size_t)py * S], work);
        } else {
            atomicAdd(&a.heat[px + (size_t)py * S], work / 2.0f);
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* eval_pixels_d, quad layout.  A Deriv is four floats (dx, dy, dz, v); instead of giving   */
/* each lane a whole Deriv (16-byte slots, an 8x8 patch per wave, up to a dozen different   */
/* tapes per patch) a pixel is spread over FOUR lanes, one component each: a wave is the    */
/* 16 pixels of one 4x4 footprint (pixels of a footprint share their micro-tile's tape      */
/* unless their heights fall into different tiles), slots are 4 bytes per lane, and the      */
/* value component is broadcast inside each quad with a DPP quad_perm move where the         */
/* product / quotient / chain rules need it.  Same operations per component as Deriv        */
/* (inc/gpu_deriv.hpp), hence the same bits.                                                */
/* ------------------------------------------------------------------------------------ */
DEV float quad_bcast(float x, int which)
{
    const int v = (int)mpr_f2u(x);
    int r;
    switch (which) {
        case 0: r = __builtin_amdgcn_update_dpp(0, v, 0x00, 0xF, 0xF, true); break;
        case 1: r = __builtin_amdgcn_update_dpp(0, v, 0x55, 0xF, 0xF, true); break;
        case 2: r = __builtin_amdgcn_update_dpp(0, v, 0xAA, 0xF, 0xF, true); break;
        default: r = __builtin_amdgcn_update_dpp(0, v, 0xFF, 0xF, 0xF, true); break;
    }
    return mpr_u2f((uint32_t)r);
}

/* rare / long opcodes of the quad layout, out of line.  a, b: this lane's component of lhs / rhs;
 * av, bv: their value components; isv: this lane holds the value component. */
__device__ __noinline__ float deriv_rare_q(uint32_t op, float a, float av, float b, float bv, float imm, bool isv)
{
    switch (op) {
        case MPR_OP_SQRT_LHS: { const float s = __builtin_sqrtf(av); const float d = 2 * s; return isv ? s : a / d; }
        case MPR_OP_SIN_LHS: { const float c = mpr_cosf(av); return isv ? mpr_sinf(av) : c * a; }
        case MPR_OP_COS_LHS: { const float s = -mpr_sinf(av); return isv ? mpr_cosf(av) : s * a; }
        case MPR_OP_ASIN_LHS: { const float d = __builtin_sqrtf(1 - av * av); return isv ? mpr_asinf(av) : a / d; }
        case MPR_OP_ACOS_LHS: { const float d = -__builtin_sqrtf(1 - av * av); return isv ? mpr_acosf(av) : a / d; }
        case MPR_OP_ATAN_LHS: { const float d = av * av + 1; return isv ? mpr_atanf(av) : a / d; }
        case MPR_OP_EXP_LHS: { const float v = mpr_expf(av); return isv ? v : v * a; }
        case MPR_OP_LOG_LHS: return isv ? mpr_logf(av) : a / av;
        case MPR_OP_DIV_LHS_IMM: return a / imm;
        case MPR_OP_DIV_IMM_RHS: { const float d = bv * bv; return isv ? imm / b : (-imm * b) / d; }
        default: { const float d = bv * bv; return isv ? a / b : (bv * a - av * b) / d; }      /* DIV_LHS_RHS */
    }
}

__global__ void __launch_bounds__(256)
k_eval_normals_q(NormalArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    unsigned char* const myslot = smem + (size_t)wave * a.nslots * 256 + lane * 4;   /* slot s: myslot + s*256 */
    const int S = a.size;
    const int fps = S / 4;                                   /* footprints per side */
    /* a block is a 2x2 group of footprints = one 8x8 patch */
    const int patches = S / 8;
    const int fxi = (blockIdx.x % patches) * 2 + (wave & 1), fyi = (blockIdx.x / patches) * 2 + (wave >> 1);
    (void)fps;
    const int pix = lane >> 2, comp = lane & 3;
    const bool isv = comp == 3;
    const int px = fxi * 4 + (pix & 3), py = fyi * 4 + (pix >> 2);
    const int pxy = px + py * S;
    int pz = a.image[pxy];
    const bool filled = pz != 0;
    uint64_t todo = ballot(filled);
    if (todo == 0) return;
    if (pz < S - 1) pz += 1;                                   /* :1003-1005 */

    const float size_recip = 1.0f / (float)(unsigned)S;
    const float fx = ((px + 0.5f) * size_recip - 0.5f) * 2.0f;
    const float fy = ((py + 0.5f) * size_recip - 0.5f) * 2.0f;
    const float fz = ((pz + 0.5f) * size_recip - 0.5f) * 2.0f;
    const float fw = a.mat[3] * fx + a.mat[7] * fy + a.mat[11] * fz + a.mat[15];
    const float vx = (a.mat[0] * fx + a.mat[4] * fy + a.mat[8] * fz + a.mat[12]) / fw;
    const float vy = (a.mat[1] * fx + a.mat[5] * fy + a.mat[9] * fz + a.mat[13]) / fw;
    const float vz = (a.mat[2] * fx + a.mat[6] * fy + a.mat[10] * fz + a.mat[14]) / fw;

    /* deepest tile's tape (:1034-1066) */
    int my_tape = 0;
    if (filled) {
        const int t64 = S / 64;
        const int tile = px / 64 + (py / 64) * t64 + (pz / 64) * t64 * t64;
        const mpr_tile_node tn = a.tiles[tile];
        if (tn.next == -1) {
            my_tape = tn.tape;
        } else {
            const int subtile = tn.next * 64 + (px % 64) / 16 + ((py % 64) / 16) * 4 + ((pz % 64) / 16) * 16;
            const mpr_tile_node sn = a.subtiles[subtile];
            if (sn.next == -1) {
                my_tape = sn.tape;
            } else {
                const int micro = sn.next * 64 + (px % 16) / 4 + ((py % 16) / 4) * 4 + ((pz % 16) / 4) * 16;
                my_tape = a.microtiles[micro].tape;
            }
        }
    }

    const uint64_t* __restrict__ const tro = a.tape_ro;
    const uint64_t head0 = tro[0];
    const uint32_t sx = (head0 >> 8) & 0xFF, sy = (head0 >> 16) & 0xFF, sz = (head0 >> 24) & 0xFF;
    float result = 0.0f;
    long long words_total = 0, lane_clauses = 0;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int tape = __builtin_amdgcn_readlane(my_tape, leader);
        const bool mine = filled && my_tape == tape;
        const uint64_t grp = ballot(mine);
        todo &= ~grp;

        /* :1021-1031 — value first, then the unit partials (unused axes alias slot 0) */
        *reinterpret_cast<float*>(myslot + sx * 256) = isv ? vx : 0.0f;
        *reinterpret_cast<float*>(myslot + sy * 256) = isv ? vy : 0.0f;
        *reinterpret_cast<float*>(myslot + sz * 256) = isv ? vz : 0.0f;
        if (comp == 0) *reinterpret_cast<float*>(myslot + sx * 256) = 1.0f;
        if (comp == 1) *reinterpret_cast<float*>(myslot + sy * 256) = 1.0f;
        if (comp == 2) *reinterpret_cast<float*>(myslot + sz * 256) = 1.0f;

        int base = tape + 1;
        uint64_t blk = tro[base + lane];
        int j = 0, words = 0;
        uint32_t dlo = 0, dhi = 0;
        for (;;) {
            if (j == 64) {
                base += 64;
                blk = tro[base + lane];
                j = 0;
            }
            dlo = rdlane((uint32_t)blk, j);
            dhi = rdlane((uint32_t)(blk >> 32), j);
            ++words;
            const uint32_t op = dlo & 0xFF;
            if (op < 2) {
                if (op == 0) break;
                base = base + j + (int32_t)dhi + 1;      /* JUMP */
                blk = tro[base + lane];
                j = 0;
                continue;
            }
            ++j;
            const uint32_t vlo = to_vgpr(dlo);
            const float imm = mpr_u2f(to_vgpr(dhi));
            const uint32_t r8 = vlo >> 24;
            const float A = *reinterpret_cast<const float*>(myslot + ((vlo >> 8) & 0xFF00));
            const float Bs = *reinterpret_cast<const float*>(myslot + (r8 << 8));
            float* const outp = reinterpret_cast<float*>(myslot + (vlo & 0xFF00));
            const float av = quad_bcast(A, 3);
            /* rhs as a Deriv: a slot, or the constant (0, 0, 0, imm) of the immediate forms */
            const float B = r8 ? Bs : (isv ? imm : 0.0f);
            const float bv = r8 ? quad_bcast(Bs, 3) : imm;
            float out;
            if (op >= MPR_OP_ADD_LHS_IMM) {
                if (op <= MPR_OP_MUL_LHS_RHS) {
                    if (op <= MPR_OP_ADD_LHS_RHS) out = (r8 || isv) ? A + B : A;       /* a + float leaves the partials alone */
                    else if (op == MPR_OP_MUL_LHS_IMM) out = A * imm;
                    else out = isv ? A * B : A * bv + B * av;
                } else if (op <= MPR_OP_MAX_LHS_RHS) {
                    const bool take_a = (op <= MPR_OP_MIN_LHS_RHS) ? (av < bv) : (av >= bv);
                    out = take_a ? A : B;
                } else if (op <= MPR_OP_SUB_LHS_RHS) {
                    if (op == MPR_OP_SUB_LHS_IMM) out = isv ? A - imm : A;
                    else if (op == MPR_OP_SUB_IMM_RHS) out = isv ? imm - B : -B;
                    else out = A - B;
                } else if (op <= MPR_OP_DIV_LHS_RHS) {
                    out = deriv_rare_q(op, A, av, B, bv, imm, isv);
                } else {
                    out = (op == MPR_OP_COPY_LHS) ? A : B;          /* COPY_IMM: rhs == 0, B is (0,0,0,imm) */
                }
            } else if (op == MPR_OP_SQUARE_LHS) {
                out = isv ? A * A : A * av + A * av;                /* evaluated as lhs * lhs (:1081) */
            } else if (op == MPR_OP_NEG_LHS) {
                out = -A;
            } else if (op == MPR_OP_ABS_LHS) {
                out = (av < 0.0f) ? -A : A;
            } else {
                out = deriv_rare_q(op, A, av, B, bv, imm, isv);
            }
            *outp = out;
        }
        const float rr = *reinterpret_cast<const float*>(myslot + (dlo & 0xFF00));
        if (mine) result = rr;
        words_total += words;
        lane_clauses += (long long)(words - 1) * (__popcll(grp) / 4);
    }

    /* :1123-1131 */
    const float gx = quad_bcast(result, 0), gy = quad_bcast(result, 1), gz = quad_bcast(result, 2);
    const float norm = __builtin_sqrtf(gx * gx + gy * gy + gz * gz);
    const uint32_t u = f2u8((result / norm) * 127 + 128);
    const uint32_t ux = mpr_f2u(quad_bcast(mpr_u2f(u), 0)), uy = mpr_f2u(quad_bcast(mpr_u2f(u), 1)),
                   uz = mpr_f2u(quad_bcast(mpr_u2f(u), 2));
    if (filled && comp == 0) a.output[pxy] = (0xFFu << 24) | (uz << 16) | (uy << 8) | ux;
    const unsigned long long npx = (unsigned long long)(__popcll(ballot(filled)) / 4);
    if (a.counters && lane == 0) {
        atomicAdd((unsigned long long*)&a.counters[CNT_FWD], (unsigned long long)words_total);
        atomicAdd((unsigned long long*)&a.counters[CNT_FWD_NORM], (unsigned long long)words_total);
        atomicAdd((unsigned long long*)&a.counters[CNT_LANE], (unsigned long long)lane_clauses);
        atomicAdd((unsigned long long*)&a.counters[CNT_NORMAL_PX], npx);
    }
}

/* ---- launchers ---------------------------------------------------------------------- */
/* gfx950 offers 160 KiB of LDS per workgroup; anything above the 64 KiB default must be opted in */
template <typename K>
static void allow_big_lds(K kernel)
{
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
static void opt_in_once()
{
    static OncePerDevice once;
    once.run([] {
        allow_big_lds(k_eval_voxels<2>);
        allow_big_lds(k_eval_voxels<3>);
        allow_big_lds(k_eval_normals_q);
    });
}
size_t voxel_lds_bytes(int nslots) { return (size_t)nslots * 256 * 4; }
void launch_eval_voxels(hipStream_t s, int dim, const VoxelArgs& a)
{
    if (a.count <= 0) return;
    opt_in_once();
    const dim3 g((a.count + 3) / 4), b(256);
    const size_t lds = voxel_lds_bytes(a.nslots);
    if (dim == 3) hipLaunchKernelGGL(k_eval_voxels<3>, g, b, lds, s, a);
    else hipLaunchKernelGGL(k_eval_voxels<2>, g, b, lds, s, a);
}
size_t normals_lds_bytes(int nslots) { return (size_t)nslots * 1024; }   /* 4 waves x nslots x 64 lanes x 4 B */
void launch_eval_normals(hipStream_t s, const NormalArgs& a)
{
    opt_in_once();
    const int patches = a.size / 8;
    /* quad layout: 4 waves per 8x8 patch, 4-byte slots */
    hipLaunchKernelGGL(k_eval_normals_q, dim3(patches * patches), dim3(256), normals_lds_bytes(a.nslots), s, a);
}

}  // namespace mprk
