/*
 * kernels_voxel_pair_asm.hip — the float voxel pass for PAIRS of smallest tiles that share a tape.
 *
 * A 4^3 tile that made no min/max choice in the last interval stage keeps its parent's tape, so
 * among the 64 children of one 16^3 tile many walk the very same clauses (bear 1024^3: 54 % of
 * the voxel tiles, in groups of ~35).  The compaction of the last stage (k_compact_subdivide)
 * pairs such siblings up; here one wavefront evaluates both tiles of a pair at once: every lane
 * carries the same voxel of tile 0 and of tile 1, the slot file holds both values 256 bytes apart
 * (one ds_read2st64_b32 / ds_write2st64_b32 per operand), and the packed-FP32 instructions of
 * CDNA3/4 (v_pk_add_f32, v_pk_mul_f32; IEEE, same rounding and denormal mode as the scalar ones)
 * do add / sub / mul / square for both tiles in one issue.  Clause fetch, dispatch and address
 * arithmetic — most of what the single-tile interpreter (kernels_voxel_asm.hip; read that header
 * first) spends per clause — are paid once per pair.  Results are those of the single-tile walk,
 * bit for bit (tests: frames with MPR_VOXEL_PAIRS=0 / 1, and every opcode through this
 * interpreter on NaN / inf / subnormal operands).
 */
#include "asm_float_bodies.hpp"
#include "kernel_common.hpp"

namespace mprk {

__device__ __noinline__ float np_sin(float v) { return mpr_sinf(v); }
__device__ __noinline__ float np_cos(float v) { return mpr_cosf(v); }
__device__ __noinline__ float np_asin(float v) { return mpr_asinf(v); }
__device__ __noinline__ float np_acos(float v) { return mpr_acosf(v); }
__device__ __noinline__ float np_atan(float v) { return mpr_atanf(v); }
DEV float rare_unary_p(uint32_t op, float v)
{
    switch (op) {
        case MPR_OP_SIN_LHS: return np_sin(v);
        case MPR_OP_COS_LHS: return np_cos(v);
        case MPR_OP_ASIN_LHS: return np_asin(v);
        case MPR_OP_ACOS_LHS: return np_acos(v);
        case MPR_OP_ATAN_LHS: return np_atan(v);
        default: return mpr_u2f(0x7FC00000u);       /* not an opcode */
    }
}

/* Fixed registers (clobbers):
 *   s[80:81] handler address  s[82:83] table base  s[84:85] block address  s86 clause word  s87 immediate
 *   (s[86:87] doubles as the packed operand "immediate for both tiles": op_sel picks the high dword)
 *   s88 clause counter  s89 block base  s90 0x260  s96 0xff00  s[70:71] return address of the
 *   scalar routines, s[72:79] their entry points (div, sqrt, exp, log)
 *   v32 aA  v33 aB  v34 aO   v[44:45] A (tile 0, tile 1)  v[46:47] B  v[48:49] result (and previous result)
 *   v35 / v36 -> v37: argument(s) and result of the scalar routines, v38..v42 their temporaries, v50 scratch */
#define VP_DISPATCH                                    \
    "s_add_u32 s88, s88, 1\n"                          \
    "v_readlane_b32 s86, %[blo], s88\n"                \
    "s_and_b32 s80, s86, s96\n"                        \
    "s_add_u32 s80, s80, s82\n"                        \
    "s_addc_u32 s81, s83, 0\n"                         \
    "s_setpc_b64 s[80:81]\n"
#define VP_IMM "v_readlane_b32 s87, %[bhi], s88\n"
#define VP_AL "v_perm_b32 v32, s86, %[lb], %[selL]\n ds_read2st64_b32 v[44:45], v32 offset1:1\n"
#define VP_AR "v_perm_b32 v33, s86, %[lb], %[selR]\n ds_read2st64_b32 v[46:47], v33 offset1:1\n"
#define VP_AO "v_perm_b32 v34, s86, %[lb], %[selO]\n"
#define VP_W "s_waitcnt lgkmcnt(0)\n"
#define VP_END "ds_write2st64_b32 v34, v48, v49 offset1:1\n" VP_DISPATCH
#define VP_H(v, n) ".p2align 8\nL_q" #v "_" #n "_%=:\n"
#define VP_EXIT VP_IMM "s_branch L_exit_%=\n"
#define VP_CALL(pair) "s_swappc_b64 s[70:71], " pair "\n"
#define VP_IMMSEL " op_sel:[0,1] op_sel_hi:[1,1]"          /* src1 = s[86:87]: its high dword for both halves */
#define VP_NEGB " neg_lo:[0,1] neg_hi:[0,1]"
/* a scalar routine on both tiles: X0 / X1 argument registers, entry point `pair` */
#define VP_UNARY2(A0, A1, pair)                                                             \
    "v_mov_b32 v35, " A0 "\n" VP_CALL(pair) "v_mov_b32 v50, v37\n"                          \
    "v_mov_b32 v35, " A1 "\n" VP_CALL(pair) "v_mov_b32 v49, v37\n v_mov_b32 v48, v50\n"
#define VP_DIV2(N0, N1, D0, D1)                                                             \
    "v_mov_b32 v35, " N0 "\n v_mov_b32 v36, " D0 "\n" VP_CALL("s[72:73]") "v_mov_b32 v50, v37\n" \
    "v_mov_b32 v35, " N1 "\n v_mov_b32 v36, " D1 "\n" VP_CALL("s[72:73]") "v_mov_b32 v49, v37\n v_mov_b32 v48, v50\n"
/* min / max with canonicalised operands, per tile (there is no packed f32 min / max) */
#define VP_MM2(insn, A0, A1, B0, B1)                                                        \
    "v_max_f32 v35, " A0 ", " A0 "\n v_max_f32 v36, " B0 ", " B0 "\n " insn " v50, v35, v36\n" \
    "v_max_f32 v35, " A1 ", " A1 "\n v_max_f32 v36, " B1 ", " B1 "\n " insn " v49, v35, v36\n v_mov_b32 v48, v50\n"

/* LDL / LDR load lhs / rhs (or nothing when forwarded), A2 / A0 / A1 the register pair / halves the
 * lhs is then in (same for B), WL / WR / WLR s_waitcnt if lhs / rhs / either was loaded */
#define VP_TABLE(v, LDL, LDR, WL, WR, WLR, A2, A0, A1, B2, B0, B1)                                           \
    VP_H(v, 0) "s_branch L_exit_%=\n"                                                                        \
    VP_H(v, 1) VP_IMM "s_add_u32 s89, s89, s88\n s_add_u32 s89, s89, s87\n s_add_u32 s89, s89, 1\n s_branch L_load_%=\n" \
    VP_H(v, 2) LDL VP_AO WL "v_pk_mul_f32 v[48:49], " A2 ", " A2 "\n" VP_END                                  \
    VP_H(v, 3) LDL VP_AO WL VP_UNARY2(A0, A1, "s[74:75]") VP_END                                              \
    VP_H(v, 4) LDL VP_AO WL "v_xor_b32 v48, 0x80000000, " A0 "\n v_xor_b32 v49, 0x80000000, " A1 "\n" VP_END  \
    VP_H(v, 5) VP_EXIT VP_H(v, 6) VP_EXIT VP_H(v, 7) VP_EXIT VP_H(v, 8) VP_EXIT VP_H(v, 9) VP_EXIT           \
    VP_H(v, 10) LDL VP_AO WL VP_UNARY2(A0, A1, "s[76:77]") VP_END                                             \
    VP_H(v, 11) LDL VP_AO WL "v_and_b32 v48, 0x7fffffff, " A0 "\n v_and_b32 v49, 0x7fffffff, " A1 "\n" VP_END \
    VP_H(v, 12) LDL VP_AO WL VP_UNARY2(A0, A1, "s[78:79]") VP_END                                             \
    VP_H(v, 13) VP_IMM LDL VP_AO WL "s_nop 0\n v_pk_add_f32 v[48:49], " A2 ", s[86:87]" VP_IMMSEL "\n" VP_END \
    VP_H(v, 14) LDL LDR VP_AO WLR "v_pk_add_f32 v[48:49], " A2 ", " B2 "\n" VP_END                            \
    VP_H(v, 15) VP_IMM LDL VP_AO WL "s_nop 0\n v_pk_mul_f32 v[48:49], " A2 ", s[86:87]" VP_IMMSEL "\n" VP_END \
    VP_H(v, 16) LDL LDR VP_AO WLR "v_pk_mul_f32 v[48:49], " A2 ", " B2 "\n" VP_END                            \
    VP_H(v, 17) VP_IMM LDL VP_AO WL "s_nop 0\n v_mov_b32 v51, s87\n" VP_MM2("v_min_f32", A0, A1, "v51", "v51") VP_END \
    VP_H(v, 18) LDL LDR VP_AO WLR VP_MM2("v_min_f32", A0, A1, B0, B1) VP_END                                  \
    VP_H(v, 19) VP_IMM LDL VP_AO WL "s_nop 0\n v_mov_b32 v51, s87\n" VP_MM2("v_max_f32", A0, A1, "v51", "v51") VP_END \
    VP_H(v, 20) LDL LDR VP_AO WLR VP_MM2("v_max_f32", A0, A1, B0, B1) VP_END                                  \
    VP_H(v, 21) VP_IMM LDL VP_AO WL "s_nop 0\n v_pk_add_f32 v[48:49], " A2 ", s[86:87]" VP_IMMSEL VP_NEGB "\n" VP_END   /* lhs - imm */ \
    VP_H(v, 22) VP_IMM LDR VP_AO WR "s_nop 0\n v_pk_add_f32 v[48:49], s[86:87], " B2 " op_sel:[1,0] op_sel_hi:[1,1]" VP_NEGB "\n" VP_END /* imm - rhs */ \
    VP_H(v, 23) LDL LDR VP_AO WLR "v_pk_add_f32 v[48:49], " A2 ", " B2 VP_NEGB "\n" VP_END                    \
    VP_H(v, 24) VP_IMM LDL VP_AO WL "s_nop 0\n v_mov_b32 v51, s87\n" VP_DIV2(A0, A1, "v51", "v51") VP_END     \
    VP_H(v, 25) VP_IMM LDR VP_AO WR "s_nop 0\n v_mov_b32 v51, s87\n" VP_DIV2("v51", "v51", B0, B1) VP_END     \
    VP_H(v, 26) LDL LDR VP_AO WLR VP_DIV2(A0, A1, B0, B1) VP_END                                              \
    VP_H(v, 27) VP_IMM VP_AO "s_nop 0\n v_mov_b32 v48, s87\n v_mov_b32 v49, s87\n" VP_END                     \
    VP_H(v, 28) LDL VP_AO WL "v_mov_b32 v50, " A0 "\n v_mov_b32 v49, " A1 "\n v_mov_b32 v48, v50\n" VP_END    \
    VP_H(v, 29) LDR VP_AO WR "v_mov_b32 v50, " B0 "\n v_mov_b32 v49, " B1 "\n v_mov_b32 v48, v50\n" VP_END    \
    VP_H(v, 30) VP_EXIT                                                                                      \
    VP_H(v, 31) "s_add_u32 s89, s89, 63\n s_branch L_load_%=\n"

/* Walks the tape at tro[first] over the two-tile slot file at LDS offset 0 (slot s of lane l: tile 0 at
 * s * 512 + l * 4, tile 1 256 bytes further); returns the result slot.  nslots <= 128. */
DEV uint32_t interp_pair_asm(const uint64_t* __restrict__ tro, uint32_t first, unsigned char* smem, int lane)
{
    float* const plane = reinterpret_cast<float*>(smem);
    uint32_t blo = 0, bhi = 0;
    uint32_t base = first, sj = 0, dlo = 0, dhi = 0;
    const uint32_t lb = (uint32_t)(uintptr_t)smem + (uint32_t)lane * 4u;
    const uint32_t lane8 = (uint32_t)lane * 8u;
    const uint32_t selO = to_vgpr(0x0c0c0400u), selL = to_vgpr(0x0c0c0600u), selR = to_vgpr(0x0c0c0700u);
    const uint32_t tlo = (uint32_t)(uintptr_t)tro, thi = (uint32_t)((uintptr_t)tro >> 32);
    float p0 = 0.0f, p1 = 0.0f;
    uint32_t mode = 0;

    for (;;) {
        base = rdfirst(base);
        sj = rdfirst(sj);
        mode = rdfirst(mode);
        asm volatile(
            "s_mov_b32 s89, %[base]\n"
            "s_mov_b32 s88, %[sj]\n"
            "s_mov_b32 s90, 0x260\n"
            "s_mov_b32 s96, 0xff00\n"
            "v_mov_b32 v48, %[p0]\n"
            "v_mov_b32 v49, %[p1]\n"
            "s_getpc_b64 s[82:83]\n"
            "L_pc_%=:\n"
            "s_add_u32 s72, s82, L_div_%=-L_pc_%=\n s_addc_u32 s73, s83, 0\n"
            "s_add_u32 s74, s82, L_sqrt_%=-L_pc_%=\n s_addc_u32 s75, s83, 0\n"
            "s_add_u32 s76, s82, L_exp_%=-L_pc_%=\n s_addc_u32 s77, s83, 0\n"
            "s_add_u32 s78, s82, L_log_%=-L_pc_%=\n s_addc_u32 s79, s83, 0\n"
            "s_add_u32 s82, s82, L_q0_0_%=-L_pc_%=\n"
            "s_addc_u32 s83, s83, 0\n"
            "s_cmp_eq_u32 %[mode], 0\n"
            "s_cbranch_scc1 L_load_%=\n"
            VP_DISPATCH
            "L_load_%=:\n"
            "s_mov_b32 s84, s89\n"
            "s_mov_b32 s85, 0\n"
            "s_lshl_b64 s[84:85], s[84:85], 3\n"
            "s_add_u32 s84, s84, %[tlo]\n"
            "s_addc_u32 s85, s85, %[thi]\n"
            "global_load_dword %[blo], %[lane8], s[84:85]\n"
            "global_load_dword %[bhi], %[lane8], s[84:85] offset:4\n"
            "s_mov_b32 s88, -1\n"
            "v_mov_b32 v55, 0\n"
            "v_mov_b32 v57, 32\n"
            "v_mov_b32 v58, 64\n"
            "s_waitcnt vmcnt(0)\n"
            "v_bfe_u32 v54, %[blo], 8, 8\n"                 /* out slot */
            "v_and_b32 v52, 0xff, %[blo]\n"
            "v_min_u32 v52, 30, v52\n"
            "v_mov_b32_dpp v55, v54 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
            "v_bfe_u32 v56, %[blo], 16, 8\n"
            "v_lshrrev_b32 v53, 24, %[blo]\n"
            "v_cmp_eq_u32 s[92:93], v53, v55\n"
            "v_cmp_eq_u32 vcc, v56, v55\n"
            "v_cmp_ne_u32 s[94:95], 0, v55\n"
            "v_cndmask_b32 v56, 0, v58, s[92:93]\n"
            "v_cndmask_b32 v56, v56, v57, vcc\n"
            "v_cndmask_b32 v56, 0, v56, s[94:95]\n"
            "v_cmp_eq_u32 vcc, 0x1f8, %[lane8]\n"
            "v_mov_b32 v53, 31\n"
            "v_add_u32 v52, v52, v56\n"
            "v_cndmask_b32 v52, v52, v53, vcc\n"
            /* byte 0 = 2 * out slot, byte 1 = handler index, bytes 2, 3 = 2 * lhs, 2 * rhs (slots < 128) */
            "v_lshlrev_b32 v54, 1, v54\n"
            "v_lshl_or_b32 v52, v52, 8, v54\n"
            "v_and_b32 %[blo], 0xffff0000, %[blo]\n"
            "v_lshlrev_b32 %[blo], 1, %[blo]\n"
            "v_or_b32 %[blo], %[blo], v52\n"
            "s_nop 0\n"
            VP_DISPATCH
            VP_TABLE(0, VP_AL, VP_AR, VP_W, VP_W, VP_W, "v[44:45]", "v44", "v45", "v[46:47]", "v46", "v47")
            VP_TABLE(1, "", VP_AR, "", VP_W, VP_W, "v[48:49]", "v48", "v49", "v[46:47]", "v46", "v47")
            VP_TABLE(2, VP_AL, "", VP_W, "", VP_W, "v[44:45]", "v44", "v45", "v[48:49]", "v48", "v49")
            ".p2align 8\n"
            "L_div_%=:\n" MPR_ASM_DIV_BODY "s_setpc_b64 s[70:71]\n"
            "L_sqrt_%=:\n" MPR_ASM_SQRT_BODY "s_setpc_b64 s[70:71]\n"
            "L_exp_%=:\n" MPR_ASM_EXP_BODY "s_setpc_b64 s[70:71]\n"
            "L_log_%=:\n" MPR_ASM_LOG_BODY "s_setpc_b64 s[70:71]\n"
            "L_exit_%=:\n"
            "s_waitcnt lgkmcnt(0)\n"
            "s_mov_b32 %[dlo], s86\n"
            "s_mov_b32 %[dhi], s87\n"
            "s_mov_b32 %[base], s89\n"
            "s_mov_b32 %[sj], s88\n"
            : [blo] "+&v"(blo), [bhi] "+&v"(bhi), [base] "+&s"(base), [sj] "+&s"(sj), [dlo] "=&s"(dlo), [dhi] "=&s"(dhi)
            : [lb] "v"(lb), [selL] "v"(selL), [selR] "v"(selR), [selO] "v"(selO), [lane8] "v"(lane8),
              [tlo] "s"(tlo), [thi] "s"(thi), [mode] "s"(mode), [p0] "v"(p0), [p1] "v"(p1)
            : "memory", "vcc", "scc",
              "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84",
              "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96",
              "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v44", "v45", "v46", "v47",
              "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58");
        const uint32_t op = (dlo >> 8) & 31;
        if (op == 0) break;
        /* sin, cos, asin, acos, atan (or not an opcode): both tiles in C++ */
        const uint32_t l2 = (dlo >> 16) & 0xFF, o2 = dlo & 0xFF;                 /* 2 * slot */
        p0 = rare_unary_p(op, plane[l2 * 64 + lane]);
        p1 = rare_unary_p(op, plane[l2 * 64 + 64 + lane]);
        plane[o2 * 64 + lane] = p0;
        plane[o2 * 64 + 64 + lane] = p1;
        mode = 1;
    }
    return (dlo & 0xFF) >> 1;
}

/* {position of tile 0, position of tile 1, tape, -} */
__global__ void __launch_bounds__(64)
k_eval_voxel_pairs_asm(PairVoxelArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* const plane = reinterpret_cast<float*>(smem);
    const int lane = threadIdx.x;
    const int4 item = a.pairs[blockIdx.x];
    const int tape = __builtin_amdgcn_readfirstlane(item.z);
    const uint64_t* __restrict__ const tro = a.tape_ro;
    const int S = a.tps * 4;
    const int4_ sub = unpack(lane, 4);
    const float size_recip = 1.0f / (float)(unsigned)S;

    int px[2], py[2], pz[2];
    bool skip[2];
    float vx[2], vy[2], vz[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int4_ pos = unpack(__builtin_amdgcn_readfirstlane(k ? item.y : item.x), a.tps);
        px[k] = pos.x * 4 + sub.x;
        py[k] = pos.y * 4 + sub.y;
        pz[k] = pos.z * 4 + sub.z;
        /* reference :852-864: the thread owning (pz_low, pz_low + 2) leaves when image >= pz_low + 2 */
        const int pz_low = pos.z * 4 + (sub.z & 1);
        skip[k] = a.image[px[k] + py[k] * S] >= pz_low + 2;
        const float fx = ((px[k] + 0.5f) * size_recip - 0.5f) * 2.0f;
        const float fy = ((py[k] + 0.5f) * size_recip - 0.5f) * 2.0f;
        const float fz = ((pz[k] + 0.5f) * size_recip - 0.5f) * 2.0f;
        const float fw = a.mat[3] * fx + a.mat[7] * fy + a.mat[11] * fz + a.mat[15];
        vx[k] = (a.mat[0] * fx + a.mat[4] * fy + a.mat[8] * fz + a.mat[12]) / fw;
        vy[k] = (a.mat[1] * fx + a.mat[5] * fy + a.mat[9] * fz + a.mat[13]) / fw;
        vz[k] = (a.mat[2] * fx + a.mat[6] * fy + a.mat[10] * fz + a.mat[14]) / fw;
    }
    if (ballot(!(skip[0] && skip[1])) == 0) return;

    const uint64_t head0 = tro[0];
    const uint32_t sx = (head0 >> 8) & 0xFF, sy = (head0 >> 16) & 0xFF, sz = (head0 >> 24) & 0xFF;
    plane[sx * 128 + lane] = vx[0]; plane[sx * 128 + 64 + lane] = vx[1];
    plane[sy * 128 + lane] = vy[0]; plane[sy * 128 + 64 + lane] = vy[1];
    plane[sz * 128 + lane] = vz[0]; plane[sz * 128 + 64 + lane] = vz[1];

    const uint32_t rslot = interp_pair_asm(tro, (uint32_t)(tape + 1), smem, lane);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float res = plane[rslot * 128 + k * 64 + lane];
        if (!skip[k] && res < 0.0f) {
            int* p = &a.image[px[k] + py[k] * S];
            if (*p < pz[k]) atomicMax(p, pz[k]);
        }
    }
}

/* both tiles of one "pair" through the interpreter, one clause (tests): tape as in k_test_float_asm */
__global__ void __launch_bounds__(64)
k_test_float_pair_asm(const uint64_t* tape, int n, const float* a, const float* b, float* out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* const plane = reinterpret_cast<float*>(smem);
    const int lane = threadIdx.x;
    const int i0 = blockIdx.x * 128 + lane, i1 = i0 + 64;
    plane[1 * 128 + lane] = i0 < n ? a[i0] : 0.0f;
    plane[1 * 128 + 64 + lane] = i1 < n ? a[i1] : 0.0f;
    plane[2 * 128 + lane] = (i0 < n && b) ? b[i0] : 0.0f;
    plane[2 * 128 + 64 + lane] = (i1 < n && b) ? b[i1] : 0.0f;
    plane[3 * 128 + lane] = 0.0f;
    plane[3 * 128 + 64 + lane] = 0.0f;
    const uint32_t rslot = interp_pair_asm(tape, 1u, smem, lane);
    if (i0 < n) out[i0] = plane[rslot * 128 + lane];
    if (i1 < n) out[i1] = plane[rslot * 128 + 64 + lane];
}

void launch_eval_voxel_pairs_asm(hipStream_t s, const PairVoxelArgs& a)
{
    if (a.count <= 0) return;
    hipLaunchKernelGGL(k_eval_voxel_pairs_asm, dim3(a.count), dim3(64), (size_t)a.nslots * 512, s, a);
}
void launch_test_float_pair_asm(hipStream_t s, const uint64_t* tape, int n, const float* a, const float* b, float* out)
{
    hipLaunchKernelGGL(k_test_float_pair_asm, dim3((n + 127) / 128), dim3(64), 8 * 512, s, tape, n, a, b, out);
}

}  // namespace mprk
