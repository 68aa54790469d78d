/*
 * asm_float_bodies.hpp — round-to-nearest float routines shared by the assembly interpreters of
 * the float pass (kernels_voxel_asm.hip) and of the normals pass (kernels_normals_asm.hip), as
 * inline-asm text.  Register convention: argument v35 (division: v35 / v36), result v37,
 * temporaries v38..v42, s91, s[92:95], vcc; s90 must hold 0x260 (class mask of the square root).
 * Each is, instruction for instruction, what the compiler makes of the C++ definition
 * (IEEE division; mpr_expf / mpr_logf of include/mpr_fmath.h; the square root see below), so the
 * results are bit-identical to the compiled interpreters and to the oracle.
 *
 * Square root, exp and log come in two pieces: MPR_ASM_x_BODY is the path every lane of an ordinary frame takes, straight
 * through, without a taken branch (a taken branch costs a wavefront about 20 cycles, as much as five dependent VALU
 * instructions: scripts/ubench/branch_cost.hip); MPR_ASM_x_TAIL holds the rare cases (tiny or special operands), entered by
 * a branch from the body and left by a branch back to its end.  Put the tail behind the jump that ends the routine.
 */
#pragma once

#define MPR_ASM_DIV_BODY \
    "v_div_scale_f32 v38, s[92:93], v36, v36, v35\n" \
    "v_rcp_f32 v39, v38\n" \
    "v_div_scale_f32 v40, vcc, v35, v36, v35\n" \
    "v_fma_f32 v41, -v38, v39, 1.0\n" \
    "v_fmac_f32 v39, v41, v39\n" \
    "v_mul_f32 v41, v40, v39\n" \
    "v_fma_f32 v42, -v38, v41, v40\n" \
    "v_fmac_f32 v41, v42, v39\n" \
    "v_fma_f32 v38, -v38, v41, v40\n" \
    "v_div_fmas_f32 v38, v38, v39, v41\n" \
    "v_div_fixup_f32 v37, v38, v36, v35\n"

/* Fast path when every lane's operand is a positive normal number >= 2^-96 (one unsigned compare of the bits): no scaling of
 * tiny operands, no special values.  y = v_rsq_f32(x) is 1 / sqrt(x) to about an ulp; g = x y and h = y / 2 get one coupled
 * Newton step (r = 1/2 - h g; g += g r; h += h r), and the last step adds the exact residual of g times h: g + (x - g g) h.
 * That sum, before its one rounding, is within 2^-50 (relative) of sqrt(x), and no square root of a float lies that close to the
 * middle between two floats (the closest miss it by more than 2^-49): the rounded sum IS the correctly rounded root — the value
 * sqrtf / __builtin_sqrtf return and the oracle computes.  Checked on the device for every float (mpr_test_sqrt_all:
 * tests/test_gpu_primitives.py).  7 full-rate instructions and the quarter-rate v_rsq_f32, where v_sqrt_f32 + two residual
 * tests + selects took 5 full-rate, 5 half-rate and the quarter-rate one: a fifth of the float pass's issue time for bear. */
#define MPR_ASM_SQRT_BODY \
    "v_add_u32 v38, 0xf0800000, v35\n"                 /* bits - 0x0f800000 */ \
    "v_cmp_gt_u32 vcc, 0x70000000, v38\n"              /* < 0x7f800000 - 0x0f800000: positive, normal, finite, not tiny */ \
    "s_cmp_eq_u64 vcc, exec\n" \
    "s_cbranch_scc0 L_sqrtslow_%=\n" \
    "v_rsq_f32 v39, v35\n" \
    "s_nop 0\n" \
    "v_mul_f32 v40, v35, v39\n"                        /* g */ \
    "v_mul_f32 v39, 0.5, v39\n"                        /* h */ \
    "v_fma_f32 v41, -v39, v40, 0.5\n"                  /* r */ \
    "v_fmac_f32 v40, v40, v41\n" \
    "v_fmac_f32 v39, v39, v41\n" \
    "v_fma_f32 v41, -v40, v40, v35\n"                  /* x - g g, exactly */ \
    "v_fma_f32 v37, v41, v39, v40\n" \
    "L_sqrtdone_%=:\n"
#define MPR_ASM_SQRT_TAIL \
    "L_sqrtslow_%=:\n" \
    "v_mul_f32 v38, 0x4f800000, v35\n" \
    "v_cmp_gt_f32 vcc, 0xf800000, v35\n" \
    "s_nop 1\n" \
    "v_cndmask_b32 v38, v35, v38, vcc\n" \
    "v_sqrt_f32 v39, v38\n" \
    "s_nop 0\n" \
    "v_add_u32 v40, -1, v39\n" \
    "v_fma_f32 v41, -v40, v39, v38\n" \
    "v_cmp_ge_f32 s[92:93], 0, v41\n" \
    "v_add_u32 v41, 1, v39\n" \
    "s_nop 0\n" \
    "v_cndmask_b32 v40, v39, v40, s[92:93]\n" \
    "v_fma_f32 v39, -v41, v39, v38\n" \
    "v_cmp_lt_f32 s[92:93], 0, v39\n" \
    "s_nop 1\n" \
    "v_cndmask_b32 v39, v40, v41, s[92:93]\n" \
    "v_mul_f32 v40, 0x37800000, v39\n" \
    "v_cndmask_b32 v39, v39, v40, vcc\n" \
    "v_cmp_class_f32 vcc, v38, s90\n" \
    "s_nop 1\n" \
    "v_cndmask_b32 v37, v39, v38, vcc\n" \
    "s_branch L_sqrtdone_%=\n"

/* mpr_expf: t = x log2(e); kf = (t + 1.5 * 2^23) - 1.5 * 2^23 (t rounded to the nearest integer, ties to even: the C++
 * definition word for word); r = x - kf ln2 in two steps; p = the polynomial; p * 2^k.  For |x| <= 87 the result is a normal number,
 * so 2^k is an addition to p's exponent field — and k sits in the low bits of t + 1.5 * 2^23 (0x4b400000 + k), whose upper bits a
 * shift by 23 pushes out: one v_lshl_add_u32 where v_cvt_i32_f32 + v_ldexp_f32 (both half rate) stood.  The tail redoes it by the
 * book for everything else. */
#define MPR_ASM_EXP_BODY \
    "v_mul_f32 v38, 0x3fb8aa3b, v35\n" \
    "v_add_f32 v42, 0x4b400000, v38\n"              /* t + 1.5 * 2^23 */ \
    "v_add_f32 v38, 0xcb400000, v42\n"              /* kf */ \
    "v_fmamk_f32 v39, v38, 0xbf318000, v35\n" \
    "v_fmamk_f32 v39, v38, 0x395e8083, v39\n"       /* r */ \
    "v_mov_b32 v40, 0x3ab743ce\n" \
    "v_fmac_f32 v40, 0x39506967, v39\n" \
    "v_fmaak_f32 v40, v40, v39, 0x3c088908\n" \
    "v_fmaak_f32 v40, v40, v39, 0x3d2aa9c1\n" \
    "v_fmaak_f32 v40, v40, v39, 0x3e2aaaaa\n" \
    "v_mul_f32 v41, v39, v39\n" \
    "v_fma_f32 v40, v40, v39, 0.5\n" \
    "v_fmac_f32 v39, v40, v41\n" \
    "v_add_f32 v39, 1.0, v39\n"                     /* p */ \
    "v_lshl_add_u32 v37, v42, 23, v39\n"            /* p * 2^k for a normal result */ \
    /* |x| <= 87 in every lane (NaN counts as not): none of the special cases applies */ \
    "s_mov_b32 s91, 0x42ae0000\n" \
    "v_cmp_nle_f32 vcc, |v35|, s91\n" \
    "s_cbranch_vccnz L_expspecial_%=\n" \
    "L_expdone_%=:\n"
#define MPR_ASM_EXP_TAIL \
    "L_expspecial_%=:\n" \
    "v_cvt_i32_f32 v38, v38\n"                      /* k */ \
    "s_mov_b32 s91, 0x42b17218\n" \
    /* p * 2^k: the C++ definition multiplies by 2^(k/2) and 2^(k - k/2), the first product exact, \
     * i.e. one rounding of p * 2^k — which is what v_ldexp_f32 does in one instruction */ \
    "v_ldexp_f32 v37, v39, v38\n" \
    "v_cmp_ngt_f32 vcc, 0xc2cff5c3, v35\n"          /* !(x < -103.98) */ \
    "v_cmp_nlt_f32 s[92:93], s91, v35\n"            /* !(x > 88.72284) */ \
    "v_cmp_o_f32 s[94:95], v35, v35\n" \
    "v_mov_b32 v40, 0x7f800000\n" \
    "v_cndmask_b32 v37, 0, v37, vcc\n" \
    "v_cndmask_b32 v37, v40, v37, s[92:93]\n" \
    "v_cndmask_b32 v37, v35, v37, s[94:95]\n" \
    "s_branch L_expdone_%=\n"

#define MPR_ASM_LOG_BODY \
    /* every lane a positive normal number: no subnormal scaling, none of the special results */ \
    "s_movk_i32 s91, 0x100\n" \
    "v_mov_b32 v40, 0xffffff82\n" \
    "v_cmp_class_f32 vcc, v35, s91\n" \
    "v_mov_b32 v38, v35\n" \
    "s_cmp_eq_u64 vcc, exec\n" \
    "s_cselect_b32 s91, 1, 0\n" \
    "s_cbranch_scc0 L_logscale_%=\n" \
    "L_logmain_%=:\n" \
    "v_lshrrev_b32 v39, 23, v38\n" \
    "v_add_u32 v39, v39, v40\n"                     /* e */ \
    "v_and_b32 v38, 0x7fffff, v38\n" \
    "v_or_b32 v38, 0x3f000000, v38\n"               /* m in [0.5, 1) */ \
    "v_cmp_gt_f32 vcc, 0x3f3504f3, v38\n" \
    "v_bfrev_b32 v41, 1\n" \
    "s_nop 0\n" \
    "v_subbrev_co_u32 v39, s[92:93], 0, v39, vcc\n" /* e -= (m < sqrt(1/2)) */ \
    "v_cndmask_b32 v41, v41, v38, vcc\n" \
    "v_add_f32 v38, v41, v38\n"                     /* m + m, or m + (-0) */ \
    "v_add_f32 v38, -1.0, v38\n" \
    "v_mov_b32 v40, 0xbdebd1b8\n" \
    "v_fmac_f32 v40, 0x3d9021bb, v38\n" \
    "v_fmaak_f32 v40, v40, v38, 0x3def251a\n" \
    "v_fmaak_f32 v40, v40, v38, 0xbdfe5d4f\n" \
    "v_fmaak_f32 v40, v40, v38, 0x3e11e9bf\n" \
    "v_fmaak_f32 v40, v40, v38, 0xbe2aae50\n" \
    "v_fmaak_f32 v40, v40, v38, 0x3e4cceac\n" \
    "v_fmaak_f32 v40, v40, v38, 0xbe7ffffc\n" \
    "v_cvt_f32_i32 v39, v39\n"                      /* fe */ \
    "v_fmaak_f32 v40, v40, v38, 0x3eaaaaaa\n" \
    "v_mul_f32 v41, v38, v38\n"                     /* z */ \
    "v_mul_f32 v40, v40, v38\n" \
    "v_mul_f32 v40, v40, v41\n"                     /* (y * m) * z */ \
    "v_fmamk_f32 v40, v39, 0xb95e8083, v40\n" \
    "v_fmac_f32 v40, -0.5, v41\n" \
    "v_add_f32 v40, v38, v40\n" \
    "v_fmamk_f32 v37, v39, 0x3f318000, v40\n" \
    "s_cmp_eq_u32 s91, 1\n" \
    "s_cbranch_scc0 L_logspecial_%=\n" \
    "L_logdone_%=:\n"
#define MPR_ASM_LOG_TAIL \
    "L_logscale_%=:\n" \
    "v_mul_f32 v38, 0x4b000000, v35\n" \
    "v_cmp_gt_u32 vcc, 0x800000, v35\n"             /* subnormal: scale by 2^23 */ \
    "v_mov_b32 v41, 0xffffff6b\n" \
    "s_nop 0\n" \
    "v_cndmask_b32 v38, v35, v38, vcc\n" \
    "v_cndmask_b32 v40, v40, v41, vcc\n" \
    "s_branch L_logmain_%=\n" \
    "L_logspecial_%=:\n" \
    "v_cmp_ne_u32 vcc, 0x7f800000, v35\n"           /* log(+inf) = +inf */ \
    "v_mov_b32 v40, 0xff800000\n" \
    "v_mov_b32 v41, 0x7fc00000\n" \
    "v_cndmask_b32 v37, v35, v37, vcc\n" \
    "v_cmp_eq_f32 vcc, 0, v35\n"                    /* log(+-0) = -inf */ \
    "s_nop 1\n" \
    "v_cndmask_b32 v37, v37, v40, vcc\n" \
    "v_cmp_gt_f32 vcc, 0, v35\n"                    /* log(x < 0) = NaN */ \
    "s_nop 1\n" \
    "v_cndmask_b32 v37, v37, v41, vcc\n" \
    "v_cmp_u_f32 vcc, v35, v35\n"                   /* NaN in, the same NaN out */ \
    "s_nop 1\n" \
    "v_cndmask_b32 v37, v37, v35, vcc\n" \
    "s_branch L_logdone_%=\n"

/* v37 = mpr_sinf(v35), v36 = mpr_cosf(v35) (include/mpr_fmath.h): double-precision Cody-Waite
 * reduction by pi/2, the two Cephes polynomials, quadrant by selects instead of the switch.
 * (long long)j & 3 is taken as (int)j & 3: |x| < 2^31 there, so j fits 32 bits.  Temporaries
 * v[38:41] (double), v42..v47, s[40:49], vcc. */
#define MPR_ASM_SINCOS_BODY \
    "s_mov_b32 s40, 0x6dc9c883\n s_mov_b32 s41, 0x3fe45f30\n"      /* 2 / pi */ \
    "s_mov_b32 s42, 0\n s_mov_b32 s43, 0x43380000\n"               /* 1.5 * 2^52 */ \
    "s_mov_b32 s44, 0x54442d18\n s_mov_b32 s45, 0xbff921fb\n"      /* -pi/2 (high part) */ \
    "s_mov_b32 s46, 0x33145c07\n s_mov_b32 s47, 0xbc91a626\n"      /* -pi/2 (low part) */ \
    "v_cvt_f64_f32 v[38:39], v35\n" \
    "v_mul_f64 v[40:41], v[38:39], s[40:41]\n" \
    "v_add_f64 v[40:41], v[40:41], s[42:43]\n" \
    "v_add_f64 v[40:41], v[40:41], -s[42:43]\n"                    /* j = round(x * 2 / pi) */ \
    "v_fma_f64 v[38:39], v[40:41], s[44:45], v[38:39]\n" \
    "v_fma_f64 v[38:39], v[40:41], s[46:47], v[38:39]\n"           /* r */ \
    "v_cvt_i32_f64 v42, v[40:41]\n" \
    "v_cvt_f32_f64 v43, v[38:39]\n" \
    "v_and_b32 v42, 3, v42\n"                                       /* quadrant */ \
    "v_mul_f32 v44, v43, v43\n"                                     /* z */ \
    "v_mov_b32 v45, 0x3c08839e\n" \
    "v_fmac_f32 v45, 0xb94ca1f9, v44\n" \
    "v_fmaak_f32 v45, v45, v44, 0xbe2aaaa3\n" \
    "v_mul_f32 v45, v44, v45\n" \
    "v_mul_f32 v45, v45, v43\n" \
    "v_add_f32 v45, v45, v43\n"                                     /* sin polynomial */ \
    "v_mov_b32 v46, 0xbab6061a\n" \
    "v_fmac_f32 v46, 0x37ccf5ce, v44\n" \
    "v_fmaak_f32 v46, v46, v44, 0x3d2aaaa5\n" \
    "v_mul_f32 v46, v44, v46\n" \
    "v_mul_f32 v46, v44, v46\n" \
    "v_fmac_f32 v46, -0.5, v44\n" \
    "v_add_f32 v46, 1.0, v46\n"                                     /* cos polynomial */ \
    "v_and_b32 v47, 1, v42\n" \
    "v_cmp_eq_u32 vcc, 1, v47\n" \
    "v_lshlrev_b32 v47, 30, v42\n"                                  /* quadrants 2, 3: sin negated */ \
    "v_add_u32 v42, 1, v42\n" \
    "v_cndmask_b32 v37, v45, v46, vcc\n"                            /* sin: odd quadrant -> cos polynomial */ \
    "v_cndmask_b32 v36, v46, v45, vcc\n"                            /* cos: odd quadrant -> sin polynomial */ \
    "v_and_b32 v47, 0x80000000, v47\n" \
    "v_lshlrev_b32 v42, 30, v42\n"                                  /* quadrants 1, 2: cos negated */ \
    "v_xor_b32 v37, v37, v47\n" \
    "v_and_b32 v42, 0x80000000, v42\n" \
    "v_and_b32 v47, 0x7fffffff, v35\n" \
    "v_xor_b32 v36, v36, v42\n" \
    "v_cmp_le_u32 vcc, 0x4f000000, v47\n"                           /* |x| >= 2^31: sin 0, cos 1 */ \
    "v_mov_b32 v44, 0x7fc00000\n" \
    "s_nop 0\n" \
    "v_cndmask_b32 v37, v37, 0, vcc\n" \
    "v_cndmask_b32 v36, v36, 1.0, vcc\n" \
    "v_cmp_le_u32 vcc, 0x7f800000, v47\n"                           /* inf / NaN: NaN */ \
    "s_nop 1\n" \
    "v_cndmask_b32 v37, v37, v44, vcc\n" \
    "v_cndmask_b32 v36, v36, v44, vcc\n"
