"""Interval arithmetic against EXACT arithmetic, and float functions against mpmath.

Nothing here uses the oracle's directed-rounding code or the product's: the expected bounds are
computed with Python's rational numbers (fractions.Fraction, exact) and rounded to float32 towards
-inf / +inf by comparison.  What is asserted is what the reference's intrinsics guarantee
(__fadd_rd / __fmul_ru / __fdiv_rd / __fsqrt_ru ..., inc/gpu_interval.hpp:71-203, :284-304):

  * tightness: for + - * / sqrt (and square, negation, abs) of finite, ordered intervals the result is
    the TIGHTEST float32 interval that contains every exact result — lower = RD(exact min),
    upper = RU(exact max) — all the way down to subnormal operands and results;
  * soundness: for asin / acos / atan / exp / log (double libm + directed conversion, :306-390) the
    exact image of sampled points (mpmath, 60 digits) lies inside [lower, upper];
  * the float pass's exp / log / sin / cos / asin / acos / atan (compiled routines and all six handler
    tables of the assembly interpreter) are within 2 ulp (acos near 1: 4) of mpmath on the GPU's outputs.

The same checks run on the oracle (CPU, everywhere) and on the GPU primitives (`gpu` marker).
Documented exceptions, kept from the reference on purpose and therefore excluded: cos / sin return
[-1, 1] (sound, not tight); log of an interval touching zero or below returns lower bound 0
(inc/gpu_interval.hpp:385-386: unsound); a divisor interval containing zero returns [-inf, inf].
"""
from fractions import Fraction

import numpy as np
import pytest

F32_MAX = Fraction(float(np.finfo(np.float32).max))


def f32(x):
    return np.float32(x)


def round_down(q):
    """largest float32 <= q (q: Fraction); -inf below -FLT_MAX"""
    if q > F32_MAX:
        return f32(np.finfo(np.float32).max)
    if q < -F32_MAX:
        return f32(-np.inf)
    c = f32(float(q))                       # within one float32 ulp of q
    if not np.isfinite(c):
        c = f32(np.copysign(np.finfo(np.float32).max, float(q)))
    while Fraction(float(c)) > q:
        c = np.nextafter(c, f32(-np.inf), dtype=np.float32)
    while True:
        up = np.nextafter(c, f32(np.inf), dtype=np.float32)
        if np.isfinite(up) and Fraction(float(up)) <= q:
            c = up
        else:
            return c


def round_up(q):
    return -round_down(-q)


def exact_sqrt_bounds(x):
    """(RD(sqrt x), RU(sqrt x)) for a float32 x >= 0, by exact comparison of squares"""
    if x == 0:
        return f32(x), f32(x)
    X = Fraction(float(x))
    c = f32(np.sqrt(np.float64(x)))
    while Fraction(float(c)) ** 2 > X:
        c = np.nextafter(c, f32(0), dtype=np.float32)
    while Fraction(float(np.nextafter(c, f32(np.inf), dtype=np.float32))) ** 2 <= X:
        c = np.nextafter(c, f32(np.inf), dtype=np.float32)
    lo = c
    hi = c if Fraction(float(c)) ** 2 == X else np.nextafter(c, f32(np.inf), dtype=np.float32)
    return lo, hi


def gen_intervals(rng, n, scale):
    """finite ordered float32 intervals; scale = "unit", "wide" (2^+-60) or "tiny" (around 2^-120..2^-149)"""
    if scale == "unit":
        a, b = rng.uniform(-3, 3, n), rng.uniform(-3, 3, n)
    elif scale == "wide":
        a = rng.standard_normal(n) * np.exp2(rng.uniform(-60, 60, n))
        b = rng.standard_normal(n) * np.exp2(rng.uniform(-60, 60, n))
    else:
        a = rng.standard_normal(n) * np.exp2(rng.uniform(-149, -100, n))
        b = rng.standard_normal(n) * np.exp2(rng.uniform(-149, -100, n))
    a, b = a.astype(np.float32), b.astype(np.float32)
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    hi = np.where(rng.random(n) < 0.15, lo, hi)          # some point intervals
    zero = rng.random(n) < 0.05
    lo = np.where(zero & (lo < 0), lo, np.where(zero, np.float32(0), lo)).astype(np.float32)
    return lo, np.maximum(lo, hi).astype(np.float32)


def expected(opname, alo, ahi, blo, bhi, imm):
    """tightest float32 interval containing the exact results, or None where the reference does not promise one"""
    A = (Fraction(float(alo)), Fraction(float(ahi)))
    B = (Fraction(float(blo)), Fraction(float(bhi)))
    I = Fraction(float(np.float32(imm)))
    if opname == "ADD_LHS_RHS":
        return round_down(A[0] + B[0]), round_up(A[1] + B[1])
    if opname == "ADD_LHS_IMM":
        return round_down(A[0] + I), round_up(A[1] + I)
    if opname == "SUB_LHS_RHS":
        return round_down(A[0] - B[1]), round_up(A[1] - B[0])
    if opname == "SUB_LHS_IMM":
        return round_down(A[0] - I), round_up(A[1] - I)
    if opname == "SUB_IMM_RHS":
        return round_down(I - B[1]), round_up(I - B[0])
    if opname in ("MUL_LHS_RHS", "MUL_LHS_IMM"):
        Y = B if opname == "MUL_LHS_RHS" else (I, I)
        p = [x * y for x in A for y in Y]
        return round_down(min(p)), round_up(max(p))
    if opname in ("DIV_LHS_RHS", "DIV_LHS_IMM", "DIV_IMM_RHS"):
        X, Y = (A, B) if opname == "DIV_LHS_RHS" else ((A, (I, I)) if opname == "DIV_LHS_IMM" else ((I, I), B))
        if Y[0] <= 0 <= Y[1]:
            return np.float32(-np.inf), np.float32(np.inf)          # inc/gpu_interval.hpp:163-164, :197-199
        p = [x / y for x in X for y in Y]
        return round_down(min(p)), round_up(max(p))
    if opname == "SQUARE_LHS":
        lo = 0 if A[0] <= 0 <= A[1] else min(A[0] ** 2, A[1] ** 2)
        return round_down(Fraction(lo)), round_up(max(A[0] ** 2, A[1] ** 2))
    if opname == "SQRT_LHS":
        if ahi < 0:
            return None
        lo = np.float32(0) if alo <= 0 else exact_sqrt_bounds(alo)[0]
        return lo, exact_sqrt_bounds(ahi)[1]
    if opname == "NEG_LHS":
        return np.float32(-ahi), np.float32(-alo)
    if opname == "ABS_LHS":
        if alo >= 0:
            return alo, ahi
        if ahi < 0:
            return np.float32(-ahi), np.float32(-alo)
        return np.float32(0), max(np.float32(-alo), ahi)
    raise KeyError(opname)


TIGHT_OPS = ["ADD_LHS_RHS", "ADD_LHS_IMM", "SUB_LHS_RHS", "SUB_LHS_IMM", "SUB_IMM_RHS", "MUL_LHS_RHS", "MUL_LHS_IMM",
             "DIV_LHS_RHS", "DIV_LHS_IMM", "DIV_IMM_RHS", "SQUARE_LHS", "SQRT_LHS", "NEG_LHS", "ABS_LHS"]
N_TIGHT = 3000


def check_tight(run, mpr, opname, scale, seed):
    rng = np.random.default_rng(seed)
    alo, ahi = gen_intervals(rng, N_TIGHT, scale)
    blo, bhi = gen_intervals(rng, N_TIGHT, scale if scale != "tiny" else ("tiny" if seed & 1 else "unit"))
    imm = {"unit": 0.75, "wide": -3.0e-7, "tiny": 1.5e-38}[scale]
    lo, hi = run(mpr.OP[opname], alo, ahi, blo, bhi, imm)
    checked = 0
    for i in range(N_TIGHT):
        want = expected(opname, alo[i], ahi[i], blo[i], bhi[i], imm)
        if want is None:
            continue
        checked += 1
        ok = (lo[i] == want[0]) and (hi[i] == want[1])
        assert ok, "%s %s: [%r, %r] op [%r, %r] (imm %r) gave [%r, %r], the tightest interval is [%r, %r]" % (
            opname, scale, alo[i], ahi[i], blo[i], bhi[i], imm, lo[i], hi[i], want[0], want[1])
    assert checked > N_TIGHT // 2


@pytest.mark.parametrize("scale", ["unit", "wide", "tiny"])
@pytest.mark.parametrize("opname", TIGHT_OPS)
def test_oracle_interval_ops_are_the_tightest_bounds(mpr, orc, opname, scale):
    def run(op, alo, ahi, blo, bhi, imm):
        lo, hi, _ = orc.interval_op(op, alo, ahi, blo, bhi, imm)
        return lo, hi
    check_tight(run, mpr, opname, scale, seed=len(opname) * 7 + len(scale))


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["compiled", "asm", "asm-lhs-forwarded"])
@pytest.mark.parametrize("scale", ["unit", "wide", "tiny"])
@pytest.mark.parametrize("opname", TIGHT_OPS)
def test_gpu_interval_ops_are_the_tightest_bounds(mpr, opname, scale, path):
    kw = {"compiled": {}, "asm": dict(asm=True), "asm-lhs-forwarded": dict(asm=True, variant=1)}[path]

    def run(op, alo, ahi, blo, bhi, imm):
        lo, hi, _ = mpr.dev_interval_op(op, alo, ahi, blo, bhi, imm, **kw)
        return lo, hi
    check_tight(run, mpr, opname, scale, seed=len(opname) * 7 + len(scale))


# ---------------------------------------------------------------------------------------------
SOUND_OPS = {"ASIN_LHS": ("asin", -1, 1), "ACOS_LHS": ("acos", -1, 1), "ATAN_LHS": ("atan", -50, 50),
             "EXP_LHS": ("exp", -80, 80), "LOG_LHS": ("log", 1e-30, 1e30), "SIN_LHS": ("sin", -20, 20), "COS_LHS": ("cos", -20, 20)}
N_SOUND = 400


def check_sound(run, mpr, opname):
    import mpmath
    mpmath.mp.dps = 60
    fn, lo_dom, hi_dom = SOUND_OPS[opname]
    rng = np.random.default_rng(len(opname))
    if opname == "LOG_LHS":
        a = np.exp(rng.uniform(np.log(lo_dom), np.log(hi_dom), N_SOUND))
        w = a * rng.uniform(0, 0.5, N_SOUND)
    else:
        a = rng.uniform(lo_dom, hi_dom, N_SOUND)
        w = rng.uniform(0, 0.3, N_SOUND) * (hi_dom - lo_dom) * rng.choice([1e-6, 1e-3, 1.0], N_SOUND)
    alo = a.astype(np.float32)
    ahi = np.minimum(a + w, hi_dom).astype(np.float32)
    ahi = np.maximum(alo, ahi)
    z = np.zeros(N_SOUND, np.float32)
    lo, hi = run(mpr.OP[opname], alo, ahi, z, z, 0.0)
    f = getattr(mpmath, fn)
    for i in range(N_SOUND):
        L, H = mpmath.mpf(float(lo[i])), mpmath.mpf(float(hi[i]))
        assert lo[i] <= hi[i]
        for t in (0.0, 0.37, 1.0):
            x = mpmath.mpf(float(alo[i])) * (1 - t) + mpmath.mpf(float(ahi[i])) * t
            y = f(x)
            assert L <= y <= H, "%s([%r, %r]) = [%r, %r] does not contain %s(%s) = %s" % (
                fn, alo[i], ahi[i], lo[i], hi[i], fn, x, y)
        if fn not in ("sin", "cos"):
            # and it is not needlessly wide: a few float32 ulps around the exact end points
            ends = sorted([f(mpmath.mpf(float(alo[i]))), f(mpmath.mpf(float(ahi[i])))])
            for got, exact in ((lo[i], ends[0]), (hi[i], ends[1])):
                ulp = float(np.spacing(np.float32(abs(float(exact))))) or 1e-45
                assert abs(float(got) - float(exact)) <= 2 * ulp, (fn, alo[i], ahi[i], got, float(exact))


@pytest.mark.parametrize("opname", list(SOUND_OPS))
def test_oracle_transcendental_intervals_are_sound(mpr, orc, opname):
    def run(op, alo, ahi, blo, bhi, imm):
        lo, hi, _ = orc.interval_op(op, alo, ahi, blo, bhi, imm)
        return lo, hi
    check_sound(run, mpr, opname)


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["compiled", "asm"])
@pytest.mark.parametrize("opname", list(SOUND_OPS))
def test_gpu_transcendental_intervals_are_sound(mpr, opname, path):
    kw = {"compiled": {}, "asm": dict(asm=True)}[path]

    def run(op, alo, ahi, blo, bhi, imm):
        lo, hi, _ = mpr.dev_interval_op(op, alo, ahi, blo, bhi, imm, **kw)
        return lo, hi
    check_sound(run, mpr, opname)


# ---------------------------------------------------------------------------------------------
FLOAT_FNS = {"SIN_LHS": ("sin", lambda r, n: r.uniform(-100, 100, n)), "COS_LHS": ("cos", lambda r, n: r.uniform(-100, 100, n)),
             "ASIN_LHS": ("asin", lambda r, n: r.uniform(-1, 1, n)), "ACOS_LHS": ("acos", lambda r, n: r.uniform(-1, 1, n)),
             "ATAN_LHS": ("atan", lambda r, n: np.concatenate([r.uniform(-10, 10, n // 2), r.standard_cauchy(n - n // 2) * 100])),
             "EXP_LHS": ("exp", lambda r, n: r.uniform(-87, 88, n)), "LOG_LHS": ("log", lambda r, n: np.exp(r.uniform(-80, 80, n)))}


def ulp_error(got, x, fn, sample):
    """max error in float32 ulps: all points against float64 numpy, `sample` of them against mpmath"""
    import mpmath
    mpmath.mp.dps = 40
    ref64 = getattr(np, {"asin": "arcsin", "acos": "arccos", "atan": "arctan"}.get(fn, fn))(x.astype(np.float64))
    ulp = np.maximum(np.spacing(np.abs(ref64.astype(np.float32))).astype(np.float64), 2.0 ** -149)
    err = np.abs(got.astype(np.float64) - ref64) / ulp
    f = getattr(mpmath, fn)
    worst = 0.0
    for i in sample:
        exact = f(mpmath.mpf(float(x[i])))
        worst = max(worst, float(abs(mpmath.mpf(float(got[i])) - exact) / mpmath.mpf(float(ulp[i]))))
        # numpy's double result is itself good to a tiny fraction of a float32 ulp
        assert abs(mpmath.mpf(float(ref64[i])) - exact) <= mpmath.mpf(float(ulp[i])) * 1e-6
    return max(float(err.max()), worst), int(err.argmax())


@pytest.mark.parametrize("opname", list(FLOAT_FNS))
def test_oracle_float_functions_against_mpmath(mpr, orc, opname):
    fn, gen = FLOAT_FNS[opname]
    rng = np.random.default_rng(11)
    x = gen(rng, 100000).astype(np.float32)
    got = orc.float_op(mpr.OP[opname], x, x, 0.0)
    worst, at = ulp_error(got, x, fn, rng.choice(x.size, 300, replace=False))
    assert worst <= (4.0 if fn == "acos" else 2.0), (fn, worst, x[at])


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [-1, 0, 1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("opname", list(FLOAT_FNS))
def test_gpu_float_functions_against_mpmath(mpr, opname, variant):
    """variant -1: the compiled float interpreter; 0..5: the six handler tables of the assembly one;
    6: machine code generated on the device (kernels_voxel_jit.hip)."""
    fn, gen = FLOAT_FNS[opname]
    rng = np.random.default_rng(11)
    x = gen(rng, 100000).astype(np.float32)
    kw = {} if variant < 0 else dict(asm=True, variant=variant)
    got = mpr.dev_float_op(mpr.OP[opname], x, x, 0.0, **kw)
    worst, at = ulp_error(got, x, fn, rng.choice(x.size, 300, replace=False))
    assert worst <= (4.0 if fn == "acos" else 2.0), (fn, variant, worst, x[at])
