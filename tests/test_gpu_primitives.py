"""GPU primitives vs oracle, bit for bit: interval ops (directed rounding on gfx950), float
ops (shared fmath), derivative ops.  Calls go through the C ABI (mpr_test_*_op)."""
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 1 << 16


def same_bits(a, b):
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    both_nan = np.isnan(a) & np.isnan(b)
    return both_nan | (a.view(np.uint32) == b.view(np.uint32))


def gen_floats(rng, n, kind):
    if kind == "bits":       # any bit pattern: subnormals, infinities, NaN, extreme exponents
        return rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    if kind == "unit":
        return rng.uniform(-2, 2, n).astype(np.float32)
    if kind == "wide":
        return (rng.standard_normal(n) * np.exp(rng.uniform(-30, 30, n))).astype(np.float32)
    if kind == "special":
        pool = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 3.4e38, -3.4e38,
                         1.17549435e-38, 0.5, 2.0, 1e-20, -1e-20, 16777216.0, 0.1, -0.1, 3.0], dtype=np.float32)
        return pool[rng.integers(0, pool.size, n)]
    raise ValueError(kind)


def gen_intervals(rng, n, kind):
    a = gen_floats(rng, n, kind)
    b = gen_floats(rng, n, kind)
    if kind in ("bits", "special"):
        # not necessarily ordered: the reference never checks lower <= upper either
        return a, b
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    pt = rng.random(n) < 0.1          # some point intervals
    hi = np.where(pt, lo, hi)
    return lo, hi


INTERVAL_OPS = ["SQUARE_LHS", "SQRT_LHS", "NEG_LHS", "SIN_LHS", "COS_LHS", "ASIN_LHS", "ACOS_LHS", "ATAN_LHS",
                "EXP_LHS", "ABS_LHS", "LOG_LHS", "ADD_LHS_IMM", "ADD_LHS_RHS", "MUL_LHS_IMM", "MUL_LHS_RHS",
                "MIN_LHS_IMM", "MIN_LHS_RHS", "MAX_LHS_IMM", "MAX_LHS_RHS", "SUB_LHS_IMM", "SUB_IMM_RHS",
                "SUB_LHS_RHS", "DIV_LHS_IMM", "DIV_IMM_RHS", "DIV_LHS_RHS", "COPY_IMM", "COPY_LHS", "COPY_RHS"]


@pytest.mark.parametrize("path", ["compiled", "asm", "asm-lhs-forwarded", "asm-rhs-forwarded"])
@pytest.mark.parametrize("kind", ["unit", "wide", "special", "bits"])
@pytest.mark.parametrize("opname", INTERVAL_OPS)
def test_interval_ops_bit_exact(mpr, orc, opname, kind, path):
    """interval_clause (device_math.hpp) and the tile stages' assembly forward walk
    (tile_interp_asm.hpp, all three handler tables) against the oracle: bounds and min/max choice."""
    op = mpr.OP[opname]
    rng = np.random.default_rng(zlib.crc32((opname + kind).encode()))
    a_lo, a_hi = gen_intervals(rng, N, kind)
    b_lo, b_hi = gen_intervals(rng, N, kind)
    kw = {"compiled": {}, "asm": dict(asm=True), "asm-lhs-forwarded": dict(asm=True, variant=1),
          "asm-rhs-forwarded": dict(asm=True, variant=2)}[path]
    for imm in (0.75, -1.25, 0.0):
        g_lo, g_hi, g_ch = mpr.dev_interval_op(op, a_lo, a_hi, b_lo, b_hi, imm, **kw)
        o_lo, o_hi, o_ch = orc.interval_op(op, a_lo, a_hi, b_lo, b_hi, imm)
        ok = same_bits(g_lo, o_lo) & same_bits(g_hi, o_hi) & (g_ch == o_ch)
        bad = np.flatnonzero(~ok)
        assert bad.size == 0, (opname, kind, imm, bad.size,
                               [(a_lo[i], a_hi[i], b_lo[i], b_hi[i], g_lo[i], g_hi[i], o_lo[i], o_hi[i], g_ch[i], o_ch[i]) for i in bad[:5]])


@pytest.mark.parametrize("kind", ["unit", "wide", "special", "bits"])
@pytest.mark.parametrize("opname", INTERVAL_OPS)
def test_float_ops_bit_exact(mpr, orc, opname, kind):
    op = mpr.OP[opname]
    rng = np.random.default_rng(zlib.crc32((opname + kind + "f").encode()))
    a = gen_floats(rng, N, kind)
    b = gen_floats(rng, N, kind)
    for imm in (0.75, -1.25):
        g = mpr.dev_float_op(op, a, b, imm)
        o = orc.float_op(op, a, b, imm)
        bad = np.flatnonzero(~same_bits(g, o))
        assert bad.size == 0, (opname, kind, bad.size, [(a[i], b[i], g[i], o[i]) for i in bad[:5]])


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("kind", ["unit", "wide", "special", "bits"])
@pytest.mark.parametrize("opname", INTERVAL_OPS)
def test_float_ops_assembly_interpreter_bit_exact(mpr, orc, opname, kind, variant):
    """Every opcode through the float pass's assembly interpreter (short tapes): same bits as the
    oracle, including NaN / inf / subnormal / signed-zero operands, in all six handler tables:
    operands from the slot file / lhs forwarded / rhs forwarded (variants 0..2), and the same three
    for a clause whose result dies in the next clause (3..5: neither stored nor addressed); and as
    machine code generated on the device (kernels_voxel_jit.hip, 6..8: the clause's operands come
    from the axis registers / from a copy placed in front of it)."""
    op = mpr.OP[opname]
    rng = np.random.default_rng(zlib.crc32((opname + kind + "f").encode()))
    a = gen_floats(rng, N, kind)
    b = gen_floats(rng, N, kind)
    for imm in (0.75, -1.25, 0.5, -4.0):          # the last two: division by a power of two is translated to a multiplication (6..8)
        g = mpr.dev_float_op(op, a, b, imm, asm=True, variant=variant)
        o = orc.float_op(op, a, b, imm)
        bad = np.flatnonzero(~same_bits(g, o))
        assert bad.size == 0, (opname, kind, bad.size, [(a[i], b[i], g[i], o[i]) for i in bad[:5]])


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("kind", ["unit", "wide", "special", "bits"])
@pytest.mark.parametrize("opname", INTERVAL_OPS)
def test_float_ops_host_generated_code_bit_exact(mpr, orc, opname, kind, variant):
    """Every opcode through the float walk the HOST generates (csrc/voxel_gen.cpp: what the float pass runs for bear-class tapes):
    same bits as the oracle on every class of bit pattern — the inline bodies' fast paths (unit: all 64 lanes ordinary) and their
    stubs into the interpreters' full routines (special / bits / wide: some lane is not); variant 0: result in a fresh register,
    1 / 2: over its lhs / rhs operand (the inline bodies read their operand to the end)."""
    op = mpr.OP[opname]
    rng = np.random.default_rng(zlib.crc32((opname + kind + "f").encode()))
    a = gen_floats(rng, N, kind)
    b = gen_floats(rng, N, kind)
    for imm in (0.75, -1.25, 0.5, -4.0, 3.0e10, 0.0):    # ... a constant division in line / as a product / by the general routine
        g = mpr.dev_float_op_gen(op, a, b, imm, variant=variant)
        o = orc.float_op(op, a, b, imm)
        bad = np.flatnonzero(~same_bits(g, o))
        assert bad.size == 0, (opname, kind, imm, bad.size, [(a[i], b[i], g[i], o[i]) for i in bad[:5]])


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("opname", ["MIN_LHS_IMM", "MIN_LHS_RHS", "MAX_LHS_IMM", "MAX_LHS_RHS"])
def test_decided_min_max_in_host_generated_code(mpr, orc, opname, variant):
    """A min / max clause the tile decided is a copy of the chosen operand, raw (COPY_LHS / COPY_RHS / COPY_IMM of the tile's own
    tape, reference src/context.cu:415-447): NaNs and signed zeros go through untouched."""
    op = mpr.OP[opname]
    rng = np.random.default_rng(zlib.crc32(opname.encode()))
    a = gen_floats(rng, N, "bits")
    b = gen_floats(rng, N, "bits")
    imm = -1.25
    has_rhs = opname.endswith("RHS")
    assert same_bits(mpr.dev_float_op_gen(op, a, b, imm, variant=variant, dl=1), a).all()
    want = b if has_rhs else np.full_like(a, imm)
    assert same_bits(mpr.dev_float_op_gen(op, a, b, imm, variant=variant, dr=1), want).all()
    # a decision for ANOTHER clause (bit 1) changes nothing
    assert same_bits(mpr.dev_float_op_gen(op, a, b, imm, variant=variant, dl=2, dr=2), orc.float_op(op, a, b, imm)).all()


@pytest.mark.parametrize("imm", [0.9, 0.3, 11.0, 5.612245082855225, 30.55555534362793, 0.15, 1.5, 3.0, 7.0, 0.1, -0.7,
                                 1.0000001, 0.99999994, 1.9999999, 2.0 ** -30 * 1.5, 2.0 ** 30 * 1.25, 1e-12, 1e12, 1.17549435e-38])
def test_division_by_a_constant_in_host_generated_code(mpr, orc, imm):
    """The same inline division as kernels_voxel_jit.hip's row 30, with RN(1 / c) computed on the host: the IEEE quotient for
    every operand (test_division_by_a_constant_in_generated_code's operand classes)."""
    op = mpr.OP["DIV_LHS_IMM"]
    rng = np.random.default_rng(zlib.crc32(repr(imm).encode()))
    n = 1 << 19
    cases = [
        (rng.standard_normal(n) * np.exp2(rng.uniform(-55, 55, n))).astype(np.float32),
        (rng.choice([-1.0, 1.0], n) * rng.uniform(1, 2, n) * np.exp2(rng.integers(-63, 63, n))).astype(np.float32),
        (rng.choice([-1.0, 1.0], n) * rng.uniform(1, 2, n) * np.exp2(rng.choice([-63.0, -62.0, 62.0, 63.0], n))).astype(np.float32),
        gen_floats(rng, 1 << 16, "wide"), gen_floats(rng, 1 << 16, "special"), gen_floats(rng, 1 << 16, "bits"),
        (np.exp2(rng.integers(-20, 20, n)) * (1 + rng.integers(-3, 4, n) * 2.0 ** -23)).astype(np.float32),
    ]
    for a in cases:
        for variant in (0, 1):
            g = mpr.dev_float_op_gen(op, a, a, imm, variant=variant)
            o = orc.float_op(op, a, a, imm)
            bad = np.flatnonzero(~same_bits(g, o))
            assert bad.size == 0, (imm, variant, bad.size, [(a[i], g[i], o[i]) for i in bad[:5]])


@pytest.mark.parametrize("kind", ["unit", "wide", "special"])
@pytest.mark.parametrize("opname", INTERVAL_OPS)
def test_deriv_ops_bit_exact(mpr, orc, opname, kind):
    op = mpr.OP[opname]
    rng = np.random.default_rng(zlib.crc32((opname + kind + "d").encode()))
    a = gen_floats(rng, N * 4, kind).reshape(-1, 4)
    b = gen_floats(rng, N * 4, kind).reshape(-1, 4)
    g = mpr.dev_deriv_op(op, a, b, 0.75)
    o = orc.deriv_op(op, a, b, 0.75)
    bad = np.flatnonzero(~same_bits(g, o).all(axis=1))
    assert bad.size == 0, (opname, kind, bad.size, [(a[i], b[i], g[i], o[i]) for i in bad[:3]])


@pytest.mark.parametrize("imm", [0.9, 0.3, 11.0, 0.8, 5.612245082855225, 30.55555534362793, 0.6, 0.15, 1.5, 3.0, 7.0, 0.1, -0.7,
                                 1.0000001, 0.99999994, 1.9999999, 3.4028235e38 / 2 ** 98, 2.0 ** -30 * 1.5, 2.0 ** 30 * 1.25,
                                 1e-12, 1e12, 1.17549435e-38])
def test_division_by_a_constant_in_generated_code(mpr, orc, imm):
    """kernels_voxel_jit.hip turns DIV_LHS_IMM by a constant with 2^-30 <= |c| <= 2^30 into q = x * RN(1/c) and two
    rounds of r = x - c q, q += r * RN(1/c), inline, when every lane's x * x is a positive normal number
    (2^-63 <= |x| < 2^64), and into the general division otherwise: the quotient must be the IEEE one for every operand
    (variant 6 = generated code; the last three constants are outside the range and take the general route)."""
    op = mpr.OP["DIV_LHS_IMM"]
    rng = np.random.default_rng(zlib.crc32(repr(imm).encode()))
    n = 1 << 20
    cases = [
        (rng.standard_normal(n) * np.exp2(rng.uniform(-55, 55, n))).astype(np.float32),       # all lanes in range: the short route
        (rng.choice([-1.0, 1.0], n) * rng.uniform(1, 2, n) * np.exp2(rng.integers(-63, 63, n))).astype(np.float32),   # ... up to its edges
        (rng.choice([-1.0, 1.0], n) * rng.uniform(1, 2, n) * np.exp2(rng.choice([-63.0, -62.0, 62.0, 63.0], n))).astype(np.float32),
        rng.uniform(-4, 4, n).astype(np.float32),
        gen_floats(rng, 1 << 16, "wide"), gen_floats(rng, 1 << 16, "special"), gen_floats(rng, 1 << 16, "bits"),
        # around powers of two, where quotients sit next to rounding boundaries
        (np.exp2(rng.integers(-20, 20, n)) * (1 + rng.integers(-3, 4, n) * 2.0 ** -23)).astype(np.float32),
    ]
    for a in cases:
        g = mpr.dev_float_op(op, a, a, imm, asm=True, variant=6)
        o = orc.float_op(op, a, a, imm)
        bad = np.flatnonzero(~same_bits(g, o))
        assert bad.size == 0, (imm, bad.size, [(a[i], g[i], o[i]) for i in bad[:5]])


def test_square_root_routine_on_every_float(mpr):
    """The float pass's square root (asm_float_bodies.hpp: v_rsq_f32, one coupled Newton step, the exact residual) must be the
    correctly rounded root — what sqrtf gives the oracle — for all 2^32 bit patterns: fast path (positive normal numbers from
    2^-96 up), scaled path (tiny and subnormal), zeros, infinities, negative numbers, NaNs."""
    bad, example = mpr.dev_sqrt_all(0, 1 << 32)
    assert bad == 0, "sqrt differs from the correctly rounded root for %d inputs, e.g. bits 0x%08x" % (bad, example)


# ---- the tile stages' scheduled interval code (csrc/interval_gen.cpp), one clause at a time, on the chip ----
@pytest.mark.parametrize("kind", ["unit", "wide", "special", "bits"])
@pytest.mark.parametrize("opname", INTERVAL_OPS)
def test_scheduled_exact_interval_code_bit_exact(mpr, orc, opname, kind):
    """the EXACT code: the interpreter's routines in line on renamed registers (square, abs, the product's sign-case table, min / max
    with the lanes' decisions) and the calls it keeps (sqrt, div, exp, log, asin, acos, atan) — bounds and choices, every operand class"""
    op = mpr.OP[opname]
    rng = np.random.default_rng(zlib.crc32((opname + kind + "gen").encode()))
    a_lo, a_hi = gen_intervals(rng, N // 4, kind)
    b_lo, b_hi = gen_intervals(rng, N // 4, kind)
    for imm in (0.75, -1.25, 0.0):
        g_lo, g_hi, g_ch, _ = mpr.dev_interval_gen_op(op, a_lo, a_hi, b_lo, b_hi, imm)
        o_lo, o_hi, o_ch = orc.interval_op(op, a_lo, a_hi, b_lo, b_hi, imm)
        ok = same_bits(g_lo, o_lo) & same_bits(g_hi, o_hi) & (g_ch == o_ch)
        bad = np.flatnonzero(~ok)
        assert bad.size == 0, (opname, kind, imm, bad.size,
                               [(a_lo[i], a_hi[i], b_lo[i], b_hi[i], g_lo[i], g_hi[i], o_lo[i], o_hi[i], g_ch[i], o_ch[i]) for i in bad[:5]])


LOOSE_OPS = [o for o in INTERVAL_OPS if o not in ("ASIN_LHS", "ACOS_LHS", "ATAN_LHS")]


@pytest.mark.parametrize("kind", ["unit", "wide"])
@pytest.mark.parametrize("opname", LOOSE_OPS)
def test_scheduled_loose_interval_code_encloses(mpr, orc, opname, kind):
    """the LOOSE code on ordered intervals: wherever a lane does not ask for the exact walk its result encloses the oracle's and it
    decides nothing the oracle does not; ordinary operands never ask"""
    op = mpr.OP[opname]
    rng = np.random.default_rng(zlib.crc32((opname + kind + "loose").encode()))
    a_lo, a_hi = gen_intervals(rng, N // 4, kind)
    b_lo, b_hi = gen_intervals(rng, N // 4, kind)
    for imm in (0.75, -1.25, 3.0e-3):
        g_lo, g_hi, g_ch, asks = mpr.dev_interval_gen_op(op, a_lo, a_hi, b_lo, b_hi, imm, loose=True)
        o_lo, o_hi, o_ch = orc.interval_op(op, a_lo, a_hi, b_lo, b_hi, imm)
        ok = asks == 0
        with np.errstate(invalid="ignore"):
            encl = (g_lo <= o_lo) & (g_hi >= o_hi)
        bad = np.flatnonzero(ok & ~encl)
        assert bad.size == 0, (opname, kind, imm, bad.size, [(a_lo[i], a_hi[i], b_lo[i], b_hi[i], g_lo[i], g_hi[i], o_lo[i], o_hi[i]) for i in bad[:5]])
        assert not (ok & (g_ch != 0) & (g_ch != o_ch)).any(), opname
        if opname not in ("SQRT_LHS", "LOG_LHS", "MUL_LHS_RHS", "DIV_LHS_RHS", "DIV_IMM_RHS"):
            assert ok.all(), (opname, int((~ok).sum()))             # (no domain to leave, no 0 x inf to meet)
        else:
            assert ok.mean() > 0.15, (opname, ok.mean())


@pytest.mark.parametrize("case", ["EXP_LHS", "LOG_LHS", "SQRT_LHS", "SQUARE_LHS", "ABS_LHS", "DIV_LHS_IMM:3.7", "DIV_LHS_IMM:-0.0125", "DIV_LHS_IMM:1.5e6",
                                  "MUL_LHS_IMM:-2.5", "ADD_LHS_IMM:0.3", "MUL_LHS_RHS:-2.5:3.0", "MUL_LHS_RHS:0.25:7.0", "DIV_LHS_RHS/lhs:1.5:4.0",
                                  "DIV_LHS_RHS/rhs:-2.0:5.0", "MIN_LHS_RHS:-1.0:2.0", "MAX_LHS_RHS:-1.0:2.0", "ADD_LHS_RHS:-3.0:1.0e30", "SUB_LHS_RHS:-1.0:2.0"])
def test_loose_interval_code_on_every_float(mpr, case):
    """Every float (all 2^32 bit patterns, as [x, x] and as one end of a wide interval) through the loose code of one clause on the
    chip, against the exact routine's enclosure: NO end that fails to enclose wherever the code does not ask for the exact walk.  This
    is the check of what the hardware's v_exp_f32 / v_log_f32 / v_sqrt_f32 / v_rcp_f32 return that the loose frames' soundness rests on
    (csrc/interval_gen.cpp: two ulps assumed); and the results are not much wider than the exact ones."""
    name, _, rest = case.partition(":")
    name, _, side = name.partition("/")
    args = [float(v) for v in rest.split(":")] if rest else []
    op = mpr.OP[name]
    kw = {}
    if name.endswith("_IMM"):
        kw["imm"] = args[0]
    elif len(args) == 2:
        kw["other"] = (args[0], args[1])
        kw["x_is_rhs"] = side == "lhs"           # ".../lhs: the OTHER operand is the lhs": x is the divisor
    r = mpr.dev_loose_gen(op, **kw)
    assert r["tested"] > (1 << 32) and r["bad"] == 0, (case, r, hex(r["example"]))
    # an operand outside the operation's domain asks for the exact walk, nothing else does
    if name in ("EXP_LHS", "SQUARE_LHS", "ABS_LHS", "MUL_LHS_IMM", "ADD_LHS_IMM", "DIV_LHS_IMM", "MIN_LHS_RHS", "MAX_LHS_RHS", "SUB_LHS_RHS"):
        assert r["asked_for_exact"] <= 4, (case, r)          # ([inf, inf] and [-inf, -inf]: no width)
    # widths: 2^-24 units of the value beyond the exact enclosure's (exp: (|t| + 4) 2^-23 either side at t up to 128 -> a few hundred)
    assert r["widest"] < (1 << 12), (case, r)


def test_loose_division_by_constants_on_every_float(mpr, tapes):
    """... the division by a constant c (1 / c rounded down and up on the host: csrc/interval_gen.cpp: recip_bounds) for bear's own
    divisors and a few awkward ones, every float x as [x, x] and as one end of a wide interval (a NEGATIVE divisor swaps the ends: a
    routine that takes them from the wrong side passes every [x, x] — round 4's bug)"""
    d = mpr.decode(tapes("bear").data)
    consts = sorted({c[4] for c in d if c[0] == "DIV_LHS_IMM"})
    assert len(consts) >= 8
    for c in consts + [3.0, -7.0, 1e-30, -1e30, 1.0000001, 0.99999994, 2.0 ** -100, -(2.0 ** 100)]:
        r = mpr.dev_loose_gen(mpr.OP["DIV_LHS_IMM"], imm=c)
        assert r["tested"] > 8_000_000_000 and r["bad"] == 0, (c, r, hex(r["example"]))
        assert r["widest"] < 64, (c, r)


# (operation, float values beyond an end of the exact enclosure of [x, x], farthest in units of the end's last place, x that has a
# value where the enclosure has none or the other way round) over all 2^32 bit patterns: profiles/r05_float_in_enclosure.txt
FLOAT_IN_ENCLOSURE = [("SQUARE_LHS", 0, 0, 0), ("SQRT_LHS", 0, 0, 0), ("NEG_LHS", 0, 0, 0), ("ABS_LHS", 0, 0, 0), ("SIN_LHS", 0, 0, 2), ("COS_LHS", 0, 0, 2),
                      ("EXP_LHS", 22, 1, 0), ("LOG_LHS", 2, None, 0), ("ASIN_LHS", 761614, 1, 0), ("ACOS_LHS", 191574, 1, 0), ("ATAN_LHS", 1227468, 2, 0)]


@pytest.mark.parametrize("opname,outside,farthest,nan_kind", FLOAT_IN_ENCLOSURE)
def test_float_values_against_the_exact_enclosures_on_every_float(mpr, opname, outside, farthest, nan_kind):
    """Is the float pass's f(x) inside the exact interval routine's enclosure of [x, x] (the reference's inc/gpu_interval.hpp:306-390
    with double-precision libm inside, against include/mpr_fmath.h's single-precision routines)?  Every float.  This is the premise
    of every shortcut that lets a tile decide LESS than the reference's tile did (looser enclosures; a 16^3 tile on the root tape): a
    min / max the reference decided on the strength of its enclosures picks the same operand in a float walk that still compares.
    It holds for the arithmetic operations and the square root on every float; it fails by one unit in the last place for 22
    arguments of exp, by one or two units for 0.02 - 0.03 % of the arguments of asin / acos / atan (routines that are accurate to a
    unit or two against enclosures that are tight to the unit), and for log at +-0 (the reference's [0, RU(log 0)] = [0, -inf]).  What
    the shortcuts do about it: no looser code is generated for tapes with asin / acos / atan (csrc/interval_gen.cpp); frames that
    start at the 16^3 tiles are verified against the 64^3 tiles unless the whole view stays inside the domain where every routine
    is isotone (csrc/frame_domain.cpp: tapes with asin / acos / atan never do; log at 0 leaves it); for exp the gap stands: a pixel
    differs only if one of 22 floats meets a min / max that the reference decides by less than a unit in the last place — no frame of
    the 2040 + 14000 seeds of the two sweeps (profiles/r05_fuzz_sweep.txt, r05_paranoid_sweep.txt) does.  The counts are pinned: a
    change of either routine shows here."""
    r = mpr.dev_float_in_enclosure(mpr.OP[opname])
    assert r["tested"] == (1 << 32) - 2 * ((1 << 23) - 1), r          # every bit pattern that is not a NaN
    assert r["outside"] == outside and r["nan_mismatch"] == nan_kind, (opname, r, hex(r["example_outside"]))
    if farthest is not None:
        assert r["farthest"] == farthest, (opname, r)
    if opname == "LOG_LHS":
        assert r["example_outside"] & 0x7FFFFFFF == 0                  # +-0 and nothing else


@pytest.mark.parametrize("opname,imm", [("DIV_LHS_IMM", 3.7), ("DIV_LHS_IMM", -0.0125), ("DIV_IMM_RHS", 1.0), ("DIV_IMM_RHS", -7.5), ("MUL_LHS_IMM", -2.5),
                                        ("ADD_LHS_IMM", 0.3), ("SUB_IMM_RHS", 1e-3)])
def test_float_values_of_operations_with_a_constant_inside_the_exact_enclosures(mpr, opname, imm):
    r = mpr.dev_float_in_enclosure(mpr.OP[opname], imm=imm)
    assert r["tested"] == (1 << 32) - 2 * ((1 << 23) - 1) and r["outside"] == 0, (opname, imm, r, hex(r["example_outside"]))


@pytest.mark.parametrize("opname,imm", [("EXP_LHS", 0.0), ("LOG_LHS", 0.0), ("SQRT_LHS", 0.0), ("SQUARE_LHS", 0.0), ("NEG_LHS", 0.0), ("ABS_LHS", 0.0),
                                        ("DIV_LHS_IMM", 3.7), ("DIV_LHS_IMM", -0.3), ("DIV_LHS_IMM", 30.55555534362793), ("DIV_LHS_IMM", 4.0),
                                        ("DIV_LHS_IMM", 1e-12), ("MUL_LHS_IMM", -2.5), ("ADD_LHS_IMM", 0.3), ("SUB_IMM_RHS", 1.0), ("DIV_IMM_RHS", 1.0),
                                        ("MIN_LHS_IMM", 0.25), ("MAX_LHS_IMM", -0.0)])
def test_host_generated_float_code_on_every_float(mpr, opname, imm):
    """One clause through the float pass's host-generated code (csrc/voxel_gen.cpp) on all 2^32 bit patterns against the float pass's
    definition (csrc/device_math.hpp: float_clause = include/mpr_fmath.h, which the oracle shares): the same bits, or two NaNs — the
    in-line paths, their range tests and the routines behind them, with nothing sampled."""
    r = mpr.dev_float_gen_all(mpr.OP[opname], imm=imm)
    assert r["tested"] == 1 << 32 and r["bad"] == 0, (opname, imm, r, hex(r["example"]))
