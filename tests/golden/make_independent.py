"""Independent brute-force evaluator -> tests/golden/independent_*.npz  (TEST INFRASTRUCTURE).

Shares NOTHING with the product or with oracle/mpr_oracle.c: it parses the .frep archives itself
(format: SURVEY.md section 8(c)) and evaluates the expression DAG node by node — no tape, no slots,
no tiles, no hierarchy — at every voxel centre, in numpy float32.

Why float32 and not "more precise": the reference decides `f < 0` on a float32 value.  + - * / sqrt,
negation, abs, min, max are IEEE-754 operations, so every faithful implementation of the reference's
algorithm — the CUDA renderer, this repository's kernels, its oracle, numpy — gets the SAME bits out of
them; they cannot make two renderers disagree.  What can are (a) the transcendental functions (CUDA's
libm, mpr_fmath and the correctly rounded value used here are each within 2 ulp of the truth) and (b)
fused multiply-adds the compiler may or may not form — which for the benchmark views
(identity / T(3,2) = 0.3) cannot change a position: ((0*x + 0*y) + m32*z) + m33 has no product feeding
an addition with an inexact partner.  evaluate32() therefore carries, next to
the value, the SPREAD two faithful implementations can be apart by: 3 ulp at each transcendental,
propagated to first order, plus one rounding per later operation whose operands already differ.

Per pixel that gives what ANY faithful renderer must produce:
  hmin   highest z with f < -2*spread   (inside for every implementation)
  hmax   highest z with f < +2*spread   (inside for some implementation)
         -> the heightmap h (src/context.cu:936-948: max z with f < 0; z = 0 is invisible) satisfies
            hmin <= h <= hmax and h == hmin wherever hmin == hmax.  Pixels with hmin != hmax are the
            FRAGILITY MASK of SURVEY.md section 8(c): empty for the models without transcendentals.
  normal the packed normal (src/context.cu:1123-1131) from a float64 forward-mode gradient one voxel in
         front of hmin (:1001-1005), every min / max / abs branch taken as the float32 evaluation takes
         it (inc/gpu_deriv.hpp:106-130); `nfrag` marks pixels where the float32 gradient is not
         comparable: an undecided branch between different gradients, a vanishing gradient, acos / asin /
         sqrt at the ill-conditioned end of their domain, or hmin != hmax.
2-D models: `lo` / `hi` images the same way (1 where inside for every / some implementation).

Sizes too large for numpy (bear 1024^3, architecture 2048^3, gears 4096^2) are sampled: a seeded random
set of pixel columns, full depth.  evaluate() — float64 with a running rounding-error bound — supplies
the gradient and serves as a cross-check of evaluate32 (they must agree wherever |f| is above the bound).

Usage:  python tests/golden/make_independent.py [name ...]      (writes next to this file)
"""
import multiprocessing as mp
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
MODELS = os.path.join(os.path.dirname(os.path.dirname(HERE)), "fixtures", "models")
U = 2.0 ** -24

# libfive packed opcode numbers (CMakeLists.txt:6-8 LIBFIVE_PACKED_OPCODES)
CONST, VX, VY, VZ = 1, 2, 3, 4
SQUARE, SQRT, NEG, SIN, COS, TAN, ASIN, ACOS, ATAN, EXP, ABS, LOG, RECIP = range(7, 20)
ADD, MUL, MIN, MAX, SUB, DIV = range(20, 26)


def parse_frep(path):
    """-> list of (op, a, b, value); operands are indices into the list, root last."""
    raw = open(path, "rb").read()
    assert raw[0:1] == b"T"
    p = 1
    for _ in range(2):                       # "name", "doc"
        assert raw[p:p + 1] == b'"'
        q = raw.index(b'"', p + 1)
        p = q + 1
    nodes = []
    while raw[p] != 0xFF:
        op = raw[p]
        p += 1
        if op == CONST:
            nodes.append((op, -1, -1, struct.unpack_from("<f", raw, p)[0]))
            p += 4
        elif op in (VX, VY, VZ):
            nodes.append((op, -1, -1, 0.0))
        elif 7 <= op <= 19:
            a, = struct.unpack_from("<I", raw, p)
            p += 4
            nodes.append((op, a, -1, 0.0))
        elif 20 <= op <= 31:
            r, l = struct.unpack_from("<II", raw, p)          # right operand first
            p += 8
            nodes.append((op, l, r, 0.0))
        else:
            raise ValueError("opcode %d" % op)
    return nodes


def expression(name):
    """The in-code expressions of the reference's benchmarks, as the same node lists."""
    n = []

    def k(v): n.append((CONST, -1, -1, float(np.float32(v)))); return len(n) - 1
    def ax(o): n.append((o, -1, -1, 0.0)); return len(n) - 1
    def un(o, a): n.append((o, a, -1, 0.0)); return len(n) - 1
    def bi(o, a, b): n.append((o, a, b, 0.0)); return len(n) - 1
    X, Y, Z = ax(VX), ax(VY), ax(VZ)
    if name == "two_spheres":     # benchmark/brute.cu:87-91
        def sph(cx):
            d = bi(ADD, bi(ADD, un(SQUARE, bi(ADD, X, k(cx))), un(SQUARE, Y)), un(SQUARE, Z))
            return bi(SUB, un(SQRT, d), k(0.25))
        bi(MIN, sph(0.5), sph(-0.5))
    elif name == "circle":        # benchmark/circle.cpp:22-24
        bi(SUB, un(SQRT, bi(ADD, un(SQUARE, bi(ADD, X, k(1))), un(SQUARE, bi(ADD, Y, k(1))))), k(1.8))
    else:
        raise KeyError(name)
    return n


def evaluate(nodes, x, y, z, deriv=False, picks=None):
    """float64 values of the root, the running float32 error bound, and (deriv) the gradient plus the
    smallest min/max margin relative to its bound.  x, y, z: float64 arrays (transformed position)."""
    with np.errstate(all="ignore"):
        val = [None] * len(nodes)
        err = [None] * len(nodes)
        grd = [None] * len(nodes) if deriv else None
        tie = np.full(x.shape, np.inf) if deriv else None
        last = {}
        for i, (op, a, b, c) in enumerate(nodes):
            for o in (a, b):
                if o >= 0:
                    last[o] = i
        zero = np.zeros_like(x)
        for i, (op, a, b, c) in enumerate(nodes):
            if op == CONST:
                v, e = np.full_like(x, c), zero
                g = (zero, zero, zero)
            elif op in (VX, VY, VZ):
                v = (x, y, z)[op - VX]
                e = 4 * U * np.abs(v) + 2 * U          # p -> [-1,1] -> matrix row -> / w, each rounded
                g = tuple(np.ones_like(x) if k == op - VX else zero for k in range(3))
            else:
                A, eA = val[a], err[a]
                gA = grd[a] if deriv else None
                if b >= 0:
                    B, eB = val[b], err[b]
                    gB = grd[b] if deriv else None
                if op == ADD:
                    v = A + B; e = eA + eB
                    if deriv: g = tuple(p + q for p, q in zip(gA, gB))
                elif op == SUB:
                    v = A - B; e = eA + eB
                    if deriv: g = tuple(p - q for p, q in zip(gA, gB))
                elif op == MUL:
                    v = A * B; e = np.abs(A) * eB + np.abs(B) * eA + eA * eB
                    if deriv: g = tuple(p * B + q * A for p, q in zip(gA, gB))
                elif op == DIV:
                    v = A / B
                    e = (eA + np.abs(v) * eB) / np.maximum(np.abs(B) - eB, 1e-300)
                    if deriv: g = tuple((p * B - q * A) / (B * B) for p, q in zip(gA, gB))
                elif op in (MIN, MAX):
                    pickA = (A < B) if op == MIN else (A >= B)       # inc/gpu_deriv.hpp:106-130
                    v = np.where(np.isnan(A), B, np.where(np.isnan(B), A, np.minimum(A, B) if op == MIN else np.maximum(A, B)))
                    # the losing operand's error matters only where the two can swap places
                    lessA = (A + eA < B - eB) if op == MIN else (A - eA > B + eB)
                    lessB = (B + eB < A - eA) if op == MIN else (B - eB > A + eA)
                    e = np.where(lessA, eA, np.where(lessB, eB, np.maximum(eA, eB)))
                    if deriv:
                        firm = None
                        if picks is not None:              # the float32 evaluation's decision (reproducible)
                            pickA, firm = picks[i]
                        g = tuple(np.where(pickA, p, q) for p, q in zip(gA, gB))
                        # an undecided pick only matters when the two branches have different gradients
                        gdiff = sum(np.abs(p - q) for p, q in zip(gA, gB))
                        gsum = sum(np.abs(p) + np.abs(q) for p, q in zip(gA, gB))
                        if firm is None:
                            close = np.abs(A - B) / (4 * (eA + eB) + 1e-300)
                        else:
                            close = np.where(firm, np.inf, 0.0)
                        tie = np.minimum(tie, np.where(gdiff > 1e-5 * gsum, close, np.inf))
                elif op == SQUARE:
                    v = A * A; e = 2 * np.abs(A) * eA + eA * eA
                    if deriv: g = tuple(2 * A * p for p in gA)
                elif op == SQRT:
                    v = np.sqrt(A)
                    e = eA / np.maximum(2 * np.sqrt(np.maximum(A - eA, 0)), 1e-300)
                    e = np.where(A <= eA, np.sqrt(np.maximum(A + eA, 0)), e)     # argument may reach zero
                    if deriv:
                        g = tuple(p / (2 * v) for p in gA)
                        tie = np.minimum(tie, np.where((A < 1e-4) & (sum(np.abs(p) for p in gA) > 0), 0.0, np.inf))
                elif op == NEG:
                    v = -A; e = eA
                    if deriv: g = tuple(-p for p in gA)
                elif op == ABS:
                    v = np.abs(A); e = eA
                    if deriv:
                        neg, firm = (A < 0, None) if picks is None else picks[i]
                        g = tuple(np.where(neg, -p, p) for p in gA)
                        gsum = sum(np.abs(p) for p in gA)
                        close = np.abs(A) / (4 * eA + 1e-300) if firm is None else np.where(firm, np.inf, 0.0)
                        tie = np.minimum(tie, np.where(gsum > 0, close, np.inf))
                elif op == SIN:
                    v = np.sin(A); e = eA + U
                    if deriv: g = tuple(p * np.cos(A) for p in gA)
                elif op == COS:
                    v = np.cos(A); e = eA + U
                    if deriv: g = tuple(-p * np.sin(A) for p in gA)
                elif op == ASIN:
                    v = np.arcsin(A); d = np.sqrt(np.maximum(1 - (np.abs(A) + eA) ** 2, 1e-300)); e = eA / d + U * np.abs(v)
                    if deriv:
                        g = tuple(p / np.sqrt(1 - A * A) for p in gA)
                        tie = np.minimum(tie, np.where((1 - np.abs(A) < 2e-2) & (sum(np.abs(p) for p in gA) > 0), 0.0, np.inf))   # ill-conditioned in float32
                elif op == ACOS:
                    v = np.arccos(A); d = np.sqrt(np.maximum(1 - (np.abs(A) + eA) ** 2, 1e-300)); e = eA / d + U * np.abs(v)
                    if deriv:
                        g = tuple(-p / np.sqrt(1 - A * A) for p in gA)
                        tie = np.minimum(tie, np.where((1 - np.abs(A) < 2e-2) & (sum(np.abs(p) for p in gA) > 0), 0.0, np.inf))
                elif op == ATAN:
                    v = np.arctan(A); e = eA + U * np.abs(v)
                    if deriv: g = tuple(p / (1 + A * A) for p in gA)
                elif op == EXP:
                    v = np.exp(A); e = v * np.expm1(np.minimum(eA, 50.0)) + U * v
                    if deriv: g = tuple(p * v for p in gA)
                elif op == LOG:
                    v = np.log(A); e = -np.log1p(-np.minimum(eA / np.maximum(np.abs(A), 1e-300), 0.999)) + U * np.abs(v)
                    if deriv: g = tuple(p / A for p in gA)
                else:
                    raise ValueError("opcode %d unsupported by src/tape.cpp:122-180" % op)
                e = e + 2 * U * np.abs(v)            # the operation's own rounding (2u: transcendentals within 2 ulp)
                e = np.where(np.isfinite(e), e, np.inf)
            val[i], err[i] = v, e
            if deriv:
                grd[i] = g
            for o in (a, b):                      # free operands after their last use
                if o >= 0 and last.get(o) == i and o != len(nodes) - 1:
                    val[o] = err[o] = None
                    if deriv:
                        grd[o] = None
        if deriv:
            return val[-1], err[-1], grd[-1], tie
        return val[-1], err[-1]



def evaluate32(nodes, x, y, z, coord_spread, trace=None, picks=None):
    """What a faithful float32 implementation computes, and how far two such implementations can be
    apart.  + - * / sqrt, negation, abs, min, max are IEEE operations: every implementation returns
    the same bits for the same operands, so they create no spread of their own; numpy float32 IS such
    an implementation.  Spread enters through (a) the transcendental functions — CUDA's libm, this
    repository's mpr_fmath and the correctly rounded value used here are each within 2 ulp of the
    truth, so 3 ulp of each other — and (b) the perspective division of the position, where the
    compiler's choice to fuse  m32 * z + m33  or not moves w by an ulp (`coord_spread`, in ulps).
    It then propagates to first order, plus one rounding per operation whose operands differ.
    x, y, z: float32 arrays.  Returns (value float32, spread float64)."""
    f32 = np.float32
    with np.errstate(all="ignore"):
        val = [None] * len(nodes)
        spr = [None] * len(nodes)
        last = {}
        for i, (op, a, b, c) in enumerate(nodes):
            for o in (a, b):
                if o >= 0:
                    last[o] = i
        zero = np.zeros(x.shape, np.float64)
        for i, (op, a, b, c) in enumerate(nodes):
            if op == CONST:
                v, s = np.full(x.shape, c, f32), zero
            elif op in (VX, VY, VZ):
                v = (x, y, z)[op - VX]
                s = coord_spread * U * np.abs(v.astype(np.float64))
            else:
                A, sA = val[a], spr[a]
                A64 = A.astype(np.float64)
                if b >= 0:
                    B, sB = val[b], spr[b]
                    B64 = B.astype(np.float64)
                own = 0.0                                   # ulps the operation itself may differ by
                if op == ADD:
                    v = A + B; s = sA + sB
                elif op == SUB:
                    v = A - B; s = sA + sB
                elif op == MUL:
                    v = A * B; s = np.abs(A64) * sB + np.abs(B64) * sA + sA * sB
                elif op == DIV:
                    v = A / B
                    s = (sA + np.abs(v.astype(np.float64)) * sB) / np.maximum(np.abs(B64) - sB, 1e-300)
                elif op in (MIN, MAX):
                    fn = np.minimum if op == MIN else np.maximum
                    v = np.where(np.isnan(A), B, np.where(np.isnan(B), A, fn(A, B)))
                    lessA = (A64 + sA < B64 - sB) if op == MIN else (A64 - sA > B64 + sB)
                    lessB = (B64 + sB < A64 - sA) if op == MIN else (B64 - sB > A64 + sA)
                    s = np.where(lessA, sA, np.where(lessB, sB, np.maximum(sA, sB)))
                    s = np.where(np.isnan(A), sB, np.where(np.isnan(B), sA, s))
                    if picks is not None:      # inc/gpu_deriv.hpp:106-130: which operand the Deriv takes, and is that firm
                        picks[i] = ((A < B) if op == MIN else (A >= B), lessA | lessB | ((sA == 0) & (sB == 0)))
                elif op == SQUARE:
                    v = A * A; s = 2 * np.abs(A64) * sA + sA * sA
                elif op == SQRT:
                    v = np.sqrt(A)
                    lo = np.sqrt(np.maximum(A64 - sA, 0)); hi = np.sqrt(np.maximum(A64 + sA, 0))
                    s = np.maximum(hi - v.astype(np.float64), v.astype(np.float64) - lo)
                    s = np.where((sA > 0) & (A64 - sA < 0), np.inf, s)          # NaN or not depends on the implementation
                elif op == NEG:
                    v = -A; s = sA
                elif op == ABS:
                    v = np.abs(A); s = sA
                    if picks is not None:
                        picks[i] = (A < 0, np.abs(A64) > sA)
                elif op in (SIN, COS, ASIN, ACOS, ATAN, EXP, LOG):
                    fn = {SIN: np.sin, COS: np.cos, ASIN: np.arcsin, ACOS: np.arccos, ATAN: np.arctan, EXP: np.exp, LOG: np.log}[op]
                    v64 = fn(A64)
                    v = v64.astype(f32)
                    loA, hiA = A64 - sA, A64 + sA
                    if op in (ASIN, ACOS):
                        loA, hiA = np.clip(loA, -1, 1), np.clip(hiA, -1, 1)
                    if op == LOG:
                        loA = np.maximum(loA, 1e-300)
                    if op in (SIN, COS):
                        s = sA                                                   # |derivative| <= 1
                    else:
                        s = np.maximum(np.abs(fn(hiA) - v64), np.abs(fn(loA) - v64))
                    # domain edges: whether the argument is inside may itself depend on the implementation
                    if op in (ASIN, ACOS):
                        s = np.where((sA > 0) & (np.abs(A64) + sA > 1), np.inf, s)
                    if op == LOG:
                        s = np.where((sA > 0) & (A64 - sA <= 0), np.inf, s)
                    own = 3.0
                else:
                    raise ValueError("opcode %d unsupported by src/tape.cpp:122-180" % op)
                if op not in (NEG, ABS, MIN, MAX):
                    any_spread = (sA > 0) | ((sB > 0) if b >= 0 else False)
                    own = np.where(any_spread, np.maximum(own, 1.0), own)
                    s = s + own * U * np.abs(v.astype(np.float64)) + np.where(np.asarray(own) > 0, 1e-45, 0.0)
                s = np.where(np.isfinite(s), s, np.inf)
                # an infinite result (exp overflow and what follows from it) is the same infinity in every
                # implementation unless an operand was itself undecided or sits at the overflow threshold
                ops_ok = np.isfinite(sA) & (np.isfinite(sB) if b >= 0 else True)
                if op == EXP:
                    ops_ok = ops_ok & np.isinf(np.exp(A64 - sA).astype(f32))
                s = np.where(np.isinf(v) & ops_ok, 0.0, s)
            val[i], spr[i] = v, s
            if trace is not None:
                trace.append((i, op, float(np.isinf(s).mean()), float(np.isnan(v).mean())))
            for o in (a, b):
                if o >= 0 and last.get(o) == i:
                    val[o] = spr[o] = None
        return val[-1], spr[-1]


def positions32(px, py, pz, S, mat):
    """Voxel centre -> transformed position exactly as src/context.cu:734-760 computes it in float32
    (no fused multiply-adds; S is a power of two, so (p + 0.5) / S is exact)."""
    f32 = np.float32
    r = f32(1.0) / f32(S)
    fx = ((px.astype(f32) + f32(0.5)) * r - f32(0.5)) * f32(2)
    fy = ((py.astype(f32) + f32(0.5)) * r - f32(0.5)) * f32(2)
    fz = ((pz.astype(f32) + f32(0.5)) * r - f32(0.5)) * f32(2)
    m = mat.astype(f32)
    w = m[3, 0] * fx + m[3, 1] * fy + m[3, 2] * fz + m[3, 3]
    return tuple((m[i, 0] * fx + m[i, 1] * fy + m[i, 2] * fz + m[i, 3]) / w for i in range(3))


def positions(px, py, pz, S, mat):
    """Voxel centre -> transformed model position, in float64 (src/context.cu:734-760)."""
    fx = ((px + 0.5) / S - 0.5) * 2.0
    fy = ((py + 0.5) / S - 0.5) * 2.0
    fz = ((pz + 0.5) / S - 0.5) * 2.0
    m = mat.astype(np.float64)
    w = m[3, 0] * fx + m[3, 1] * fy + m[3, 2] * fz + m[3, 3]
    return tuple((m[i, 0] * fx + m[i, 1] * fy + m[i, 2] * fz + m[i, 3]) / w for i in range(3))


def columns_3d(args):
    nodes, S, mat, xs, ys = args
    n = xs.size
    hmin = np.zeros(n, np.int32)
    hmax = np.zeros(n, np.int32)
    zc = max(1, (1 << 19) // n)
    for z0 in range(0, S, zc):
        zz = np.arange(z0, min(S, z0 + zc))
        PX = np.repeat(xs[None, :], zz.size, 0).astype(np.float64)
        PY = np.repeat(ys[None, :], zz.size, 0).astype(np.float64)
        PZ = np.repeat(zz[:, None], n, 1).astype(np.float64)
        v, e = evaluate32(nodes, *positions32(PX, PY, PZ, S, mat), coord_spread=0.0)
        v = v.astype(np.float64)
        sure = (v < -2 * e) & (v < 0)
        maybe = (v < 2 * e) | ((v < 0) & True)                # NaN compares false: never filled (f < 0)
        maybe = np.where(np.isnan(v), np.isinf(e), maybe)     # a NaN whose existence depends on the implementation
        Z = np.repeat(zz[:, None], n, 1)
        hmin = np.maximum(hmin, np.where(sure, Z, 0).max(0))
        hmax = np.maximum(hmax, np.where(maybe, Z, 0).max(0))
    # normals one voxel in front of the certain surface
    pz = np.where(hmin < S - 1, hmin + 1, hmin).astype(np.float64)
    picks = {}
    evaluate32(nodes, *positions32(xs.astype(np.float64), ys.astype(np.float64), pz, S, mat), coord_spread=0.0, picks=picks)
    v, e, g, tie = evaluate(nodes, *positions(xs.astype(np.float64), ys.astype(np.float64), pz, S, mat), deriv=True, picks=picks)
    norm = np.sqrt(g[0] ** 2 + g[1] ** 2 + g[2] ** 2)
    with np.errstate(all="ignore"):
        ch = [np.floor(np.clip(c / norm * 127 + 128, 0, 255)) for c in g]
    ok = np.isfinite(norm) & (norm > 1e-6)
    ch = [np.where(ok, c, 0).astype(np.int64) for c in ch]
    packed = np.where(ok & (hmin > 0), (0xFF << 24) | (ch[2] << 16) | (ch[1] << 8) | ch[0], 0)
    nfrag = (~ok) | (tie < 1.0) | (hmin != hmax)
    return hmin, hmax, packed.astype(np.uint32), nfrag


def render3d(nodes, S, mat, pixels=None, procs=8):
    if pixels is None:
        ys, xs = np.divmod(np.arange(S * S), S)
    else:
        ys, xs = pixels // S, pixels % S
    chunks = np.array_split(np.arange(xs.size), max(1, xs.size // 2048))
    with mp.Pool(procs) as pool:
        parts = pool.map(columns_3d, [(nodes, S, mat, xs[c], ys[c]) for c in chunks])
    return tuple(np.concatenate([p[k] for p in parts]) for k in range(4))


def pixels_2d(args):
    nodes, S, mat, z, xs, ys = args
    fx = ((xs + 0.5) / S - 0.5) * 2.0
    fy = ((ys + 0.5) / S - 0.5) * 2.0
    m = mat.astype(np.float64)
    w = m[2, 0] * fx + m[2, 1] * fy + m[2, 2]
    x = (m[0, 0] * fx + m[0, 1] * fy + m[0, 2]) / w
    y = (m[1, 0] * fx + m[1, 1] * fy + m[1, 2]) / w
    f32 = np.float32
    v, e = evaluate32(nodes, x.astype(f32), y.astype(f32), np.full(x.shape, z, f32), coord_spread=0.0)
    v = v.astype(np.float64)
    sure = (v < -2 * e) & (v < 0)
    maybe = np.where(np.isnan(v), np.isinf(e), (v < 2 * e) | (v < 0))
    return sure.astype(np.uint8), maybe.astype(np.uint8)


def render2d(nodes, S, mat, z=0.0, pixels=None, procs=8):
    if pixels is None:
        ys, xs = np.divmod(np.arange(S * S), S)
    else:
        ys, xs = pixels // S, pixels % S
    chunks = np.array_split(np.arange(xs.size), max(1, xs.size // 65536))
    with mp.Pool(procs) as pool:
        parts = pool.map(pixels_2d, [(nodes, S, mat, z, xs[c].astype(np.float64), ys[c].astype(np.float64)) for c in chunks])
    return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])


def view3():
    T = np.eye(4, dtype=np.float32)
    T[3, 2] = 0.3                                  # benchmark/render_3d_table.cpp:48-49
    return T


# name -> (model, dim, size, number of sampled pixels or 0 for the whole image)
CONFIGS = {
    "bear_3d_128": ("bear", 3, 128, 0),
    "bear_3d_256": ("bear", 3, 256, 0),
    "bear_3d_1024_sample": ("bear", 3, 1024, 4096),
    "architecture_3d_128": ("architecture", 3, 128, 0),
    "architecture_3d_256": ("architecture", 3, 256, 0),
    "architecture_3d_2048_sample": ("architecture", 3, 2048, 4096),
    "involute_gear_3d_128": ("involute_gear_3d", 3, 128, 0),
    "hello_world_3d_128": ("hello_world", 3, 128, 0),
    "two_spheres_3d_128": ("two_spheres", 3, 128, 0),
    "prospero_2d_512": ("prospero", 2, 512, 0),
    "prospero_2d_1024": ("prospero", 2, 1024, 0),
    "involute_gear_2d_2d_512": ("involute_gear_2d", 2, 512, 0),
    "involute_gear_2d_2d_4096_sample": ("involute_gear_2d", 2, 4096, 1 << 18),
    "hello_world_2d_256": ("hello_world", 2, 256, 0),
    "circle_2d_256": ("circle", 2, 256, 0),
    # random shapes of tests/test_gpu_fuzz_shapes.py (tests/golden/make_shapes.py wrote their .frep files): divisions by negative
    # constants, steep exp / log blends, asin / acos.  Not the shapes whose asin / acos LEAVE their domain inside the view: the
    # reference's hierarchy then draws what no evaluation voxel by voxel predicts (a tile "filled" on the strength of a NaN end:
    # DESIGN.md 3, frames that start at the 16^3 tiles) - those are held against the oracle only
    "shape_12_0_3d_128": ("shape_12_0", 3, 128, 0),
    "shape_12_3_3d_128": ("shape_12_3", 3, 128, 0),
    "shape_4_15_3d_128": ("shape_4_15", 3, 128, 0),
    "shape_4_9_3d_128": ("shape_4_9", 3, 128, 0),
    "shape_12_21_3d_128": ("shape_12_21", 3, 128, 0),
    "shape_64_5_3d_128": ("shape_64_5", 3, 128, 0),
    "shape_4_15_2d_256": ("shape_4_15", 2, 256, 0),
    "shape_4_9_2d_256": ("shape_4_9", 2, 256, 0),
    "shape_12_21_2d_256": ("shape_12_21", 2, 256, 0),
    "shape_64_5_2d_256": ("shape_64_5", 2, 256, 0),
}
SHAPES = os.path.join(HERE, "shapes")


def make(name):
    model, dim, S, nsample = CONFIGS[name]
    path = os.path.join(SHAPES if model.startswith("shape_") else MODELS, model + ".frep")
    nodes = parse_frep(path) if os.path.exists(path) else expression(model)
    pixels = None
    if nsample:
        pixels = np.sort(np.random.default_rng(12345).choice(S * S, nsample, replace=False)).astype(np.int64)
    out = {"model": model, "dim": dim, "size": S, "pixels": pixels if pixels is not None else np.zeros(0, np.int64)}
    if dim == 3:
        hmin, hmax, normal, nfrag = render3d(nodes, S, view3(), pixels)
        out.update(hmin=hmin.astype(np.int16), hmax=hmax.astype(np.int16), normal=normal, nfrag=np.packbits(nfrag))
        frag = float((hmin != hmax).mean())
        print("%-32s filled %7d  fragile heights %.4f%%  fragile normals %.3f%%" % (
            name, int((hmin > 0).sum()), 100 * frag, 100 * float(nfrag[hmin > 0].mean()) if (hmin > 0).any() else 0))
    else:
        lo, hi = render2d(nodes, S, np.eye(3, dtype=np.float32), 0.0, pixels)
        out.update(lo=np.packbits(lo), hi=np.packbits(hi))
        print("%-32s inside %8d  fragile %.4f%%" % (name, int(lo.sum()), 100 * float((lo != hi).mean())))
    np.savez_compressed(os.path.join(HERE, "independent_" + name + ".npz"), **out)


if __name__ == "__main__":
    for nm in (sys.argv[1:] or list(CONFIGS)):
        make(nm)
