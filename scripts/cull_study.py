"""Would a centred (mean-value) form cull the smallest tiles that the float pass finds empty?  

Test / measurement infrastructure (uses the CPU oracle for the frame).  For every smallest (4^3) tile the reference's last tile
stage leaves ambiguous — the tiles the float pass evaluates voxel by voxel — this script computes, in float64 numpy over all
tiles at once:
  truth        is any of its 64 voxels inside (f < 0)?
  natural      the natural interval extension of the root tape over the tile (what the tile stages do; float64, no directed
               rounding: a study, not a renderer)
  mean value   f(c) + sum_i dF/dx_i(X) (X_i - c_i): interval forward-mode AD of the same tape (value + three gradient intervals per
               slot), c = the tile's centre
and reports how many of the EMPTY tiles each form proves empty, with the operation counts the centred form costs per clause.

    python scripts/cull_study.py [model] [size]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import mpr_amd as m
from oracle import orc

name = sys.argv[1] if len(sys.argv) > 1 else "bear"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
tape = m.Tape(m.model(name))
dec = m.decode(tape.data)
T = np.eye(4, dtype=np.float32)
T[3, 2] = 0.3
ref = orc.Frame(tape.data, 3, S, m.colmajor(T, 4), threads=0, keep_pool=False)
tiles = ref.tiles[3]
pos = tiles["position"][tiles["position"] != -1].astype(np.int64)
tps = S // 4
tx, ty, tz = pos % tps, (pos // tps) % tps, pos // (tps * tps)
N = pos.size
print("%s %d^3: %d smallest tiles reach the float pass" % (name, S, N))


def view(fx, fy, fz):
    """the float pass's position arithmetic (reference src/context.cu:707-760), float64"""
    M = T.astype(np.float64)
    w = M[3, 0] * fx + M[3, 1] * fy + M[3, 2] * fz + M[3, 3]
    return ((M[0, 0] * fx + M[0, 1] * fy + M[0, 2] * fz + M[0, 3]) / w, (M[1, 0] * fx + M[1, 1] * fy + M[1, 2] * fz + M[1, 3]) / w,
            (M[2, 0] * fx + M[2, 1] * fy + M[2, 2] * fz + M[2, 3]) / w)


# ---- plain float64 evaluation (truth) ----
def feval(x, y, z):
    slot = {}
    h = dec[0]
    slot[h[1]], slot[h[2]], slot[h[3]] = x, y, z
    for op, o, l, r, imm in dec[1:-1]:
        a = slot.get(l)
        b = slot.get(r)
        with np.errstate(all="ignore"):
            if op == "SQUARE_LHS": v = a * a
            elif op == "SQRT_LHS": v = np.sqrt(a)
            elif op == "NEG_LHS": v = -a
            elif op == "SIN_LHS": v = np.sin(a)
            elif op == "COS_LHS": v = np.cos(a)
            elif op == "ASIN_LHS": v = np.arcsin(a)
            elif op == "ACOS_LHS": v = np.arccos(a)
            elif op == "ATAN_LHS": v = np.arctan(a)
            elif op == "EXP_LHS": v = np.exp(a)
            elif op == "ABS_LHS": v = np.abs(a)
            elif op == "LOG_LHS": v = np.log(a)
            elif op == "ADD_LHS_IMM": v = a + imm
            elif op == "ADD_LHS_RHS": v = a + b
            elif op == "MUL_LHS_IMM": v = a * imm
            elif op == "MUL_LHS_RHS": v = a * b
            elif op == "MIN_LHS_IMM": v = np.minimum(a, imm)
            elif op == "MIN_LHS_RHS": v = np.minimum(a, b)
            elif op == "MAX_LHS_IMM": v = np.maximum(a, imm)
            elif op == "MAX_LHS_RHS": v = np.maximum(a, b)
            elif op == "SUB_LHS_IMM": v = a - imm
            elif op == "SUB_IMM_RHS": v = imm - b
            elif op == "SUB_LHS_RHS": v = a - b
            elif op == "DIV_LHS_IMM": v = a / imm
            elif op == "DIV_IMM_RHS": v = imm / b
            elif op == "DIV_LHS_RHS": v = a / b
            else: raise SystemExit("opcode " + op)
        slot[o] = v
    return slot[dec[-1][1]]


sub = np.arange(4)
vx, vy, vz = np.meshgrid(sub, sub, sub, indexing="ij")
px = (tx[:, None] * 4 + vx.ravel()[None, :] + 0.5) / S * 2 - 1
py = (ty[:, None] * 4 + vy.ravel()[None, :] + 0.5) / S * 2 - 1
pz = (tz[:, None] * 4 + vz.ravel()[None, :] + 0.5) / S * 2 - 1
fv = feval(*view(px, py, pz))
empty = ~(fv < 0).any(axis=1)
print("  truth: %d (%.1f %%) contain a voxel of the shape, %d (%.1f %%) are empty" % ((~empty).sum(), 100 * (~empty).mean(), empty.sum(), 100 * empty.mean()))


# ---- intervals and interval gradients ----
class IV:
    __slots__ = ("lo", "hi")

    def __init__(self, lo, hi):
        self.lo, self.hi = lo, hi


def c_(k):
    return IV(np.full(N, k), np.full(N, k))


def iadd(a, b): return IV(a.lo + b.lo, a.hi + b.hi)
def isub(a, b): return IV(a.lo - b.hi, a.hi - b.lo)
def ineg(a): return IV(-a.hi, -a.lo)
def imul(a, b):
    p = np.stack([a.lo * b.lo, a.lo * b.hi, a.hi * b.lo, a.hi * b.hi])
    return IV(np.nanmin(p, axis=0), np.nanmax(p, axis=0))
def iscale(a, k): return IV(a.lo * k, a.hi * k) if k >= 0 else IV(a.hi * k, a.lo * k)
def irecip(b):
    bad = (b.lo <= 0) & (b.hi >= 0)
    with np.errstate(all="ignore"):
        return IV(np.where(bad, -np.inf, 1 / b.hi), np.where(bad, np.inf, 1 / b.lo))
def idiv(a, b): return imul(a, irecip(b))
def isq(a):
    lo = np.where((a.lo <= 0) & (a.hi >= 0), 0.0, np.minimum(a.lo * a.lo, a.hi * a.hi))
    return IV(lo, np.maximum(a.lo * a.lo, a.hi * a.hi))
def hull(a, b): return IV(np.minimum(a.lo, b.lo), np.maximum(a.hi, b.hi))
def mono(fn, a):
    with np.errstate(all="ignore"):
        return IV(fn(a.lo), fn(a.hi))


OPS = {"value": 0, "grad": 0}          # interval operations (an interval product counted as 4) of the plain and of the gradient part


def walk(X, Y, Z, with_grad):
    one, zero = c_(1.0), c_(0.0)
    slot = {}
    h = dec[0]
    slot[h[1]] = (X, (one, zero, zero))
    slot[h[2]] = (Y, (zero, one, zero))
    slot[h[3]] = (Z, (zero, zero, one))
    for op, o, l, r, imm in dec[1:-1]:
        A, dA = slot.get(l, (None, None))
        B, dB = slot.get(r, (None, None))
        g = None
        G = with_grad
        if op == "SQUARE_LHS":
            v = isq(A); OPS["value"] += 2
            if G: g = tuple(iscale(imul(A, d), 2.0) for d in dA); OPS["grad"] += 3 * 4
        elif op == "SQRT_LHS":
            v = mono(lambda t: np.sqrt(np.maximum(t, 0)), A); OPS["value"] += 2
            if G:
                k = irecip(iscale(v, 2.0)); g = tuple(imul(d, k) for d in dA); OPS["grad"] += 2 + 3 * 4
        elif op == "NEG_LHS":
            v = ineg(A)
            if G: g = tuple(ineg(d) for d in dA)
        elif op in ("SIN_LHS", "COS_LHS"):
            v = IV(np.full(N, -1.0), np.full(N, 1.0))
            if G: g = tuple(imul(d, v) for d in dA); OPS["grad"] += 3 * 4
        elif op == "EXP_LHS":
            v = mono(np.exp, A); OPS["value"] += 2
            if G: g = tuple(imul(d, v) for d in dA); OPS["grad"] += 3 * 4
        elif op == "LOG_LHS":
            v = mono(lambda t: np.log(np.maximum(t, 1e-300)), A); OPS["value"] += 2
            if G: k = irecip(A); g = tuple(imul(d, k) for d in dA); OPS["grad"] += 2 + 3 * 4
        elif op == "ABS_LHS":
            lo = np.where((A.lo <= 0) & (A.hi >= 0), 0.0, np.minimum(np.abs(A.lo), np.abs(A.hi)))
            v = IV(lo, np.maximum(np.abs(A.lo), np.abs(A.hi)))
            if G:
                s = IV(np.where(A.hi < 0, -1.0, np.where(A.lo > 0, 1.0, -1.0)), np.where(A.hi < 0, -1.0, np.where(A.lo > 0, 1.0, 1.0)))
                g = tuple(imul(d, s) for d in dA); OPS["grad"] += 3 * 4
        elif op == "ADD_LHS_IMM": v = IV(A.lo + imm, A.hi + imm); g = dA; OPS["value"] += 2
        elif op == "SUB_LHS_IMM": v = IV(A.lo - imm, A.hi - imm); g = dA; OPS["value"] += 2
        elif op == "SUB_IMM_RHS":
            v = IV(imm - B.hi, imm - B.lo); OPS["value"] += 2
            if G: g = tuple(ineg(d) for d in dB)
        elif op == "ADD_LHS_RHS":
            v = iadd(A, B); OPS["value"] += 2
            if G: g = tuple(iadd(a, b) for a, b in zip(dA, dB)); OPS["grad"] += 3 * 2
        elif op == "SUB_LHS_RHS":
            v = isub(A, B); OPS["value"] += 2
            if G: g = tuple(isub(a, b) for a, b in zip(dA, dB)); OPS["grad"] += 3 * 2
        elif op == "MUL_LHS_IMM":
            v = iscale(A, imm); OPS["value"] += 2
            if G: g = tuple(iscale(d, imm) for d in dA); OPS["grad"] += 3 * 2
        elif op == "DIV_LHS_IMM":
            v = iscale(A, 1.0 / imm); OPS["value"] += 2
            if G: g = tuple(iscale(d, 1.0 / imm) for d in dA); OPS["grad"] += 3 * 2
        elif op == "MUL_LHS_RHS":
            v = imul(A, B); OPS["value"] += 4
            if G: g = tuple(iadd(imul(a, B), imul(b, A)) for a, b in zip(dA, dB)); OPS["grad"] += 3 * 10
        elif op == "DIV_LHS_RHS":
            v = idiv(A, B); OPS["value"] += 6
            if G:
                k = irecip(B)
                g = tuple(imul(isub(a, imul(v, b)), k) for a, b in zip(dA, dB)); OPS["grad"] += 2 + 3 * 10
        elif op == "DIV_IMM_RHS":
            k = irecip(B); v = iscale(k, imm); OPS["value"] += 4
            if G: g = tuple(ineg(imul(imul(v, b), k)) for b in dB); OPS["grad"] += 3 * 8
        elif op in ("MIN_LHS_RHS", "MAX_LHS_RHS", "MIN_LHS_IMM", "MAX_LHS_IMM"):
            if op.endswith("IMM"):
                B, dB = c_(imm), (zero, zero, zero)
            mn = op.startswith("MIN")
            v = IV(np.minimum(A.lo, B.lo), np.minimum(A.hi, B.hi)) if mn else IV(np.maximum(A.lo, B.lo), np.maximum(A.hi, B.hi))
            OPS["value"] += 2
            if G:
                a_wins = (A.hi < B.lo) if mn else (A.lo > B.hi)
                b_wins = (B.hi < A.lo) if mn else (B.lo > A.hi)
                g = tuple(IV(np.where(a_wins, a.lo, np.where(b_wins, b.lo, np.minimum(a.lo, b.lo))),
                             np.where(a_wins, a.hi, np.where(b_wins, b.hi, np.maximum(a.hi, b.hi)))) for a, b in zip(dA, dB))
                OPS["grad"] += 3 * 2
        else:
            raise SystemExit("opcode " + op)
        slot[o] = (v, g)
    return slot[dec[-1][1]]


# tile boxes in the model's coordinates: the view's map is monotone in each screen axis on these frames (identity + perspective in z)
def screen(t, k):
    return (t * 4 + k) / S * 2 - 1


cx = [view(screen(tx, a), screen(ty, b), screen(tz, d)) for a in (0, 4) for b in (0, 4) for d in (0, 4)]
X = IV(np.min([c[0] for c in cx], axis=0), np.max([c[0] for c in cx], axis=0))
Y = IV(np.min([c[1] for c in cx], axis=0), np.max([c[1] for c in cx], axis=0))
Z = IV(np.min([c[2] for c in cx], axis=0), np.max([c[2] for c in cx], axis=0))
F, dF = walk(X, Y, Z, True)
nat_empty = F.lo > 0
mid = [(v.lo + v.hi) / 2 for v in (X, Y, Z)]
rad = [(v.hi - v.lo) / 2 for v in (X, Y, Z)]
fc = feval(*mid)
spread = sum(np.maximum(np.abs(d.lo), np.abs(d.hi)) * r for d, r in zip(dF, rad))
mv_lo = fc - spread
mv_empty = mv_lo > 0
both = nat_empty | mv_empty
E = empty.sum()
print("  natural extension over the 4^3 tile proves empty:        %7d of the %d empty tiles (%.1f %%)   [the reference's test: these are the tiles it left]"
      % ((nat_empty & empty).sum(), E, 100 * (nat_empty & empty).sum() / max(E, 1)))
print("  centred form (interval gradient over the tile) proves:   %7d (%.1f %%)" % ((mv_empty & empty).sum(), 100 * (mv_empty & empty).sum() / max(E, 1)))
print("  unsound culls (a tile with a voxel inside proved empty): natural %d, centred %d" % ((nat_empty & ~empty).sum(), (mv_empty & ~empty).sum()))
print("  interval operations per walk: value part %d, gradient part %d (x%.1f)" % (OPS["value"], OPS["grad"], (OPS["value"] + OPS["grad"]) / max(OPS["value"], 1)))
left = N - (both & empty).sum()
print("  float-pass tiles left after a centred cull: %d of %d (%.2f)" % (left, N, left / N))
fin = np.isfinite(spread)
with np.errstate(all="ignore"):
    print("  gradient bound finite for %.1f %% of the tiles; there: median sum_i |dF/dx_i|max r_i = %.4f, median |f(c)| = %.4f, median width of the natural extension = %.4f (a voxel is %.4f wide)"
          % (100 * fin.mean(), np.median(spread[fin]), np.median(np.abs(fc[fin])), np.median((F.hi - F.lo)[fin]), 2.0 / S))
    g1 = sum(np.maximum(np.abs(d.lo), np.abs(d.hi)) for d in dF)
    print("  median 1-norm of the gradient bound over a tile: %.2f (a distance field has <= 1.73)" % np.median(g1[fin]))
