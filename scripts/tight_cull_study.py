"""How many of the smallest tiles the reference's last tile stage leaves ambiguous does a SOUND sin / cos enclosure decide?


Test / measurement infrastructure: uses the CPU oracle for the frame and for every interval operation.

The reference's interval cos returns [-1, 1] whatever its argument (inc/gpu_interval.hpp:353; the range reduction behind it,
:355-375, is dead code), sin is cos(x - pi/2) on top of it (:377-380).  This script walks the root tape over every tile the
oracle's frame hands to the float pass with the oracle's own interval operations, clause by clause, and ONLY SIN_LHS / COS_LHS
replaced by a float64 monotone-piece enclosure widened by 1e-6, and counts the tiles whose result is then provably positive
(empty) or provably negative (filled).  Also at the level above (the 16^3 tiles the last stage subdivides).

    python scripts/tight_cull_study.py [model] [size] [threads]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

import helpers
import mpr_amd as mpr
from oracle import orc

name = sys.argv[1] if len(sys.argv) > 1 else "bear"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 8
tape = mpr.Tape(mpr.model(name))
T = np.eye(4, dtype=np.float32)
T[3, 2] = 0.3
mat = mpr.colmajor(T, 4)
fr = orc.Frame(tape.data, 3, S, mat, threads=threads, keep_pool=False)
image = fr.filled[3]


def tight_cos(lo, hi):
    """float64, sound: cos is monotone between multiples of pi; extrema at the multiples of pi inside [lo, hi]"""
    lo, hi = lo.astype(np.float64), hi.astype(np.float64)
    with np.errstate(all="ignore"):
        k = np.ceil(lo / np.pi)
        e1, e2, even = k * np.pi <= hi, (k + 1) * np.pi <= hi, np.mod(k, 2) == 0
        mn, mx = np.minimum(np.cos(lo), np.cos(hi)), np.maximum(np.cos(lo), np.cos(hi))
        mx = np.where((e1 & even) | e2, 1.0, mx)
        mn = np.where((e1 & ~even) | e2, -1.0, mn)
        bad = ~np.isfinite(lo) | ~np.isfinite(hi) | (hi - lo >= 2 * np.pi)
        return (np.clip(np.where(bad, -1, mn - 1e-6), -1, 1).astype(np.float32),
                np.clip(np.where(bad, 1, mx + 1e-6), -1, 1).astype(np.float32))


def walk(clauses, axes, tight):
    d = mpr.decode(clauses)
    n = axes[0][0].size
    slots = {}
    head = d[0]
    slots[head[1]], slots[head[2]], slots[head[3]] = axes[0], axes[1], axes[2]
    zero = (np.zeros(n, np.float32), np.zeros(n, np.float32))
    widths = []
    for i in range(1, len(d) - 1):
        nm, out, lhs, rhs, imm = d[i]
        a = slots.get(lhs, zero)
        b = slots.get(rhs, zero)
        if tight and nm == "COS_LHS":
            lo, hi = tight_cos(a[0], a[1])
        elif tight and nm == "SIN_LHS":
            lo, hi = tight_cos(a[0].astype(np.float64) - np.pi / 2, a[1].astype(np.float64) - np.pi / 2)
        else:
            lo, hi, _ = orc.interval_op(mpr.OP[nm], a[0], a[1], b[0], b[1], imm)
        if nm in ("COS_LHS", "SIN_LHS"):
            widths.append((i, nm, float(np.median(a[1] - a[0])), float(np.median(hi - lo))))
        slots[out] = (lo, hi)
    return slots[d[-1][1]], widths


def level(stage, ts):
    tl = fr.tiles[stage]
    pos = tl["position"]
    pos = pos[pos >= 0].astype(np.int64)
    tps = S // ts
    xyz = np.stack([pos % tps, (pos // tps) % tps, pos // tps ** 2], 1)
    axes = helpers.oracle_axes_of_tiles(mpr, orc, xyz, tps, mat)
    (rlo, rhi), _ = walk(tape.data, axes, False)
    (tlo, thi), widths = walk(tape.data, axes, True)
    n = pos.size
    # tiles wholly behind the final heightmap (what the float pass's early-out skips): top voxel <= every pixel's height
    behind = np.zeros(n, bool)
    if ts == 4:
        top = xyz[:, 2] * 4 + 3
        hmin = image.reshape(S // 4, 4, S // 4, 4).min(axis=(1, 3))
        behind = top <= hmin[xyz[:, 1], xyz[:, 0]]
    emp, fil = tlo > 0, thi < 0
    print("%s %d^3, %d^3 tiles the reference leaves ambiguous: %d" % (name, S, ts, n))
    print("  the reference's enclosures on the root tape prove empty %d, filled %d" % (int((rlo > 0).sum()), int((rhi < 0).sum())))
    print("  with a sound sin / cos: empty %d, filled %d  (%.1f %% decided)" % (int(emp.sum()), int(fil.sum()), 100.0 * (emp | fil).mean()))
    if ts == 4:
        full = ~behind
        print("  wholly behind the final heightmap (early-outed already): %d; of the other %d tiles %d are decided, %d (%.0f %%) remain" % (
            int(behind.sum()), int(full.sum()), int(((emp | fil) & full).sum()), int((~(emp | fil) & full).sum()),
            100.0 * (~(emp | fil) & full).sum() / max(1, full.sum())))
    print("  median width of the result: %.3f -> %.3f" % (float(np.median(rhi - rlo)), float(np.median(thi - tlo))))
    for i, nm, wa, wr in widths:
        print("    clause %4d %s: median width of its argument %.4f, of its tight result %.4f" % (i, nm, wa, wr))


level(3, 4)
level(2, 16)
