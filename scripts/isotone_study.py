"""Development aid (CPU, oracle only): walk a tape's ROOT clauses with the oracle's interval routines over the tiles of one level
and over their children, and report the clauses where a child's interval does NOT lie inside its parent's — the places where the
reference's interval arithmetic is not inclusion-isotone (frame_domain.hpp).  Usage: isotone_study.py fuzz:SIZE:SEED | MODEL [S]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mpr_amd as mpr
from oracle import orc

orc.lib()
OP = mpr.OP


import helpers   # noqa: E402


def axes_of_tiles(pos, tps, mat):
    return helpers.oracle_axes_of_tiles(mpr, orc, pos, tps, mat)


def walk(clauses, axes):
    return helpers.oracle_walk_tiles(mpr, orc, clauses, axes)


def main():
    what = sys.argv[1]
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    if what.startswith("fuzz:"):
        src = open(os.path.join(ROOT, "tests", "test_gpu_fuzz_shapes.py")).read().split("@pytest.mark.parametrize")[0]
        src = src.replace("from conftest import view2, view3", "").replace("from helpers import check_default_path, compare_frame, compare_reader_frame", "")
        ns = {}
        exec(src, ns)
        _, size, seed = what.split(":")
        tape = ns["fuzz_tape"](mpr, int(seed), int(size))
    else:
        tape = mpr.Tape(mpr.model(what))
    T = np.eye(4, dtype=np.float32)
    T[3, 2] = 0.3
    mat = mpr.colmajor(T, 4)
    print("frame_is_tame:", tape.frame_is_tame(T))
    for level, (tps_p, sub) in enumerate([(S // 64, 4), (S // 16, 4)]):
        g = np.arange(tps_p)
        pp = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
        if pp.shape[0] > 4096:
            pp = pp[np.random.default_rng(1).choice(pp.shape[0], 4096, replace=False)]
        o = np.arange(sub)
        off = np.stack(np.meshgrid(o, o, o, indexing="ij"), -1).reshape(-1, 3)
        cp = (pp[:, None, :] * sub + off[None, :, :]).reshape(-1, 3)
        d, plo, phi = walk(tape.data, axes_of_tiles(pp, tps_p, mat))
        _, clo, chi = walk(tape.data, axes_of_tiles(cp, tps_p * sub, mat))
        plo = np.repeat(plo, sub ** 3, axis=1)
        phi = np.repeat(phi, sub ** 3, axis=1)
        # inside: lo_c >= lo_p and hi_c <= hi_p; a NaN end anywhere counts as "not inside" unless both are NaN
        with np.errstate(invalid="ignore"):
            ok_lo = (clo >= plo) | (np.isnan(clo) & np.isnan(plo))
            ok_hi = (chi <= phi) | (np.isnan(chi) & np.isnan(phi))
        bad = ~(ok_lo & ok_hi)
        bad[0] = bad[-1] = False
        nan_p = np.isnan(plo) | np.isnan(phi)
        print("level %d (%d parents, %d children): clauses with a child outside its parent: %d; parents with a NaN end somewhere: %d" % (
            level, pp.shape[0], cp.shape[0], int(bad.any(axis=1).sum()), int(nan_p.any(axis=0).sum() // sub ** 3)))
        first = {}
        for i in np.flatnonzero(bad.any(axis=1))[:12]:
            j = int(np.flatnonzero(bad[i])[0])
            print("   clause %d %s: parent [%g, %g] child [%g, %g] (%d children)" % (i, d[i], plo[i, j], phi[i, j], clo[i, j], chi[i, j], int(bad[i].sum())))



def quirk_census(tape, S, levels=((64, "64^3"), (16, "16^3"), (4, "4^3"))):
    """tiles whose walk takes a special case of a partial function, by what the tile's own result says about it"""
    T = np.eye(4, dtype=np.float32)
    T[3, 2] = 0.3
    mat = mpr.colmajor(T, 4)
    for px, label in levels:
        tps = S // px
        g = np.arange(tps)
        pp = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
        if pp.shape[0] > 1 << 18:
            pp = pp[np.random.default_rng(1).choice(pp.shape[0], 1 << 18, replace=False)]
        d, lo, hi = walk(tape.data, axes_of_tiles(pp, tps, mat))
        cur = {}
        special = np.zeros(pp.shape[0], bool)
        for i in range(1, len(d) - 1):
            name, out, lhs, rhs, imm = d[i]
            if name in ("LOG_LHS", "SQRT_LHS", "ASIN_LHS", "ACOS_LHS") and lhs in cur:
                alo, ahi = lo[cur[lhs]], hi[cur[lhs]]
                with np.errstate(invalid="ignore"):
                    if name == "LOG_LHS":
                        special |= ~(alo > 0)
                    elif name == "SQRT_LHS":
                        special |= ~(ahi >= 0)
                    else:
                        special |= ~((alo >= -1) & (ahi <= 1))
            with np.errstate(invalid="ignore"):
                special |= np.isnan(lo[i]) | np.isnan(hi[i])
            cur[out] = i
        res = cur[d[-1][1]]
        with np.errstate(invalid="ignore"):
            empty = lo[res] > 0
            filled = ~empty & (hi[res] < 0)
        amb = ~empty & ~filled
        print("%s tiles of %d^3: %d; a special case taken in %d: %d empty, %d filled, %d ambiguous (of %d ambiguous)" % (
            label, S, pp.shape[0], int(special.sum()), int((special & empty).sum()), int((special & filled).sum()), int((special & amb).sum()), int(amb.sum())))


if len(sys.argv) > 3 and sys.argv[3] == "census":
    quirk_census(mpr.Tape(mpr.model(sys.argv[1])), int(sys.argv[2]))
else:
    main()
