"""csrc/frame_domain.cpp (host): is a (tape, view) "tame" — does every interval operation stay, over the whole view, where the
reference's interval routines are inclusion-isotone?  Frames of tame views start at the 16^3 tiles without the verification
against the 64^3 tiles that other frames get (tests/test_gpu_render.py:
test_frames_that_start_at_the_16_tiles_are_verified_against_the_64_tiles).  Checked here on the CPU against the oracle's interval
routines: the host's enclosure of every clause over the view holds what the oracle computes for the view's tiles at every level,
and, for tame views, a child tile's interval lies inside its parent's at every clause — the property the shortcut rests on."""
import zlib

import numpy as np
import pytest

from conftest import view3
from helpers import oracle_axes_of_tiles, oracle_walk_tiles


def shapes(mpr):
    X, Y, Z = mpr.Tree.X(), mpr.Tree.Y(), mpr.Tree.Z()
    r = mpr.sqrt(X * X + Y * Y + Z * Z)
    return {
        "sphere": (r - 0.7, True),
        "log_of_positive": (mpr.log(X * X + 0.5) + Y - Z, True),
        "log_reaches_zero": (mpr.log(X * X) + Y, False),
        "log_of_axis": (mpr.log(X) + Y, False),
        "asin_inside": (mpr.asin(X * 0.5) + mpr.acos(Y * 0.5) - Z - 1.0, True),
        "asin_leaves": (mpr.asin(X * 1.5) + Y, False),
        "acos_leaves": (mpr.acos(Y - 0.9) + X, False),
        "divisor_holds_zero": (1.0 / X + Y, False),
        "divisor_positive": (X / (Y * Y + 2.0) + Z, True),
        "division_by_zero": (X / 0.0 + Y, False),
        "sqrt_straddles": (mpr.sqrt(X) - 0.5 + Y * 0.0, False),        # (a tile entirely below zero: NaN)
        "sqrt_of_square": (mpr.sqrt(mpr.square(X) + Y * Y) - 0.5, True),
        "sqrt_of_negative": (mpr.sqrt(X - 3.0) + Y, False),
        "exp_overflows": (mpr.exp(X * 100.0) + Y, False),
        "exp_fits": (mpr.exp(X * 40.0) + Y, True),
        "blend_underflows": (mpr.log(mpr.exp((r - 0.5) * -64.0) + mpr.exp((r - 0.4) * -64.0)) / -64.0, False),
        "blend_gentle": (mpr.log(mpr.exp((r - 0.5) * -8.0) + mpr.exp((r - 0.4) * -8.0)) / -8.0, True),
        "min_max_abs": (mpr.tmax(mpr.tmin(mpr.tabs(X) - 0.3, mpr.tabs(Y) - 0.4), -Z - 0.2) * mpr.atan(Z) + mpr.sin(X * 9.0) * mpr.cos(Y), True),
    }


def test_verdicts(mpr):
    for name, (tree, tame) in shapes(mpr).items():
        tape = mpr.Tape(tree)
        assert tape.frame_is_tame(view3()) == tame, name
        if name != "blend_underflows":          # (the 2-D view is smaller: no corner far enough for the blend's exp to underflow)
            assert tape.frame_is_tame(np.eye(3, dtype=np.float32), dim=2, z=0.25) == tame, name
    # a view whose divisor (the matrix's last row) reaches zero inside the view is not tame whatever the shape
    T = view3(1.5)
    assert not mpr.Tape(shapes(mpr)["sphere"][0]).frame_is_tame(T)
    # the benchmark models: bear's blends underflow over the whole view (its frames are verified), hello_world has nothing to leave
    assert mpr.Tape(mpr.model("hello_world")).frame_is_tame(view3())
    assert not mpr.Tape(mpr.model("bear")).frame_is_tame(view3())


def tiles_of(level_tps, count, rng):
    g = np.arange(level_tps)
    pp = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    if pp.shape[0] > count:
        pp = pp[rng.choice(pp.shape[0], count, replace=False)]
    return pp


@pytest.mark.parametrize("name", ["sphere", "log_of_positive", "asin_inside", "divisor_positive", "sqrt_of_square", "exp_fits", "blend_gentle",
                                  "min_max_abs", "hello_world", "trig"])
@pytest.mark.parametrize("perspective", [0.3, 0.0, -0.45])
def test_enclosure_holds_the_oracles_tiles_and_children_lie_inside_parents(mpr, orc, tapes, name, perspective):
    tape = mpr.Tape(shapes(mpr)[name][0]) if name in shapes(mpr) else tapes(name)
    T = view3(perspective)
    tame, trace = tape.frame_is_tame(T, trace=True)
    assert tame and not np.isnan(trace[1:-1]).any()
    mat = mpr.colmajor(T, 4)
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    walks = {}
    for tps in (1, 2, 8, 32):
        pp = tiles_of(tps, 512, rng)
        d, lo, hi = oracle_walk_tiles(mpr, orc, tape.data, oracle_axes_of_tiles(mpr, orc, pp, tps, mat))
        walks[tps] = (pp, lo, hi)
        body = slice(1, len(d) - 1)
        assert not (np.isnan(lo[body]).any() or np.isnan(hi[body]).any()), "a tame view has no NaN end"
        assert (lo[body] >= trace[body, 0:1]).all() and (hi[body] <= trace[body, 1:2]).all(), "a tile's interval outside the host's enclosure (%d tiles per side)" % tps
    # inclusion-isotone: the 64 children of a tile, clause by clause, inside it
    for tps in (2, 8):
        pp = tiles_of(tps, 64, rng)
        o = np.arange(4)
        off = np.stack(np.meshgrid(o, o, o, indexing="ij"), -1).reshape(-1, 3)
        cp = (pp[:, None, :] * 4 + off[None, :, :]).reshape(-1, 3)
        d, plo, phi = oracle_walk_tiles(mpr, orc, tape.data, oracle_axes_of_tiles(mpr, orc, pp, tps, mat))
        _, clo, chi = oracle_walk_tiles(mpr, orc, tape.data, oracle_axes_of_tiles(mpr, orc, cp, tps * 4, mat))
        body = slice(1, len(d) - 1)
        assert (clo[body] >= np.repeat(plo[body], 64, axis=1)).all() and (chi[body] <= np.repeat(phi[body], 64, axis=1)).all()


def test_where_the_references_routines_are_not_isotone(mpr, orc):
    """... and what the check is there for: asin of an interval that leaves [-1, 1] has a NaN end (reference inc/gpu_interval.hpp:316-324),
    the product with it is [0, 0] (the sign tests of :86-146 are all false), and a child tile whose interval lies inside the domain
    gets a product that does NOT lie inside its parent's."""
    X, Y, Z = mpr.Tree.X(), mpr.Tree.Y(), mpr.Tree.Z()
    tape = mpr.Tape(mpr.asin(X * 1.5) * (Y + 2.0) - Z)
    assert not tape.frame_is_tame(view3())
    mat = mpr.colmajor(view3(), 4)
    pp = np.array([[1, 0, 0]])
    cp = np.array([[4, 0, 0]])          # x in [0, 0.25]: 1.5 x inside the domain; the parent's x in [0, 1] is not
    d, plo, phi = oracle_walk_tiles(mpr, orc, tape.data, oracle_axes_of_tiles(mpr, orc, pp, 2, mat))
    _, clo, chi = oracle_walk_tiles(mpr, orc, tape.data, oracle_axes_of_tiles(mpr, orc, cp, 8, mat))
    names = [c[0] for c in d]
    i = names.index("ASIN_LHS")
    assert np.isnan(phi[i, 0]) and not np.isnan(chi[i, 0])
    j = names.index("MUL_LHS_RHS")
    assert plo[j, 0] == 0.0 and phi[j, 0] == 0.0 and chi[j, 0] > 0.0
