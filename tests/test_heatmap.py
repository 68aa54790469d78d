"""Context::render2D_heatmap / render3D_heatmap (inc/context.hpp:51-58, src/context.cu:1984-2339):
amortised work per pixel.

2-D heatmaps are a deterministic sum per pixel (one tile per stage covers a pixel, stages run in
order), so the GPU must reproduce the oracle bit for bit.  In 3-D several tiles of one pixel
column add concurrently and tiles can be culled by a neighbour's fill while they run — upstream
has the same races — so those are checked against a lower and an upper bound the oracle computes over all execution orders."""
import numpy as np
import pytest

from conftest import view2, view3


def root_words(tape):
    return int(tape.length) - 2


# ---------------------------------------------------------------- oracle alone (CPU)
@pytest.mark.parametrize("name,S", [("circle", 128), ("hello_world", 128)])
def test_oracle_heatmap_2d_floor_and_structure(mpr, orc, tapes, name, S):
    tape = tapes(name)
    f = orc.Frame(tape.data, 2, S, mpr.colmajor(view2(), 3), heatmap=True)
    h = f.heatmap
    assert h.shape == (S, S) and h.dtype == np.float32
    # every 64-px tile walks the whole root tape once: (len-2)/64^2 per pixel, over (len-2)
    floor = np.float32(1.0 / 4096.0)
    assert h.min() >= floor
    if name == "circle":
        assert h.min() == floor              # the corners of the frame are empty at the first stage
    # tiles decided at the first stage (64x64 blocks all empty or all filled, no ambiguity
    # below them) carry exactly the floor; pixels evaluated one by one carry more
    per_tile = h.reshape(S // 64, 64, S // 64, 64).transpose(0, 2, 1, 3).reshape(-1, 64 * 64)
    flat_tiles = (per_tile == floor).all(axis=1)
    deep_tiles = (per_tile > floor).all(axis=1)
    assert flat_tiles.sum() + deep_tiles.sum() == per_tile.shape[0]     # stage-0 work is per tile
    assert deep_tiles.any()
    # the image of a heatmap frame is the ordinary image
    g = orc.Frame(tape.data, 2, S, mpr.colmajor(view2(), 3))
    assert np.array_equal(f.image, g.image)
    assert g.heatmap is None


def test_oracle_heatmap_2d_pixel_pass_share(mpr, orc, tapes):
    """a pixel that reaches the per-pixel pass carries at least one clause of float work: half a
    thread's walk (src/context.cu:1977-1980)"""
    tape = tapes("circle")
    S = 128
    f = orc.Frame(tape.data, 2, S, mpr.colmajor(view2(), 3), heatmap=True)
    vt = f.tiles[3]
    assert vt.size > 0
    n = root_words(tape)
    tps = S // 8
    for pos in vt["position"][:16]:
        x0, y0 = (pos % tps) * 8, (pos // tps) * 8
        blk = f.heatmap[y0:y0 + 8, x0:x0 + 8]
        assert (blk >= np.float32(1.0 / 4096.0) + np.float32(0.5 / n) * np.float32(0.99)).all()
        assert np.unique(blk).size == 1          # one tape per smallest tile -> one value per tile


def test_oracle_heatmap_3d_is_threading_invariant_within_rounding(mpr, orc, tapes):
    tape = tapes("sphere")
    a = orc.Frame(tape.data, 3, 128, mpr.colmajor(view3(), 4), heatmap=True, threads=1).heatmap
    b = orc.Frame(tape.data, 3, 128, mpr.colmajor(view3(), 4), heatmap=True, threads=1).heatmap
    assert np.array_equal(a, b)                  # serial: deterministic
    assert a.min() >= np.float32(2.0 / 4096.0) * np.float32(0.999)    # two 64-px tiles deep at 128^3


# ---------------------------------------------------------------- GPU vs oracle
@pytest.mark.gpu
@pytest.mark.parametrize("name,S", [("circle", 128), ("two_spheres", 256), ("hello_world", 256), ("prospero", 256),
                                    ("involute_gear_2d", 512), ("trig", 256), ("architecture", 256)])
def test_heatmap_2d_bit_exact(mpr, orc, tapes, name, S):
    tape = tapes(name)
    ctx = mpr.Context(S)
    heat = ctx.render2D_heatmap(tape, view2(), 0.0)
    ref = orc.Frame(tape.data, 2, S, mpr.colmajor(view2(), 3), heatmap=True, threads=0)
    assert np.array_equal(ctx.stages[3].filled, ref.image)
    assert np.array_equal(heat.view(np.uint32), ref.heatmap.view(np.uint32)), "heat differs at %d pixels (max |d| %g)" % (
        int((heat != ref.heatmap).sum()), float(np.abs(heat - ref.heatmap).max()))
    # and an ordinary frame afterwards is unaffected by the heatmap state
    ctx.render2D(tape, view2(), 0.0)
    assert np.array_equal(ctx.stages[3].filled, ref.image)
    ctx.close()


@pytest.mark.gpu
def test_heatmap_2d_general_view_and_z(mpr, orc, tapes):
    m = np.array([[0.9, 0.2, 0.1], [-0.15, 1.1, -0.05], [0.1, 0.05, 1.0]], dtype=np.float32)
    tape = tapes("hello_world")
    ctx = mpr.Context(256)
    heat = ctx.render2D_heatmap(tape, m, 0.1)
    ref = orc.Frame(tape.data, 2, 256, mpr.colmajor(m, 3), z=0.1, heatmap=True, threads=0)
    assert np.array_equal(heat.view(np.uint32), ref.heatmap.view(np.uint32))
    ctx.close()


def heat_bounds(mpr, orc, tape, S, threads=0):
    lo = orc.Frame(tape.data, 3, S, mpr.colmajor(view3(), 4), heatmap="lower", threads=threads, keep_pool=False)
    hi = orc.Frame(tape.data, 3, S, mpr.colmajor(view3(), 4), heatmap="upper", threads=threads, keep_pool=False)
    return lo, hi


def assert_between(heat, lo, hi):
    """lo <= heat <= hi up to the rounding of differently ordered float sums"""
    slack = 1e-5 * hi + 1e-7
    below, above = heat < lo - slack, heat > hi + slack
    assert not below.any(), "%d pixels below the lower bound (worst %g)" % (int(below.sum()), float((lo - heat).max()))
    assert not above.any(), "%d pixels above the upper bound (worst %g)" % (int(above.sum()), float((heat - hi).max()))


@pytest.mark.parametrize("name,S", [("sphere", 128), ("two_spheres", 128)])
def test_oracle_heatmap_3d_bounds_enclose_any_execution(mpr, orc, tapes, name, S):
    tape = tapes(name)
    lo, hi = heat_bounds(mpr, orc, tape, S, threads=1)
    assert (lo.heatmap <= hi.heatmap).all()
    assert (lo.heatmap < hi.heatmap).any()         # the race is real: the bounds are not vacuous
    for threads in (1, 4):
        f = orc.Frame(tape.data, 3, S, mpr.colmajor(view3(), 4), heatmap=True, threads=threads)
        assert_between(f.heatmap, lo.heatmap, hi.heatmap)
        assert np.array_equal(f.image, lo.image) and np.array_equal(f.image, hi.image)


@pytest.mark.gpu
@pytest.mark.parametrize("name,S", [("sphere", 128), ("hello_world", 256), ("bear", 256), ("architecture", 256),
                                    ("trig", 128)])
def test_heatmap_3d_within_oracle_bounds(mpr, orc, tapes, name, S):
    tape = tapes(name)
    ctx = mpr.Context(S)
    heat = ctx.render3D_heatmap(tape, view3())
    lo, hi = heat_bounds(mpr, orc, tape, S)
    # the frame itself is the ordinary frame, bit for bit
    assert np.array_equal(ctx.stages[3].filled, lo.image)
    assert np.array_equal(ctx.normals, lo.normals)
    # every pixel column is walked by all S/64 top-level tiles
    assert heat.min() >= (S // 64) / 4096.0 * (1 - 1e-5)
    assert_between(heat, lo.heatmap, hi.heatmap)
    # where nothing is timing dependent the value is exact up to the order of the additions
    fixed = lo.heatmap == hi.heatmap
    assert fixed.any()
    assert np.allclose(heat[fixed], lo.heatmap[fixed], rtol=1e-5, atol=1e-7)
    ctx.close()


@pytest.mark.gpu
def test_heatmap_3d_repeatable_totals(mpr, tapes):
    tape = tapes("bear")
    ctx = mpr.Context(256)
    a = ctx.render3D_heatmap(tape, view3())
    b = ctx.render3D_heatmap(tape, view3())
    sa, sb = float(a.astype(np.float64).sum()), float(b.astype(np.float64).sum())
    assert abs(sa - sb) <= 0.1 * sa
    ctx.close()
