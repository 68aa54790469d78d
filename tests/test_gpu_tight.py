"""The last tile stage's SECOND verdict (round 6; csrc/interval_gen.hpp: tight code, csrc/kernels.hpp: TileStageArgs::lean == 2).

The reference's interval sin / cos is [-1, 1] whatever the argument (inc/gpu_interval.hpp:353; the range reduction behind it is dead
code), and every tile a sin / cos has a say in stays ambiguous down to the voxels: 584 644 smallest tiles of bear 1024^3 go to the
float pass, 63 % of which a sound enclosure of sin / cos proves empty or filled (scripts/tight_cull_study.py).  In frames nobody
reads the last tile stage computes that second enclosure beside the reference's walk — which it leaves as it is: decisions, records
— and keeps the tiles it decides out of the float pass.  Held against the oracle here: heights and normals equal, fewer tiles walked;
the arithmetic it rests on, on every float."""
import os

import numpy as np
import pytest

from conftest import view3
from helpers import check_default_path

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fn", ["sin", "cos"])
def test_tight_sin_cos_code_on_every_float(mpr, fn):
    """Every float x as [x, x], [x, x + w] (w < 8) and the interval to a scrambled copy of its bits through the tight code of one
    clause on the chip: the second enclosure holds the float pass's own sinf / cosf at the ends, the middle and next to every
    multiple of pi / 2 inside; the first one stays [-1, 1].  And what v_sin_f32 / v_cos_f32 are off by at most (|x| <= 1024) beyond
    the |x| 2^-22 their argument's roundings account for: the constant padding in csrc/interval_gen.cpp (TIGHT_TRIG_EPS = 2^-17) is
    four times that or more (the float pass's own sinf / cosf are within 2^-23 of the real functions: tests/test_fmath.py)."""
    r = mpr.dev_tight_trig(fn == "sin")
    print(fn, r)
    assert r["tested"] > 3 * ((1 << 32) - (1 << 25)) and r["bad"] == 0, (fn, r, hex(r["example"]))
    assert r["asked_for_exact"] <= 12, r                   # ([inf, inf] and [-inf, -inf] in each of the three forms: no width)
    assert r["hw_error"] <= 2.0 ** -19, r
    assert r["narrow"] > 50 * max(r["wide"], 1), r


def frames(mpr, tape, S, mat, n=3, tight=True):
    if not tight:
        os.environ["MPR_TILE_TIGHT"] = "0"
    try:
        ctx = mpr.Context(S)
    finally:
        os.environ.pop("MPR_TILE_TIGHT", None)
    out = []
    for _ in range(n):
        ctx.render3D(tape, mat)
        out.append((ctx.image.copy(), ctx.normals.copy(), ctx.tile_stage_forms(), ctx.frame_tiles()[2], ctx.float_kernel()))
    ctx.close()
    return out


@pytest.mark.parametrize("name,S", [("bear", 256), ("bear", 512), ("trig_blend", 256), ("trig_blend", 512)])
def test_frames_with_the_second_verdict_equal_the_oracle_and_walk_fewer_tiles(mpr, orc, tapes, name, S):
    tape = tapes(name)
    ref = orc.Frame(tape.data, 3, S, mpr.colmajor(view3(), 4), threads=0, keep_pool=False)
    assert (ref.image > 0).sum() > 1000
    with_it = frames(mpr, tape, S, view3())
    without = frames(mpr, tape, S, view3(), tight=False)
    for image, normals, forms, walked, kernel in with_it + without:
        assert np.array_equal(image, ref.image), (forms, int((image != ref.image).sum()))
        assert np.array_equal(normals, ref.normals), (forms, int((normals != ref.normals).sum()))
    # frames from the second on take the form that pushes nothing from the last stage (the first measures the tapes)
    assert all("+lean+tight" in f[2] for f in with_it[1:]), [f[2] for f in with_it]
    assert not any("+tight" in f[2] for f in without), [f[2] for f in without]
    a, b = with_it[-1][3], without[-1][3]
    assert b == ref.counters["voxel_tiles"] or b > 0
    assert a < 0.85 * b, (name, S, a, b)
    # ... and the whole default path once more through the suite's usual check, CTX_PARANOID included
    check_default_path(mpr, ref, tape, 3, S, view3())


@pytest.mark.parametrize("seed", range(6))
def test_general_views_with_the_second_verdict(mpr, orc, tapes, seed):
    """rotated, sheared, zoomed, perspective views of the two shapes with sin / cos"""
    rng = np.random.default_rng(600 + seed)
    name = ["bear", "trig_blend"][seed % 2]
    S = [128, 256][(seed // 2) % 2]
    A = np.eye(4, dtype=np.float32)
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    A[:3, :3] = (q * rng.uniform(0.7, 1.4)).astype(np.float32)
    A[:3, :3] += rng.uniform(-0.15, 0.15, (3, 3)).astype(np.float32)
    A[:3, 3] = rng.uniform(-0.2, 0.2, 3).astype(np.float32)
    A[3, rng.integers(0, 3)] = np.float32(rng.uniform(-0.3, 0.3))
    tape = tapes(name)
    ref = orc.Frame(tape.data, 3, S, mpr.colmajor(A, 4), threads=0, keep_pool=False)
    for image, normals, forms, walked, kernel in frames(mpr, tape, S, A):
        assert np.array_equal(image, ref.image), (seed, forms, int((image != ref.image).sum()))
        assert np.array_equal(normals, ref.normals), (seed, forms, int((normals != ref.normals).sum()))
    ctx = mpr.Context(S, flags=mpr.CTX_PARANOID)
    for _ in range(3):
        ctx.render3D(tape, A)
    seen, again, cells = ctx.paranoid_stats()
    assert seen == 3 and cells == 0, (seed, seen, again, cells)
    ctx.close()


@pytest.mark.parametrize("name,S", [("bear", 512), ("trig_blend", 256)])
def test_tiles_the_float_pass_walks(mpr, orc, tapes, name, S, monkeypatch):
    """What the bench line's algorithmic bytes are counted over (bench.py: roofline.units): the tiles the float pass lists after the
    second verdict, and those of them it walks — the others it finds hidden behind the heights drawn so far (development counter,
    MPR_DEBUG_WALKED=1)."""
    tape = tapes(name)
    ref = orc.Frame(tape.data, 3, S, mpr.colmajor(view3(), 4), threads=0, keep_pool=False)
    monkeypatch.setenv("MPR_DEBUG_WALKED", "1")
    ctx = mpr.Context(S)
    for k in range(4):
        ctx.render3D(tape, view3())
        assert np.array_equal(ctx.image, ref.image) and np.array_equal(ctx.normals, ref.normals), k
    monkeypatch.setenv("MPR_VOXEL_FP", "0")
    old = mpr.Context(S)
    for k in range(4):
        old.render3D(tape, view3())
        assert np.array_equal(old.image, ref.image) and np.array_equal(old.normals, ref.normals), k
    assert ctx.float_kernel() == "k_eval_voxels_gen_fp<3>" and old.float_kernel() == "k_eval_voxels_gen<3>"
    listed, walked, walked_old = ctx.frame_tiles()[2], ctx.tiles_walked(), old.tiles_walked()
    assert listed == old.frame_tiles()[2]
    # by footprint segments no more tiles are walked than tile by tile in z order (a tile behind a surface waits for the one in front)
    assert 0 < walked <= walked_old * 1.02 <= listed * 1.02 and listed < ref.counters["voxel_tiles"], (walked, walked_old, listed, ref.counters["voxel_tiles"])
    print("%s %d: the reference lists %d tiles, the second verdict leaves %d; %d walked by segments, %d tile by tile" % (
        name, S, ref.counters["voxel_tiles"], listed, walked, walked_old))
    ctx.close()
    old.close()


@pytest.mark.parametrize("seed,size", [(41777, 3), (48461, 8), (48813, 3), (50609, 3), (52835, 3)])
def test_filled_tiles_of_the_bottom_layer_stay_with_the_float_pass(mpr, orc, seed, size):
    """scripts/paranoid_sweep.py over seeds 40000..53999 (round 6, profiles/r06_paranoid_sweep.txt) against the first version of the
    second verdict: five shapes with 1 to 28 pixels at height 0 where the oracle has 3.  A FILLED tile's height is its z index in its
    level's image, and 0 there is "nothing" (reference src/context.cu:664-692: copy_filled): the reference's own filled tiles of the
    bottom layer draw nothing, while a tile of that layer it leaves AMBIGUOUS is drawn voxel by voxel, up to height 3.  A tile of the
    bottom layer the second verdict proves filled therefore stays with the float pass."""
    from test_gpu_fuzz_shapes import fuzz_tape
    tape = fuzz_tape(mpr, seed, size)
    rng = np.random.default_rng(seed * 7 + size)
    S = int(rng.choice([128, 256]))
    assert rng.random() < 0.5                        # (the sweep's view for these seeds: the benchmark's)
    ref = orc.Frame(tape.data, 3, S, mpr.colmajor(view3(), 4), threads=0, keep_pool=False)
    for image, normals, forms, walked, kernel in frames(mpr, tape, S, view3()):
        assert "+tight" in forms, forms
        assert np.array_equal(image, ref.image), (forms, int((image != ref.image).sum()))
        assert np.array_equal(normals, ref.normals), (forms, int((normals != ref.normals).sum()))


@pytest.mark.parametrize("name,S", [("bear", 256), ("bear", 512), ("trig_blend", 256)])
def test_frames_that_leave_the_references_tiles_and_tapes_take_the_second_verdict_in_a_launch_of_its_own(mpr, orc, tapes, name, S, monkeypatch):
    """MPR_LAST_STAGE_PUSH=1 (bench.py: full_frames): every stage from the 64^3 tiles down on the reference's enclosures, every tape
    pushed, the reference's list of smallest tiles made.  Behind the last stage the tight code runs once more
    (TileStageArgs::verdict_only: '+verdict'): the lists, tapes and records stay the reference's — what a reader gets is held against
    the oracle tile by tile and tape by tape — and the tiles the second verdict decides are not walked by the float pass."""
    from helpers import active_positions
    tape = tapes(name)
    ref = orc.Frame(tape.data, 3, S, mpr.colmajor(view3(), 4), threads=0)
    monkeypatch.setenv("MPR_DEBUG_WALKED", "1")
    fast = mpr.Context(S)                        # (a frame nobody reads: the second verdict inside its last stage)
    monkeypatch.setenv("MPR_LAST_STAGE_PUSH", "1")
    ctx = mpr.Context(S)
    monkeypatch.setenv("MPR_TILE_TIGHT", "0")
    plain = mpr.Context(S)
    for k in range(3):
        for c in (ctx, plain, fast):
            c.render3D(tape, view3())
            assert np.array_equal(c.image, ref.image), (k, c.tile_stage_forms(), int((c.image != ref.image).sum()))
            assert np.array_equal(c.normals, ref.normals), (k, c.tile_stage_forms(), int((c.normals != ref.normals).sum()))
    assert ctx.tile_stage_forms().endswith("+verdict") and "+verdict" not in plain.tile_stage_forms(), (ctx.tile_stage_forms(), plain.tile_stage_forms())
    # the float pass: by footprint segments made BESIDE the reference's list (context.hip: vox_fp_beside) / the list tile by tile
    assert ctx.float_kernel() == "k_eval_voxels_gen_fp<3>" and plain.float_kernel() == "k_eval_voxels_gen<3>", (ctx.float_kernel(), plain.float_kernel())
    assert ctx.frame_tiles()[2] == plain.frame_tiles()[2] == ref.counters["voxel_tiles"]         # the reference's list
    assert 0 < ctx.tiles_walked() < (0.9 if S >= 512 else 1.0) * plain.tiles_walked(), (ctx.tiles_walked(), plain.tiles_walked())
    # ... as many as in a frame nobody reads, give or take the order the wavefronts came in (the list is mostly decided tiles by then: lanes
    # without a tile must not cost the wavefront its verdict — they did, kernels.hip: "a lane without a tile walks along on the leader's")
    assert "+tight" in fast.tile_stage_forms() and ctx.tiles_walked() < 1.1 * fast.tiles_walked() + 64, (ctx.tiles_walked(), fast.tiles_walked())
    fast.close()
    # the frame's tiles and tapes as it left them (no second rendering: the context's frames are the reference's way already)
    assert ctx.last_stage_pushed()
    pool = ctx.tape_data
    for s in (0, 1, 2):
        g_next, r_next = ctx.stages[s + 1].tiles, ref.tiles[s + 1]
        assert g_next.size == r_next.size, s
        g_live, r_live = g_next[g_next["position"] != -1], r_next[r_next["position"] != -1]
        go, ro = np.argsort(g_live["position"]), np.argsort(r_live["position"])
        assert np.array_equal(g_live["position"][go], r_live["position"][ro]), "survivor sets differ after stage %d" % s
        if g_live.size:
            glen, ghash = orc.tiles_digest(pool, g_live[go])
            rlen, rhash = orc.tiles_digest(ref.pool, r_live[ro])
            assert np.array_equal(glen, rlen) and np.array_equal(ghash, rhash), "shortened tapes differ after stage %d" % s
    for s in (0, 1, 2, 3):
        assert np.array_equal(ctx.stages[s].filled, ref.filled[s]), "filled image of stage %d differs" % s
    ctx.close()
    plain.close()


@pytest.mark.parametrize("switch,absent", [("MPR_TILE_GEN_GUARDS", "+guards"), ("MPR_TILE_GEN_LEAN", "+lean")])
def test_last_stage_without_guards_and_in_the_128_register_kernel(mpr, orc, tapes, switch, absent, monkeypatch):
    """The two switches of the loose last stage that had no test of their own: the forward walk that does not jump
    over what the parent's decisions left dead, and the loose stage in the ordinary kernel (four wavefronts per SIMD, no second
    verdict: that rides in the lean kernel).  The oracle's heights and normals either way."""
    tape = tapes("bear")
    ref = orc.Frame(tape.data, 3, 256, mpr.colmajor(view3(), 4), threads=0, keep_pool=False)
    monkeypatch.setenv(switch, "0")
    ctx = mpr.Context(256)
    for k in range(3):
        ctx.render3D(tape, view3())
        assert np.array_equal(ctx.image, ref.image) and np.array_equal(ctx.normals, ref.normals), (switch, k, ctx.tile_stage_forms())
    last = ctx.tile_stage_forms().split()[-1]
    assert absent not in last and "+loose" in last, ctx.tile_stage_forms()
    ctx.close()
