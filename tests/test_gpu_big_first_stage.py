"""First stages of tapes beyond the generators' 24 slots / 64 min / max clauses on the tape's own walks as host-generated code in the kernel whose
interpreter keeps 93 slots in registers: the loose forward walk (csrc/interval_gen.hpp: IW_FIRST_MASKS), which records its choices the way
the interpreter does, and the backward walk that reads them (csrc/tile_gen.hpp: tile_gen_build_big_backward).  architecture from 1536^3 on
by default (32 768 first-stage tiles at 2048^3: past the level-parallel kernel's limit; 0.33 ms on the interpreter, 0.215 ms with the
generated forward walk, 0.13 ms with both), any size with MPR_TILE_GEN_BIG_TILES=1.  The walk decides no more than the reference's: the tapes it
pushes are supersets, the frame's heights and normals the oracle's, and a reader gets the frame rendered again the reference's way."""
import ctypes

import numpy as np
import pytest

from conftest import view2, view3
from helpers import check_default_path, compare_reader_frame

pytestmark = pytest.mark.gpu


def redo_counts(mpr, ctx):
    out = (ctypes.c_uint32 * 2)()
    mpr.lib().mpr_debug_redo_counts.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    mpr.lib().mpr_debug_redo_counts(ctx._h, out)
    return int(out[0]), int(out[1])


@pytest.mark.parametrize("name,dim,S", [("architecture", 3, 256), ("architecture", 3, 512), ("prospero", 2, 512), ("prospero", 2, 1024)])
def test_first_stage_on_the_loose_walk_matches_the_oracle(mpr, orc, tapes, name, dim, S, monkeypatch):
    monkeypatch.setenv("MPR_TILE_GEN_BIG_TILES", "1")
    monkeypatch.setenv("MPR_DEBUG_REDO", "1")
    tape = tapes(name)
    mat = view3() if dim == 3 else view2()
    ref = orc.Frame(tape.data, dim, S, mpr.colmajor(mat, dim + 1), threads=0)
    ctx = mpr.Context(S)
    for _ in range(3):
        if dim == 3:
            ctx.render3D(tape, mat)
            assert np.array_equal(ctx.normals, ref.normals)
        else:
            ctx.render2D(tape, mat)
        assert np.array_equal(ctx.image, ref.filled[3])
        assert ctx.tile_stage_forms().startswith("0:loosefwd+genbwd"), ctx.tile_stage_forms()
    walks, fell_back = redo_counts(mpr, ctx)
    assert walks > 0 and fell_back == 0, (walks, fell_back)      # (nothing in these views leaves a routine's domain)
    ctx.close()
    check_default_path(mpr, ref, tape, dim, S, mat)              # (... and both renderings of the frame compared on the device)
    if dim == 3:
        # a reader after such frames: the reference's tiles and tapes, stage by stage
        ctx, _ = compare_reader_frame(mpr, orc, tape, S, mat, ref=ref, frames=2)
        ctx.close()


def test_wavefronts_that_leave_a_domain_fall_back_on_the_interpreter(mpr, orc, monkeypatch):
    """sqrt of a value that is negative in the far third of the view (a wavefront of the first stage at 512^3 is one z layer of 64 tiles):
    the loose walk raises its flag there (the reference's NaN is near) and the wavefront runs the interpreter's forward walk instead —
    same frame."""
    X, Y, Z = mpr.Tree.X(), mpr.Tree.Y(), mpr.Tree.Z()
    terms = [(X - (i % 13) * 0.11 + 0.6) * (Y + (i % 7) * 0.13 - 0.4) + Z * (0.01 * i) for i in range(40)]
    prod = terms[0]
    for s_ in terms[1:]:
        prod = mpr.tmax(prod * 0.5, s_)
    total = terms[0]
    for s_ in terms[1:]:
        total = total + s_
    t = mpr.tmin(mpr.tmin(prod - 0.2, total * 0.01 - 0.05), mpr.sqrt(Z + 0.25) - 0.6 + X * X)
    tape = mpr.Tape(t)
    assert tape.num_slots > 24
    monkeypatch.setenv("MPR_TILE_GEN_BIG_TILES", "1")
    monkeypatch.setenv("MPR_DEBUG_REDO", "1")
    S = 512
    ref = orc.Frame(tape.data, 3, S, mpr.colmajor(view3(), 4), threads=0)
    ctx = mpr.Context(S)
    for _ in range(2):
        ctx.render3D(tape, view3())
        assert np.array_equal(ctx.image, ref.filled[3]) and np.array_equal(ctx.normals, ref.normals)
        assert ctx.tile_stage_forms().startswith("0:loosefwd+genbwd"), ctx.tile_stage_forms()
    walks, fell_back = redo_counts(mpr, ctx)
    assert 0 < fell_back < walks, (walks, fell_back)
    ctx.close()
    check_default_path(mpr, ref, tape, 3, S, view3())


def test_switched_off(mpr, orc, tapes, monkeypatch):
    monkeypatch.setenv("MPR_TILE_GEN_BIG", "0")
    monkeypatch.setenv("MPR_TILE_GEN_BIG_TILES", "1")
    tape = tapes("architecture")
    ctx = mpr.Context(256)
    ctx.render3D(tape, view3())
    assert "loosefwd" not in ctx.tile_stage_forms()
    ctx.close()


@pytest.mark.parametrize("name,dim,S", [("architecture", 3, 256), ("architecture", 3, 512), ("prospero", 2, 512)])
def test_generated_backward_walk_pushes_the_interpreters_tapes(mpr, orc, tapes, name, dim, S, monkeypatch):
    """The same forward walk (same choices), then the generated backward walk in one context and the interpreter's in another: every
    first-stage tile that pushes must carry the same clause sequence (MPR_DEBUG_RAW_READS: the tiles and tapes of the frame as it ran —
    a reader otherwise gets the frame rendered again the reference's way)."""
    monkeypatch.setenv("MPR_TILE_GEN_BIG_TILES", "1")
    monkeypatch.setenv("MPR_DEBUG_RAW_READS", "1")
    tape = tapes(name)
    mat = view3() if dim == 3 else view2()
    got = []
    for bwd in ("1", "0"):
        monkeypatch.setenv("MPR_TILE_GEN_BIG_BWD", bwd)
        ctx = mpr.Context(S)
        (ctx.render3D if dim == 3 else ctx.render2D)(tape, mat)
        assert ctx.tile_stage_forms().startswith("0:loosefwd+genbwd" if bwd == "1" else "0:interp+loosefwd"), ctx.tile_stage_forms()
        tiles = ctx.stages[0].tiles
        pool = ctx.tape_data
        live = tiles[(tiles["position"] != -1) & (tiles["next"] != -1)]
        order = np.argsort(live["position"])
        length, digest = orc.tiles_digest(pool, live[order])
        got.append((live["position"][order].copy(), length, digest, int((live["tape"] != 0).sum())))
        ctx.close()
    a, b = got
    assert a[3] > 0 and a[3] == b[3]                 # tiles with a tape of their own
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert (a[1] < len(tape.data)).any()             # ... shorter than the root tape


@pytest.mark.parametrize("name,dim,S", [("involute_gear_3d", 3, 256), ("involute_gear_2d", 2, 1024)])
def test_generated_backward_walk_behind_the_interpreters_forward_walk(mpr, orc, tapes, name, dim, S, monkeypatch):
    """Tapes the loose arithmetic does not take (acos / atan: the gears) keep the interpreter's forward walk — the reference's enclosures
    — in a first stage that is not level-parallel (by default: more than 8192 tiles; here: the level-parallel kernel switched off), and
    get the generated backward walk behind it: the same tapes as the interpreter's backward walk pushes, word for word, and the oracle's
    frame."""
    monkeypatch.setenv("MPR_WIDE_STAGE0", "0")
    monkeypatch.setenv("MPR_DEBUG_RAW_READS", "1")
    tape = tapes(name)
    mat = view3() if dim == 3 else view2()
    ref = orc.Frame(tape.data, dim, S, mpr.colmajor(mat, dim + 1), threads=0)
    got = []
    for bwd in ("1", "0"):
        monkeypatch.setenv("MPR_TILE_GEN_BIG_BWD", bwd)
        ctx = mpr.Context(S)
        for _ in range(2):
            (ctx.render3D if dim == 3 else ctx.render2D)(tape, mat)
            assert np.array_equal(ctx.image, ref.filled[3])
            if dim == 3:
                assert np.array_equal(ctx.normals, ref.normals)
        assert ctx.tile_stage_forms().startswith("0:interp+genbwd" if bwd == "1" else "0:interp "), ctx.tile_stage_forms()
        tiles = ctx.stages[0].tiles
        pool = ctx.tape_data
        live = tiles[(tiles["position"] != -1) & (tiles["next"] != -1)]
        order = np.argsort(live["position"])
        length, digest = orc.tiles_digest(pool, live[order])
        got.append((live["position"][order].copy(), length, digest))
        # ... and these ARE the reference's first-stage tiles and tapes (exact enclosures): the oracle's
        rt = ref.tiles[0]
        rlive = rt[(rt["position"] != -1) & (rt["next"] != -1)]
        ro = np.argsort(rlive["position"])
        rlen, rdig = orc.tiles_digest(ref.pool, rlive[ro])
        assert np.array_equal(live["position"][order], rlive["position"][ro]) and np.array_equal(length, rlen) and np.array_equal(digest, rdig)
        ctx.close()
    assert all(np.array_equal(x, y) for x, y in zip(got[0], got[1]))
