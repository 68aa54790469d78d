"""Every BASELINE.json configuration at its FULL size, HIP path against the oracle: stage images,
heightmap, normals, per-stage survivor sets, counters and the shortened tape of every surviving
tile (helpers.compare_frame).  The oracle runs on the GPU box's host cores (tens of seconds for
bear at 1024^3), which is why these live in their own file.

  config 2  prospero render2D 1024^2          reference benchmark/render_2d_table.cpp:50-62
  config 3  involute_gear_2d render2D 4096^2
  config 4  bear render3D 1024^3               reference benchmark/render_3d_table.cpp:48-71
  config 5  architecture render3D 2048^3, single GPU and sharded over three contexts
"""
import numpy as np
import pytest

from conftest import view2, view3
from helpers import check_default_path, compare_frame

pytestmark = pytest.mark.gpu


def test_prospero_1024_full_frame(mpr, orc, tapes):
    cnt, ref = compare_frame(mpr, orc, tapes("prospero"), 2, 1024, view2())
    assert 0 < ref.image.sum() < ref.image.size
    check_default_path(mpr, ref, tapes("prospero"), 2, 1024, view2())


def test_gears_4096_full_frame(mpr, orc, tapes):
    cnt, ref = compare_frame(mpr, orc, tapes("involute_gear_2d"), 2, 4096, view2())
    assert 0 < ref.image.sum() < ref.image.size
    check_default_path(mpr, ref, tapes("involute_gear_2d"), 2, 4096, view2())


def test_bear_1024_full_frame(mpr, orc, tapes):
    cnt, ref = compare_frame(mpr, orc, tapes("bear"), 3, 1024, view3())
    assert (ref.image > 0).sum() > 300000
    assert cnt["voxel_tiles"] > 500000
    kinds = check_default_path(mpr, ref, tapes("bear"), 3, 1024, view3())
    # the root tape's host-generated float walk with the tiles' decisions as bits, by footprint segments (round 6); no tapes from the last tile stage
    assert all(k[0] == "k_eval_voxels_gen_fp<3>" for k in kinds) and [k[1] for k in kinds] == [False, False, False], kinds


def test_architecture_2048_full_frame(mpr, orc, tapes):
    cnt, ref = compare_frame(mpr, orc, tapes("architecture"), 3, 2048, view3())
    assert (ref.image > 0).sum() > 1000000
    kinds = check_default_path(mpr, ref, tapes("architecture"), 3, 2048, view3())
    # architecture's 8^3 tiles shorten their tapes about 1.6x (a sample of groups, frame by frame: 1.5 .. 1.7) — past the
    # threshold (context.hip: `pays`, 1.4) from which the last stage pushes them and the float pass walks each tile's own; a
    # frame whose sample says otherwise takes the group form and pushes nothing.  Either way the oracle's frame (above).
    assert all(k == ("k_eval_voxels_asm<3>", True) or (k[0].startswith("k_eval_voxels_jit_groups") and not k[1]) for k in kinds), kinds
    # its first stage (32 768 tiles: past the level-parallel kernel's limit) on the tape's own walks as host-generated code: the loose forward
    # walk and the backward walk that reads its choices (tests/test_gpu_big_first_stage.py)
    ctx = mpr.Context(2048)
    ctx.render3D(tapes("architecture"), view3())
    assert ctx.tile_stage_forms().startswith("0:loosefwd+genbwd"), ctx.tile_stage_forms()
    assert np.array_equal(ctx.image, ref.filled[3]) and np.array_equal(ctx.normals, ref.normals)
    ctx.close()


def test_architecture_2048_sharded_over_three_contexts(mpr, orc, tapes):
    """BASELINE config 5 in its sharded form on one device: three TileParallelRenderers play ranks
    0..2 (resident plan, asynchronous partial frame, one pack and one unpack launch, the all-gather
    emulated with copies on each context's stream); every rank must end up with the oracle's frame."""
    import torch
    from mpr_amd.multigpu import TileParallelRenderer
    tape = tapes("architecture")
    S, world = 2048, 3
    T = view3()
    ref = orc.Frame(tape.data, 3, S, mpr.colmajor(T, 4), threads=0, keep_pool=False)
    want_h, want_n = ref.image, ref.normals
    ctxs = [mpr.Context(S) for _ in range(world)]
    sends = {}

    def make_buffer(n):
        t = torch.zeros(n, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        return t, t.data_ptr()

    rs = []
    for r in range(world):
        def all_gather(out, inp, r=r):
            sends[r] = (out, inp)
        tpr = TileParallelRenderer(ctxs[r], mpr, r, world, make_buffer, all_gather, dim=3)
        tpr.plan(tape, T)
        rs.append(tpr)
    assert all(np.array_equal(rs[0].owner, t.owner) for t in rs) and rs[0].planned
    assert len(set(rs[0].owner.tolist())) == world
    for r, t in enumerate(rs):
        ctxs[r].render3D_part(tape, T, t.owner, r, blocking=False)
        ctxs[r].pack_planned(t.send_ptr)
    for r, t in enumerate(rs):
        # a rank's partial frame holds its own columns only
        mask = np.kron(t.owner.reshape(S // 64, S // 64) == r, np.ones((64, 64), dtype=bool))
        ctxs[r].sync()
        part = ctxs[r].image
        assert not part[~mask].any() and np.array_equal(part[mask], want_h[mask])
    for r, t in enumerate(rs):
        for o, u in enumerate(rs):
            ctxs[o].sync()
            with torch.cuda.stream(torch.cuda.ExternalStream(ctxs[r].stream)):
                t.recv[o * t.per_rank:(o + 1) * t.per_rank].copy_(u.send)
        ctxs[r].unpack_planned(t.recv_ptr)
        ctxs[r].sync()
        assert np.array_equal(ctxs[r].image, want_h)
        assert np.array_equal(ctxs[r].normals, want_n)
    for c in ctxs:
        c.close()


def test_bear_1024_dealt_to_eight_ranks_on_one_device(mpr, orc, tapes):
    """BASELINE's headline frame in its sharded form, eight ranks played by eight contexts on one device: the default column deal
    (first tile stage's ambiguous tiles per column — no frame rendered in advance), every rank's partial frame packed, the
    all-gather emulated by device copies, unpacked: every rank ends up with the single-device frame, bit for bit, and the deal
    leaves no rank more than 1.6 x its fair share of the smallest tiles."""
    torch = pytest.importorskip("torch")
    from mpr_amd.multigpu import TileParallelRenderer, column_weights
    tape, S, T, world = tapes("bear"), 1024, view3(), 8
    full = mpr.Context(S)
    full.render3D(tape, T)
    want_h, want_n = full.image.copy(), full.normals.copy()
    true_w = column_weights(full.stages[3].tiles, S, 3)
    full.close()
    ctxs = [mpr.Context(S) for _ in range(world)]

    def make_buffer(n):
        t = torch.zeros(n, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        return t, t.data_ptr()

    rs = []
    for r in range(world):
        tpr = TileParallelRenderer(ctxs[r], mpr, r, world, make_buffer, lambda o, i: None, dim=3)
        tpr.plan(tape, T)
        rs.append(tpr)
    assert all(np.array_equal(rs[0].owner, t.owner) for t in rs)
    share = np.array([true_w[rs[0].owner == r].sum() for r in range(world)]) / true_w.sum()
    assert share.max() < 1.6 / world, share
    for frame in range(2):
        for r, t in enumerate(rs):
            ctxs[r].render3D_part(tape, T, t.owner, r, blocking=False)
            ctxs[r].pack_planned(t.send_ptr)
        for r, t in enumerate(rs):
            for o, u in enumerate(rs):
                ctxs[o].sync()
                with torch.cuda.stream(torch.cuda.ExternalStream(ctxs[r].stream)):
                    t.recv[o * t.per_rank:(o + 1) * t.per_rank].copy_(u.send)
            ctxs[r].unpack_planned(t.recv_ptr)
            ctxs[r].sync()
            assert np.array_equal(ctxs[r].image, want_h)
            assert np.array_equal(ctxs[r].normals, want_n)
    for c in ctxs:
        c.close()
