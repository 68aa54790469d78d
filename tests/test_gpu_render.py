"""Frames on the GPU vs the oracle: bit-exact images, normals, per-stage tile sets and
shortened tapes; plus size-independent properties at the BASELINE sizes."""
import hashlib

import numpy as np
import pytest

from conftest import view2, view3
from helpers import compare_frame

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,S", [("circle", 128), ("circle", 256), ("two_spheres", 256), ("ring", 256),
                                    ("hello_world", 256), ("prospero", 256), ("involute_gear_2d", 512),
                                    ("trig", 256), ("architecture", 256), ("smooth", 256)])
def test_render2d_matches_oracle(mpr, orc, tapes, name, S):
    compare_frame(mpr, orc, tapes(name), 2, S, view2())


def test_render2d_general_view(mpr, orc, tapes):
    m = np.array([[0.9, 0.2, 0.1], [-0.15, 1.1, -0.05], [0.1, 0.05, 1.0]], dtype=np.float32)
    compare_frame(mpr, orc, tapes("hello_world"), 2, 256, m, z=0.1)


@pytest.mark.parametrize("name,S", [("sphere", 128), ("two_spheres", 256), ("hello_world", 256), ("bear", 256),
                                    ("architecture", 256), ("involute_gear_3d", 256), ("trig", 128), ("smooth", 256)])
def test_render3d_matches_oracle(mpr, orc, tapes, name, S):
    compare_frame(mpr, orc, tapes(name), 3, S, view3())


def test_render3d_general_view(mpr, orc, tapes):
    c, s = np.float32(np.cos(0.4)), np.float32(np.sin(0.4))
    m = np.array([[c, 0, s, 0.05], [0, 1, 0, -0.02], [-s, 0, c, 0], [0, 0.1, 0.25, 1]], dtype=np.float32)
    compare_frame(mpr, orc, tapes("hello_world"), 3, 256, m)


def test_brute_equals_hierarchy_prospero_1024(mpr, tapes):
    """BASELINE config 2 at full size: the hierarchical image must equal the brute-force one
    (the invariant benchmark/brute.cu:102-155 relies on)."""
    tape = tapes("prospero")
    ctx = mpr.Context(1024)
    ctx.render2D(tape, view2())
    img = ctx.image
    ctx.render2D_brute(tape, view2())
    assert np.array_equal(img, ctx.image)
    assert 0 < img.sum() < img.size
    ctx.render2D(tape, view2())       # idempotent
    assert np.array_equal(img, ctx.image)
    ctx.close()


def test_brute_force_prospero_1024_equals_the_oracles_brute_force(mpr, orc, tapes):
    """render2D_brute (reference src/context.cu:1460-1508) against the oracle's own brute-force frame, directly (VERDICT r3
    weak-3: so far only through hierarchy == brute on the GPU and hierarchy == oracle)."""
    tape = tapes("prospero")
    ref = orc.Frame(tape.data, 2, 1024, mpr.colmajor(view2(), 3), brute=True, keep_pool=False, threads=0)
    ctx = mpr.Context(1024)
    ctx.render2D_brute(tape, view2())
    assert np.array_equal(ctx.image, ref.image), int((ctx.image != ref.image).sum())
    assert 0 < ref.image.sum() < ref.image.size
    ctx.close()


def test_gears_4096_brute_equals_hierarchy(mpr, tapes):
    """BASELINE config 3 (deep tape, 4096^2)."""
    tape = tapes("involute_gear_2d")
    ctx = mpr.Context(4096)
    ctx.render2D(tape, view2())
    img = ctx.image
    ctx.render2D_brute(tape, view2())
    assert np.array_equal(img, ctx.image)
    ctx.close()


def test_bear_1024_properties(mpr, orc, tapes):
    """BASELINE config 4 at full size: every heightmap value is the top of a filled voxel
    (the oracle's float walk of the root tape is < 0 there and >= 0 just above), normals
    exist exactly on filled pixels, and the frame is reproducible."""
    tape = tapes("bear")
    S = 1024
    ctx = mpr.Context(S)
    ctx.render3D(tape, view3())
    h, n = ctx.image, ctx.normals
    assert np.array_equal(n != 0, h != 0)
    assert (h > 0).sum() > 300000
    ctx.render3D(tape, view3())
    assert np.array_equal(h, ctx.image) and np.array_equal(n, ctx.normals)
    ctx.close()
    # spot-check 4096 filled pixels against a float evaluation of the ROOT tape by the oracle's
    # primitives: f(x, y, h) < 0 and f(x, y, z) >= 0 for every z > h
    rng = np.random.default_rng(3)
    ys, xs = np.nonzero(h)
    pick = rng.choice(ys.size, 512, replace=False)
    data = tape.data
    T = view3()

    def feval(px, py, pz):
        fx = ((px + np.float32(0.5)) * np.float32(1.0 / S) - np.float32(0.5)) * np.float32(2)
        fy = ((py + np.float32(0.5)) * np.float32(1.0 / S) - np.float32(0.5)) * np.float32(2)
        fz = ((pz + np.float32(0.5)) * np.float32(1.0 / S) - np.float32(0.5)) * np.float32(2)
        fw = np.float32(T[3, 2]) * fz + np.float32(1)
        slots = {}
        h0 = int(data[0])
        slots[(h0 >> 8) & 255] = (fx / fw).astype(np.float32)
        slots[(h0 >> 16) & 255] = (fy / fw).astype(np.float32)
        slots[(h0 >> 24) & 255] = (fz / fw).astype(np.float32)
        for c in data[1:-1]:
            c = int(c)
            op, o, l, r = c & 255, (c >> 8) & 255, (c >> 16) & 255, (c >> 24) & 255
            imm = float(np.uint32(c >> 32).view(np.float32))
            a = slots.get(l, np.zeros_like(fx))
            b = slots.get(r, np.zeros_like(fx))
            slots[o] = orc.float_op(op, a, b, imm)
        return slots[(int(data[-1]) >> 8) & 255]

    px, py = xs[pick].astype(np.float32), ys[pick].astype(np.float32)
    hz = h[ys[pick], xs[pick]]
    assert (feval(px, py, hz.astype(np.float32)) < 0).all()
    for dz in (1, 2, 5, 17):
        zz = np.minimum(hz + dz, S - 1)
        above = feval(px, py, zz.astype(np.float32))
        assert ((above >= 0) | (zz == hz)).all()


def test_pool_overflow_falls_back(mpr, orc, tapes):
    """A pool too small for the pushes: tiles keep their parent's tape (reference
    src/context.cu:336-347) and the image is unchanged."""
    tape = tapes("hello_world")
    full = mpr.Context(256)
    full.render2D(tape, view2())
    img = full.image
    full.close()
    small = mpr.Context(256, pool_clauses=tape.length + 64 * 40, flags=mpr.CTX_COUNTERS)
    small.render2D(tape, view2())
    assert small.counters()["pool_overflowed"] == 1
    assert np.array_equal(small.image, img)
    small.close()


def test_golden_images(mpr, tapes):
    """Committed golden digests (tests/golden/frames.json, produced by the oracle)."""
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "frames.json")) as f:
        golden = json.load(f)
    for g in golden["frames"]:
        tape = tapes(g["model"])
        assert hashlib.sha256(tape.data.tobytes()).hexdigest() == g["tape_sha256"]
        ctx = mpr.Context(g["size"])
        if g["dim"] == 2:
            ctx.render2D(tape, view2())
        else:
            ctx.render3D(tape, view3())
        assert hashlib.sha256(ctx.image.tobytes()).hexdigest() == g["image_sha256"], g
        if g["dim"] == 3:
            assert hashlib.sha256(ctx.normals.tobytes()).hexdigest() == g["normals_sha256"], g
        ctx.close()


def test_column_partition_pack_unpack(mpr, orc, tapes):
    """The multi-GPU data path on one device: two contexts play rank 0 and rank 1, each renders
    its columns (LPT deal on measured work), packs them, and unpacks the other's pack; both end
    up with the single-GPU frame, bit for bit."""
    import torch
    from mpr_amd.multigpu import column_weights
    tape = tapes("bear")
    S, world = 256, 2
    T = view3()
    full = mpr.Context(S)
    full.render3D(tape, T)
    want_h, want_n = full.image, full.normals
    owner = mpr.partition_columns((S // 64) ** 2, world, column_weights(full.stages[3].tiles, S))
    full.close()
    cap = int(np.bincount(owner, minlength=world).max())
    ctxs = [mpr.Context(S) for _ in range(world)]
    packs = [torch.zeros(cap * 4096 * 2, dtype=torch.int32, device="cuda") for _ in range(world)]
    for r in range(world):
        ctxs[r].render3D_part(tape, T, owner, r)
        mask = np.kron(owner.reshape(S // 64, S // 64) == r, np.ones((64, 64), dtype=bool))
        got = ctxs[r].image
        assert not got[~mask].any() and np.array_equal(got[mask], want_h[mask])
        ctxs[r].pack_columns(owner, r, cap, True, packs[r].data_ptr())
    torch.cuda.synchronize()
    for r in range(world):
        for o in range(world):
            if o != r:
                ctxs[r].unpack_columns(owner, o, cap, True, packs[o].data_ptr())
        assert np.array_equal(ctxs[r].image, want_h)
        assert np.array_equal(ctxs[r].normals, want_n)
        ctxs[r].close()


def test_planned_gather_through_the_renderer(mpr, tapes):
    """The steady-state multi-GPU loop (resident plan, asynchronous frame, one pack and one unpack
    launch, collective ordered on the context's stream) on one device: three TileParallelRenderers
    play ranks 0..2, the "all-gather" concatenates their packs with torch copies issued on each
    context's own stream.  Every rank ends up with the single-GPU frame."""
    import torch
    from mpr_amd.multigpu import TileParallelRenderer
    tape = tapes("bear")
    S, world = 256, 3
    T = view3()
    full = mpr.Context(S)
    full.render3D(tape, T)
    want_h, want_n = full.image, full.normals
    full.close()
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
    # frames, by hand in the order render() uses, with the collective emulated after all packs exist; from the second
    # frame of a rank's share on, the last tile stage pushes no tapes (same tape, view and partition as the frame before)
    for frame in range(3):
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
    assert not any(c.last_stage_pushed() for c in ctxs)
    # a reader of counters / tiles makes a rank render its columns again the reference's way: the columns the other ranks sent
    # stay in place (the re-render used to clear the whole image)
    for r, c in enumerate(ctxs):
        cnt = c.counters()
        assert c.last_stage_pushed() and cnt["tape_index"] > 0
        assert np.array_equal(c.image, want_h) and np.array_equal(c.normals, want_n)
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("name,dim,S", [
    ("hello_world", 2, 256), ("prospero", 2, 512), ("involute_gear_2d", 2, 512),
    ("bear", 3, 256), ("architecture", 3, 256), ("involute_gear_3d", 3, 128),
])
def test_assembly_float_pass_matches_compiled_one(mpr, tapes, name, dim, S, monkeypatch):
    """The float pass runs an interpreter written in gfx950 assembly (kernels_voxel_asm.hip);
    MPR_VOXEL_ASM=0 selects the compiled C++ interpreter.  Same frame, bit for bit — including
    the models with division, sqrt and the transcendental opcodes that leave the assembly loop."""
    tape = tapes(name)
    monkeypatch.setenv("MPR_VOXEL_ASM", "0")
    a = mpr.Context(S)
    monkeypatch.setenv("MPR_VOXEL_ASM", "1")
    b = mpr.Context(S)
    for ctx in (a, b):
        if dim == 2:
            ctx.render2D(tape, view2())
        else:
            ctx.render3D(tape, view3())
    assert a.image.any()
    assert np.array_equal(a.image, b.image)
    if dim == 3:
        assert np.array_equal(a.normals, b.normals)
    if dim == 2:
        a.render2D_brute(tape, view2())
        b.render2D_brute(tape, view2())
        assert np.array_equal(a.image, b.image)
    a.close()
    b.close()


@pytest.mark.parametrize("name,dim,S", [
    ("hello_world", 2, 256), ("prospero", 2, 512), ("involute_gear_2d", 2, 512), ("trig", 2, 256), ("many_slots", 2, 128),
    ("bear", 3, 256), ("architecture", 3, 256), ("involute_gear_3d", 3, 128), ("trig", 3, 128), ("many_slots", 3, 128),
    ("architecture", 3, 512), ("architecture", 3, 1024), ("involute_gear_3d", 3, 256),
])
@pytest.mark.parametrize("groups", ["2", "0"])
def test_generated_code_float_pass_matches_assembly_interpreter(mpr, tapes, name, dim, S, groups, monkeypatch):
    """By default the float pass translates tapes into gfx950 machine code on the device and runs that
    (kernels_voxel_jit.hip): once per group of 64 sibling tiles with the children's min / max decisions
    applied by selects (group form, up to 128 decisions per tape; by default only while the groups' tapes are
    not much longer than the children's own, MPR_VOXEL_GROUPS=2: always, =0: once per smallest tile, each with its
    own tape).
    MPR_VOXEL_JIT=0 selects the assembly interpreter (slots in LDS).  Same frame, bit for bit,
    hierarchical and brute force — the latter runs the whole root tape as one piece of generated code."""
    tape = tapes(name)
    monkeypatch.setenv("MPR_WIDE_LATER", "0")      # a level-parallel last stage keeps no decision masks: group form off
    monkeypatch.setenv("MPR_VOXEL_GEN", "0")       # (tapes the HOST generates a float walk for would take that: see the next test but one)
    monkeypatch.setenv("MPR_VOXEL_GROUPS", groups)
    monkeypatch.setenv("MPR_VOXEL_JIT", "0")
    a = mpr.Context(S)
    monkeypatch.setenv("MPR_VOXEL_JIT", "2")       # 2: generated code per tile wherever the group form is not possible
    b = mpr.Context(S)
    for ctx in (a, b):
        if dim == 2:
            ctx.render2D(tape, view2())
        else:
            ctx.render3D(tape, view3())
    assert a.float_kernel().startswith("k_eval_voxels_asm")
    if name != "many_slots":                       # (more slots than the generated code has registers for)
        assert b.float_kernel().startswith("k_eval_voxels_jit<" if groups == "0" else "k_eval_voxels_jit"), b.float_kernel()
        if groups == "2" and (name, S) in (("architecture", 1024), ("bear", 256)):
            assert b.float_kernel().startswith("k_eval_voxels_jit_groups"), b.float_kernel()      # up to 107 / 27 decisions per tape
    assert a.image.any()
    assert np.array_equal(a.image, b.image), int((a.image != b.image).sum())
    if dim == 3:
        assert np.array_equal(a.normals, b.normals)
    if dim == 2:
        a.render2D_brute(tape, view2())
        b.render2D_brute(tape, view2())
        assert np.array_equal(a.image, b.image)
        b.render2D(tape, view2())          # and again through the hierarchy: regions are rewritten tile after tile
        a.render2D(tape, view2())
        assert np.array_equal(a.image, b.image)
    a.close()
    b.close()


@pytest.mark.parametrize("name,S", [("bear", 256), ("bear", 512), ("bear", 1024), ("trig", 128), ("trig", 256), ("two_spheres", 128), ("sphere", 128), ("smooth", 256), ("smooth", 512)])
def test_float_pass_on_the_root_tapes_host_generated_code(mpr, orc, tapes, name, S, monkeypatch):
    """Tapes of at most 24 slots and 64 min / max clauses, 3-D frames whose tile stages kept their tiles' decisions: the float pass
    runs the ROOT tape's float walk as code generated on the host (csrc/voxel_gen.cpp, k_eval_voxels_gen) with the decisions of a
    tile's 16^3 parent (its record) and its own (the group's masks) as bits — dead runs of clauses jumped over.  The oracle's
    heights and normals, frame after frame; the same with no guards at all (MPR_VOXEL_GEN_RUN=0), with a guard in front of every
    dead clause (=1), and as the device-translated group form gives them (MPR_VOXEL_GEN=0)."""
    tape = tapes(name)
    ref = orc.Frame(tape.data, 3, S, mpr.colmajor(view3(), 4), threads=0, keep_pool=False)
    # (round 6: by footprint segments — k_eval_voxels_gen_fp: the tiles over one 4 x 4 footprint inside a block of siblings in one
    # wavefront, nearest first, until one is hidden — unless MPR_VOXEL_FP=0: tile by tile down a z-sorted list, round 3's kernel)
    ctxs = [(mpr.Context(S), "k_eval_voxels_gen_fp<3>")]
    monkeypatch.setenv("MPR_VOXEL_FP", "0")
    ctxs.append((mpr.Context(S), "k_eval_voxels_gen<3>"))
    monkeypatch.setenv("MPR_VOXEL_GEN_RUN", "0")
    ctxs.append((mpr.Context(S), "k_eval_voxels_gen<3>"))
    monkeypatch.delenv("MPR_VOXEL_FP")
    ctxs.append((mpr.Context(S), "k_eval_voxels_gen_fp<3>"))
    monkeypatch.setenv("MPR_VOXEL_GEN_RUN", "1")
    ctxs.append((mpr.Context(S), "k_eval_voxels_gen_fp<3>"))
    monkeypatch.setenv("MPR_VOXEL_GEN_TILES", "1")
    ctxs.append((mpr.Context(S), "k_eval_voxels_gen_fp<3>"))
    monkeypatch.setenv("MPR_VOXEL_GEN_TILES", "7")
    ctxs.append((mpr.Context(S), "k_eval_voxels_gen_fp<3>"))
    monkeypatch.delenv("MPR_VOXEL_GEN_TILES")
    monkeypatch.delenv("MPR_VOXEL_GEN_RUN")
    monkeypatch.setenv("MPR_VOXEL_GEN", "0")
    monkeypatch.setenv("MPR_VOXEL_GROUPS", "2")    # (without the host's code the group form is the faster one only up to 1.4x shortening: trig's sample says 1.5 - 1.9)
    ctxs.append((mpr.Context(S), "k_eval_voxels_jit_groups<3, 24>"))
    monkeypatch.delenv("MPR_VOXEL_GROUPS")
    for ctx, kernel in ctxs:
        for _ in range(3):
            ctx.render3D(tape, view3())
            assert ctx.float_kernel() == kernel, (ctx.float_kernel(), kernel)
            assert np.array_equal(ctx.image, ref.filled[3]), (kernel, int((ctx.image != ref.filled[3]).sum()))
            assert np.array_equal(ctx.normals, ref.normals)
        ctx.close()


@pytest.mark.parametrize("name,S", [("bear", 256), ("bear", 1024), ("smooth", 128), ("smooth", 256), ("smooth", 1024), ("trig", 128), ("trig", 256)])
def test_tile_stages_with_loose_exp_and_log(mpr, orc, tapes, name, S, monkeypatch):
    """Frames nobody reads run their generated tile stages with exp / log enclosures from the hardware's base-2 instructions
    (sound, about 1e-5 wide instead of correctly rounded; MPR_TILE_GEN_LOOSE=0: the exact ones): a few tiles that the reference
    culls, fills or shortens stay ambiguous / undecided, the heights and normals are the oracle's bit for bit, and a reader
    still gets the reference's tiles and tapes."""
    tape = tapes(name)
    ref = orc.Frame(tape.data, 3, S, mpr.colmajor(view3(), 4), threads=0)
    monkeypatch.setenv("MPR_TILE_TIGHT", "0")      # (the last stage's second verdict — fewer tiles for the float pass, not more — has its own file: test_gpu_tight.py)
    loose = mpr.Context(S)
    monkeypatch.setenv("MPR_TILE_GEN_LOOSE", "0")
    exact = mpr.Context(S)
    counts = []
    # (asin / acos are not inclusion-isotone — a wider operand can miss their out-of-domain NaN: such a tape keeps the exact routines)
    for ctx, tag in ((loose, "" if name == "trig" else "+loose"), (exact, "")):
        for _ in range(2):
            ctx.render3D(tape, view3())
            assert np.array_equal(ctx.image, ref.filled[3]), (tag, int((ctx.image != ref.filled[3]).sum()))
            assert np.array_equal(ctx.normals, ref.normals), (tag, int((ctx.normals != ref.normals).sum()))
        forms = ctx.tile_stage_forms().split()
        # (the first stage of a frame that starts at the 16^3 tiles and is verified against the 64^3 ones keeps the exact routines:
        # a looser child decides less than its exact parent did)
        assert ("+loose" in forms[-1]) == (tag != "") and not any("+loose" in f for f in forms if tag == ""), forms
        counts.append(ctx.frame_tiles())
    # looser bounds leave more tiles ambiguous, not many more (which tiles a fill of the same launch still culls depends on timing:
    # half a per cent of slack)
    (lin, lamb, lvox), (ein, eamb, evox) = counts
    assert evox * 0.995 - 8 <= lvox <= evox * 1.03 + 8, (lvox, evox)
    assert all(a >= b * 0.995 - 8 for a, b in zip(lamb, eamb)), (lamb, eamb)
    print("smallest tiles: loose %d, exact %d; ambiguous per stage: %s / %s" % (lvox, evox, lamb, eamb))
    # the reader's state is the reference's either way
    assert loose.stages[3].tile_array_size == ref.tiles[3].size == exact.stages[3].tile_array_size
    for c in (loose, exact):
        c.close()


def test_frames_that_start_at_the_16_tiles_are_verified_against_the_64_tiles(mpr, orc, monkeypatch):
    """A frame nobody reads starts at the 16^3 tiles: each decides by itself, on the root tape, what its 64^3 parent would have
    decided for it — the reference's procedure only where its interval routines are inclusion-isotone.  They are not where a
    special case takes over (csrc/frame_domain.hpp), and then the reference's image depends on what the 64^3 tile did: this shape
    (tests/test_gpu_fuzz_shapes.py found it) takes asin / acos outside [-1, 1] in part of the view; the NaN end makes the reference's
    interval product [0, 0], the 64^3 tile decides on the strength of it, its children inherit the decision — and children
    left to themselves draw something else (shown here with the verification switched off).  With it, the 64^3 tiles are walked
    beside the frame, every 16^3 tile is held against its parent, the frame fails, is rendered again from the 64^3 tiles down
    (and so are the tape's next frames): the oracle's image.  bear's frames pass it in every frame — at 512^3 and 1024^3; at
    256^3, where a 64^3 tile is a quarter of the view, they do not, and start at the 64^3 tiles."""
    import test_gpu_fuzz_shapes
    tape = test_gpu_fuzz_shapes.fuzz_tape(mpr, 14, 12)
    assert not tape.frame_is_tame(view3())
    ref = orc.Frame(tape.data, 3, 128, mpr.colmajor(view3(), 4), threads=0)
    ctx = mpr.Context(128)
    for k in range(3):
        ctx.render3D(tape, view3())
        assert np.array_equal(ctx.image, ref.filled[3]) and np.array_equal(ctx.normals, ref.normals)
        assert ctx.skip0_vetoes() == 1                                   # the first frame, once
        assert ctx.tile_stage_forms().startswith("0:"), ctx.tile_stage_forms()
    ctx.close()
    monkeypatch.setenv("MPR_SKIP0_CHECK", "0")
    raw = mpr.Context(128)
    raw.render3D(tape, view3())
    assert raw.tile_stage_forms().startswith("1:") and raw.skip0_vetoes() == 0
    assert int((raw.image != ref.filled[3]).sum()) > 0                   # what the verification is there for
    raw.close()
    monkeypatch.delenv("MPR_SKIP0_CHECK")
    # a blend whose exp underflows far from its surface takes log's zero bound in every frame (bear does): not tame, verified, passes
    tape = mpr.Tape(mpr.model("bear"))
    for S, passes in ((1024, True), (512, True), (256, False)):
        assert not tape.frame_is_tame(view3())
        ctx = mpr.Context(S)
        for k in range(2):
            ctx.render3D(tape, view3())
        assert (ctx.skip0_vetoes() == 0) == passes, (S, ctx.skip0_vetoes())
        assert ctx.tile_stage_forms().startswith("1:" if passes else "0:"), (S, ctx.tile_stage_forms())
        ctx.close()
    # a shape that stays inside every domain over the whole view needs no verification
    assert mpr.Tape(mpr.model("hello_world")).frame_is_tame(view3())


def test_first_stage_that_leaves_records_and_no_tapes(mpr, orc, tapes, monkeypatch):
    """Frames that start at the 16^3 tiles, of a tape whose last stage pushes nothing and whose float and normals passes run the root
    tape's code with the tiles' records: nobody walks the tapes the first stage would push, so it walks forward only and leaves
    records (TileStageArgs::gen_forward_only; "+fwdonly" in mpr_ctx_tile_stage_forms).  From the second frame of a tape on (the
    first one's sample says that the group form pays), every 32nd frame an ordinary one again; the oracle's heights and normals
    in every frame, and the same with the switch off.  A shape whose last stage has too few tiles for the 64-tiles-per-wavefront
    kernel (it runs level-parallel, on the tiles' own tapes) starts over once and stays off the path."""
    tape = tapes("bear")
    ref = orc.Frame(tape.data, 3, 512, mpr.colmajor(view3(), 4), threads=0)
    ctx = mpr.Context(512)
    kinds = []
    for k in range(36):
        ctx.render3D(tape, view3())
        assert np.array_equal(ctx.image, ref.filled[3]) and np.array_equal(ctx.normals, ref.normals), k
        kinds.append("+fwdonly" in ctx.tile_stage_forms().split()[0])
        assert ctx.tile_stage_forms().startswith("1:gen") and ctx.float_kernel() == "k_eval_voxels_gen_fp<3>" and ctx.normals_kernel() == "k_eval_normals_gen"
    # (the very first frame of a fresh context may start over — its pool grows — and then already knows the hint)
    first = kinds.index(True)
    assert first <= 1 and kinds[first:first + 31] == [True] * 31 and kinds[first + 31] is False and kinds[first + 32:] == [True] * (35 - first - 31), kinds
    assert ctx.skip0_vetoes() == 0
    # a reader after such a frame still gets the reference's tiles and tapes
    assert ctx.stages[3].tile_array_size == ref.tiles[3].size
    ctx.close()
    monkeypatch.setenv("MPR_LEAN_FIRST", "0")
    ctx = mpr.Context(512)
    for k in range(3):
        ctx.render3D(tape, view3())
        assert np.array_equal(ctx.image, ref.filled[3]) and np.array_equal(ctx.normals, ref.normals)
        assert "+fwdonly" not in ctx.tile_stage_forms()
    ctx.close()
    monkeypatch.delenv("MPR_LEAN_FIRST")
    for name, S in (("two_spheres", 128), ("sphere", 256), ("smooth", 128)):
        tape = tapes(name)
        ref = orc.Frame(tape.data, 3, S, mpr.colmajor(view3(), 4), threads=0)
        ctx = mpr.Context(S)
        for k in range(5):
            ctx.render3D(tape, view3())
            assert np.array_equal(ctx.image, ref.filled[3]) and np.array_equal(ctx.normals, ref.normals), (name, k, ctx.tile_stage_forms())
        ctx.close()


@pytest.mark.parametrize("name,dim,S", [("bear", 3, 512), ("architecture", 3, 1024), ("hello_world", 2, 256)])
def test_code_ring_against_an_invalidate_per_group(mpr, tapes, name, dim, S, monkeypatch):
    """The group form writes a group's code into the next slot of a ring and invalidates the instruction cache only when
    it re-enters slot 0; that rests on the slots lying 4 KB apart (beyond the sequential instruction prefetch).
    MPR_VOXEL_JIT=3 invalidates after every translation — no such assumption — and a gap below the validated one
    (MPR_JIT_GAP) falls back to it by itself: all three give the same frame, repeatedly."""
    tape = tapes(name)
    monkeypatch.setenv("MPR_VOXEL_GROUPS", "2")
    monkeypatch.setenv("MPR_VOXEL_GEN", "0")
    monkeypatch.setenv("MPR_WIDE_LATER", "0")      # a level-parallel last stage keeps no decision masks: group form off
    ring = mpr.Context(S)
    monkeypatch.setenv("MPR_VOXEL_JIT", "3")
    every = mpr.Context(S)
    monkeypatch.setenv("MPR_VOXEL_JIT", "1")
    monkeypatch.setenv("MPR_JIT_GAP", "16")           # 64 bytes between slots: far inside the prefetch's reach
    tight = mpr.Context(S)
    for rep in range(3):
        for ctx in (ring, every, tight):
            if dim == 2:
                ctx.render2D(tape, view2())
            else:
                ctx.render3D(tape, view3())
            assert ctx.float_kernel().startswith("k_eval_voxels_jit_groups"), ctx.float_kernel()
        assert ring.image.any()
        assert np.array_equal(ring.image, every.image) and np.array_equal(ring.image, tight.image)
        if dim == 3:
            assert np.array_equal(ring.normals, every.normals) and np.array_equal(ring.normals, tight.normals)
    for ctx in (ring, every, tight):
        ctx.close()


@pytest.mark.parametrize("name,dim,S", [("prospero", 2, 256), ("involute_gear_2d", 2, 512), ("hello_world", 2, 256),
                                        ("bear", 3, 256), ("architecture", 3, 256), ("involute_gear_3d", 3, 128)])
def test_serial_first_stage_matches_oracle(mpr, orc, tapes, name, dim, S, monkeypatch):
    """By default the first tile stage runs one workgroup per tile, level by level over the tape's
    DAG (kernels_wide.hip); MPR_WIDE_STAGE0=0 selects the one-lane-per-tile walk for that stage
    too.  Both must give the oracle's tiles, images and shortened tapes (the default path is what
    every other test in this file exercises)."""
    monkeypatch.setenv("MPR_WIDE_STAGE0", "0")
    compare_frame(mpr, orc, tapes(name), dim, S, view2() if dim == 2 else view3())


@pytest.mark.parametrize("name,dim,S", [("prospero", 2, 256), ("prospero", 2, 512), ("involute_gear_2d", 2, 512), ("hello_world", 2, 256),
                                        ("hello_world", 3, 256), ("bear", 3, 256), ("architecture", 3, 256), ("architecture", 3, 512),
                                        ("involute_gear_3d", 3, 256), ("trig", 3, 128), ("two_spheres", 2, 256)])
@pytest.mark.parametrize("later", ["0", "100000000"])
def test_level_parallel_later_stages_match_oracle(mpr, orc, tapes, name, dim, S, later, monkeypatch):
    """Stages after the first run level-parallel too (kernels_wide.hip: on the root tape's schedule, with a
    two-bit table per tile for the tape it inherited) while they have at most MPR_WIDE_LATER tiles
    (default 16384) and the stage before ran that way.  Here: never, and always (with MPR_WIDE_FORCE, so that
    tapes with narrow DAGs like bear's take the path as well).  Both must give the oracle's tiles, images
    and shortened tapes at every stage."""
    monkeypatch.setenv("MPR_WIDE_LATER", later)
    if later != "0":
        monkeypatch.setenv("MPR_WIDE_FORCE", "1")
    compare_frame(mpr, orc, tapes(name), dim, S, view2() if dim == 2 else view3())


@pytest.mark.parametrize("name,S", [("bear", 256), ("architecture", 512), ("hello_world", 256), ("involute_gear_3d", 256), ("trig", 128),
                                    ("two_spheres", 128)])
def test_frames_without_last_stage_tapes(mpr, orc, tapes, name, S, monkeypatch):
    """By default a frame's last tile stage pushes no tapes when the float and the normals pass can walk the groups' tapes with
    every tile's decisions applied — decided by the stage's own measurement, every frame, the first one of a tape included — and
    a 3-D frame of this size starts at the 16^3 tiles (bear: narrow DAG; the others keep the level-parallel first stage).
    Heights and normals are the oracle's; reading tiles or tapes afterwards gives the reference's state (the context renders
    the frame again the reference's way); a tape whose last stage shortens too much gets its tapes from a second run of that
    stage once and pushes straight away from then on."""
    monkeypatch.setenv("MPR_WIDE_LATER", "0")
    tape = tapes(name)
    ref = orc.Frame(tape.data, 3, S, mpr.colmajor(view3(), 4), threads=0)
    ctx = mpr.Context(S)
    kinds = []
    for _ in range(4):
        ctx.render3D(tape, view3())
        kinds.append((ctx.last_stage_pushed(), ctx.float_kernel().split("<")[0]))
        assert np.array_equal(ctx.image, ref.filled[3])
        bad = np.flatnonzero(ctx.normals.ravel() != ref.normals.ravel())
        assert bad.size == 0, (bad.size, kinds, [(hex(ctx.normals.ravel()[i]), hex(ref.normals.ravel()[i])) for i in bad[:5]])
    if name == "bear":
        assert kinds == [(False, "k_eval_voxels_gen_fp")] * 4, kinds                  # no tapes from the very first frame
    if name == "involute_gear_3d":
        assert all(k == (True, "k_eval_voxels_asm") for k in kinds), kinds         # per-tile tapes: measured by the first frame, known after
    # the reference's state on request: tiles and tapes as a frame rendered the reference's way leaves them
    monkeypatch.setenv("MPR_LAST_STAGE_PUSH", "1")
    always = mpr.Context(S)
    monkeypatch.delenv("MPR_LAST_STAGE_PUSH")
    for _ in range(2):
        always.render3D(tape, view3())
        assert always.last_stage_pushed()
        assert np.array_equal(always.image, ref.filled[3]) and np.array_equal(always.normals, ref.normals)
    ctx.render3D(tape, view3())
    pool, apool = ctx.tape_data, always.tape_data
    assert ctx.last_stage_pushed()                     # reading rendered the frame again, the reference's way
    for s in (0, 1, 2, 3):
        g, a = ctx.stages[s].tiles, always.stages[s].tiles
        assert g.size == a.size == ref.tiles[s].size
        g, a = g[g["position"] != -1], a[a["position"] != -1]
        g, a = g[np.argsort(g["position"])], a[np.argsort(a["position"])]
        assert np.array_equal(g["position"], a["position"])
        glen, gh = orc.tiles_digest(pool, g)
        alen, ah = orc.tiles_digest(apool, a)
        assert np.array_equal(glen, alen) and np.array_equal(gh, ah)
    assert np.array_equal(ctx.image, ref.filled[3]) and np.array_equal(ctx.normals, ref.normals)
    # another view of the same tape: the oracle's image again
    T = view3().copy()
    T[0, 3] = 0.125
    ref2 = orc.Frame(tape.data, 3, S, mpr.colmajor(T, 4), threads=0)
    for k in range(3):
        ctx.render3D(tape, T)
        assert np.array_equal(ctx.image, ref2.filled[3]) and np.array_equal(ctx.normals, ref2.normals)
    if name == "bear":
        assert not ctx.last_stage_pushed()
    for c in (ctx, always):
        c.close()


@pytest.mark.parametrize("name,S", [("bear", 256), ("bear", 1024), ("hello_world", 256), ("trig", 128), ("many_slots", 128)])
def test_frames_that_start_at_the_16px_tiles(mpr, orc, tapes, name, S, monkeypatch):
    """3-D frames nobody inspects start at the 16^3 tiles when the 64^3 stage would be a handful of wavefronts walking the whole
    tape (context.hip: skip0).  Same heights and normals as with the first stage (MPR_SKIP_STAGE0=0) and as the oracle."""
    tape = tapes(name)
    ref = orc.Frame(tape.data, 3, S, mpr.colmajor(view3(), 4), threads=0, keep_pool=False)
    fast = mpr.Context(S)
    monkeypatch.setenv("MPR_SKIP_STAGE0", "0")
    slow = mpr.Context(S)
    for ctx in (fast, slow):
        for _ in range(2):
            ctx.render3D(tape, view3())
            assert np.array_equal(ctx.image, ref.filled[3]), int((ctx.image != ref.filled[3]).sum())
            assert np.array_equal(ctx.normals, ref.normals), int((ctx.normals != ref.normals).sum())
    # a reader gets the reference's first stage back
    t0 = fast.stages[0].tiles
    assert np.array_equal(np.sort(t0["position"][t0["next"] != -1]), np.sort(ref.tiles[0]["position"][ref.tiles[0]["next"] != -1]))
    fast.close()
    slow.close()


@pytest.mark.parametrize("name,dim,S", [("bear", 3, 256), ("bear", 2, 512), ("trig", 3, 128), ("trig", 2, 256), ("two_spheres", 3, 128), ("ring", 2, 256)])
@pytest.mark.parametrize("gen", ["0", "2"])
def test_first_stage_without_generated_code(mpr, orc, tapes, name, dim, S, gen, monkeypatch):
    """A frame's first tile stage — all of its tiles walk the root tape — runs that tape's interval walks as machine code
    generated on the host (csrc/tile_gen.cpp) when the tape has at most 24 slots and 64 min / max clauses: every other test
    of this file.  MPR_TILE_GEN=0: the interpreter walks it instead; 2: generated code forward, the interpreter backward.
    The oracle's frame either way, shortened tapes included."""
    monkeypatch.setenv("MPR_TILE_GEN", gen)
    compare_frame(mpr, orc, tapes(name), dim, S, view2() if dim == 2 else view3())


@pytest.mark.parametrize("name,S", [("bear", 256), ("trig", 128), ("two_spheres", 128)])
@pytest.mark.parametrize("chain", ["0", "1"])
def test_frames_that_are_read_on_chains_of_generated_stages(mpr, orc, tapes, name, S, chain, monkeypatch):
    """A frame whose tiles and tapes are read runs all three tile stages: with MPR_TILE_GEN_CHAIN=1 (default) every one of them on
    the root tape's generated code — the stages below the first shorten their parents' tapes by the backward code that follows
    the parent's tape clause by clause (records with presence bits) — and the normals pass on the 4^3 tiles' own records; with 0
    only the first stage.  compare_frame: the oracle's tiles and tapes at every stage; then frames that are not read."""
    monkeypatch.setenv("MPR_TILE_GEN_CHAIN", chain)
    tape = tapes(name)
    cnt, ref = compare_frame(mpr, orc, tape, 3, S, view3())
    monkeypatch.setenv("MPR_LAST_STAGE_PUSH", "1")
    ctx = mpr.Context(S)
    for _ in range(2):
        ctx.render3D(tape, view3())
        assert np.array_equal(ctx.image, ref.filled[3]) and np.array_equal(ctx.normals, ref.normals)
    assert ctx.normals_kernel() == ("k_eval_normals_gen" if chain == "1" else "k_eval_normals_asm")
    ctx.close()


_READER_REFS = {}


@pytest.mark.parametrize("how", ["always", "after_fast_frames"])
@pytest.mark.parametrize("chain", ["0", "1"])
@pytest.mark.parametrize("name,S", [("bear", 256), ("bear", 512), ("bear", 1024), ("trig", 128), ("two_spheres", 128), ("sphere", 128)])
def test_readers_tapes_on_chains_of_generated_stages_match_the_oracle(mpr, orc, tapes, name, S, chain, how, monkeypatch):
    """The tiles and tapes a reader of mpr.hpp's `stages[k].tiles` / `tape_data` gets by default for tapes the host generates code
    for (compare_frame's instrumented contexts never run this path).  `always`: a context whose every frame is
    the reference's kind (MPR_LAST_STAGE_PUSH=1); `after_fast_frames`: a default context, read after frames that started at the
    16^3 tiles and pushed no last-stage tapes.  MPR_TILE_GEN_CHAIN=1 (default): all three tile stages on the root tape's generated
    code, tapes shortened by TileGen::bwd_full with the parents' records; 0: the first stage only.  Survivors and the clause
    sequence of every surviving tile's tape == the oracle's, at every stage (reference src/context.cu:323-458;
    benchmark/tape_shortening.cpp:56-117 reads exactly these)."""
    from helpers import compare_reader_frame
    monkeypatch.setenv("MPR_TILE_GEN_CHAIN", chain)
    if how == "always":
        monkeypatch.setenv("MPR_LAST_STAGE_PUSH", "1")
    tape = tapes(name)
    key = (name, S)
    if key not in _READER_REFS:
        _READER_REFS.clear()             # one oracle frame at a time (bear 1024^3 keeps a 1 GB pool)
        _READER_REFS[key] = orc.Frame(tape.data, 3, S, mpr.colmajor(view3(), 4), threads=0)
    ctx, ref = compare_reader_frame(mpr, orc, tape, S, view3(), ref=_READER_REFS[key], frames=1 if how == "always" else 2,
                                    read_first={"0": "filled0", "1": "tiles"}[chain] if how != "always" else "tape_data")
    # the path this test means to exercise (a flag must not silently move it onto another one): the tile stages that left the
    # tiles and tapes just compared ...
    forms = ctx.tile_stage_forms().split()
    assert [f.split(":")[0] for f in forms] == ["0", "1", "2"], forms
    if chain == "1":      # every stage on the root tape's generated code, tapes pushed by the walk that follows the parent's tape
        assert all(f.split(":")[1].startswith("gen") and "+bwd_full" in f for f in forms), forms
        assert "/parent" in forms[1] and "/parent" in forms[2] and "/parent" not in forms[0], forms
    else:                 # the first stage only (its lean backward walk), the interpreter below it
        assert forms[0].startswith("0:gen+bwd") and "bwd_full" not in forms[0] and forms[1:] == ["1:interp", "2:interp"], forms
    # ... and the normals pass of the frame whose normals were compared (a reader's re-render runs no normals pass)
    if how == "always":
        assert ctx.normals_kernel() == ("k_eval_normals_gen" if chain == "1" else "k_eval_normals_asm"), ctx.normals_kernel()
    elif chain == "1" or ctx.skip0_vetoes() == 0:
        assert ctx.normals_kernel() == "k_eval_normals_gen" and ctx.float_kernel() == "k_eval_voxels_gen_fp<3>", (ctx.normals_kernel(), ctx.float_kernel())
    else:
        # (bear at 256^3: the frame's start at the 16^3 tiles fails its verification against the 64^3 ones, the frames start there —
        # and without the chain of records the stages below the first one, and the passes behind them, are the interpreters')
        assert (name, S) == ("bear", 256) and ctx.normals_kernel() == "k_eval_normals_asm", (name, S, ctx.normals_kernel())
    ctx.close()


@pytest.mark.parametrize("name,S", [("bear", 256), ("bear", 512), ("trig", 128), ("trig", 256), ("two_spheres", 128), ("sphere", 128)])
def test_normals_on_the_root_tapes_generated_code(mpr, orc, tapes, name, S, monkeypatch):
    """Frames that start at the 16^3 tiles on generated code and end in group form give every pixel's normal by the ROOT tape's
    generated Deriv code with the decisions of the pixel's 16^3 and 4^3 tiles applied (k_eval_normals_gen); MPR_NORMALS_GEN=0
    interprets the tiles' tapes instead.  The oracle's normals either way, frame after frame."""
    tape = tapes(name)
    ref = orc.Frame(tape.data, 3, S, mpr.colmajor(view3(), 4), threads=0, keep_pool=False)
    fast = mpr.Context(S)
    monkeypatch.setenv("MPR_TILE_GEN_LAST", "0")     # ... the last tile stage interprets its parents' tapes (default: the root tape's
    middle = mpr.Context(S)                           # generated code with the parents' decisions imposed)
    monkeypatch.setenv("MPR_NORMALS_GEN", "0")
    slow = mpr.Context(S)
    for ctx, kernel in ((fast, "k_eval_normals_gen"), (middle, "k_eval_normals_gen"), (slow, "k_eval_normals_asm")):
        seen = set()
        for _ in range(3):
            ctx.render3D(tape, view3())
            seen.add(ctx.normals_kernel())
            assert np.array_equal(ctx.image, ref.filled[3]), int((ctx.image != ref.filled[3]).sum())
            assert np.array_equal(ctx.normals, ref.normals), int((ctx.normals != ref.normals).sum())
        assert kernel in seen, seen
    fast.close()
    middle.close()
    slow.close()


@pytest.mark.parametrize("name,dim,S", [("prospero", 2, 256), ("involute_gear_2d", 2, 512), ("trig", 2, 256),
                                        ("bear", 3, 256), ("architecture", 3, 256)])
def test_compiled_forward_walk_matches_oracle(mpr, orc, tapes, name, dim, S, monkeypatch):
    """The tile stages' forward walk runs an assembly interpreter (tile_interp_asm.hpp) when the
    tape uses at most 128 slots; MPR_TILES_ASM=0 selects the compiled loop, which is also what tapes
    with more slots get.  Both must give the oracle's frame (the default path is what every other
    test in this file exercises)."""
    monkeypatch.setenv("MPR_TILES_ASM", "0")
    compare_frame(mpr, orc, tapes(name), dim, S, view2() if dim == 2 else view3())


@pytest.mark.parametrize("name,dim,S", [("prospero", 2, 256), ("prospero", 2, 1024), ("involute_gear_2d", 2, 512), ("architecture", 3, 256),
                                        ("involute_gear_3d", 3, 256), ("involute_gear_3d", 3, 512), ("bear", 3, 256), ("bear", 3, 512)])
@pytest.mark.parametrize("vgpr", ["0", "1"])
def test_slot_file_in_registers_matches_oracle(mpr, orc, tapes, name, dim, S, vgpr, monkeypatch):
    """Tapes with 40 to 93 slots walk the tile stages with the slot file in vector registers (tile_interp_asm.hpp:
    8 wavefronts per CU instead of the 3 their LDS planes leave room for), and so do tapes with 20 to 24 slots (16
    instead of 13); MPR_TILES_VGPR=0 keeps it in LDS.  Both give
    the oracle's frame at every stage (level-parallel later stages off, so that the walk in question runs them)."""
    monkeypatch.setenv("MPR_TILES_VGPR", vgpr)
    monkeypatch.setenv("MPR_WIDE_LATER", "0")
    compare_frame(mpr, orc, tapes(name), dim, S, view2() if dim == 2 else view3())


@pytest.mark.parametrize("name,S", [("bear", 256), ("architecture", 256), ("involute_gear_3d", 256), ("trig", 128), ("sphere", 128),
                                    ("two_spheres", 128)])
def test_assembly_normals_pass_matches_compiled_one(mpr, tapes, name, S, monkeypatch):
    """The normals pass runs an interpreter written in gfx950 assembly (kernels_normals_asm.hip);
    MPR_NORMALS_ASM=0 selects the compiled one.  Same normals, bit for bit."""
    tape = tapes(name)
    monkeypatch.setenv("MPR_NORMALS_ASM", "0")
    a = mpr.Context(S)
    monkeypatch.setenv("MPR_NORMALS_ASM", "1")
    b = mpr.Context(S)
    for ctx in (a, b):
        ctx.render3D(tape, view3())
    assert a.normals.any()
    assert np.array_equal(a.image, b.image)
    bad = np.flatnonzero(a.normals.ravel() != b.normals.ravel())
    assert bad.size == 0, (bad.size, [(hex(a.normals.ravel()[i]), hex(b.normals.ravel()[i])) for i in bad[:5]])
    a.close()
    b.close()


@pytest.mark.parametrize("dim,S", [(2, 128), (3, 128)])
def test_tape_with_more_than_128_slots(mpr, orc, tapes, dim, S):
    """More live values than the reference's kernels hold (128): the interval stages fall back to
    the compiled forward / backward walks (the assembly ones address slots through a pre-doubled
    byte), the active set spills to LDS; float and normals passes address up to 255 slots."""
    tape = tapes("many_slots")
    assert tape.num_slots > 128
    compare_frame(mpr, orc, tape, dim, S, view2() if dim == 2 else view3())


@pytest.mark.parametrize("name,S", [("two_spheres", 256), ("hello_world", 256), ("trig", 128), ("involute_gear_2d", 256)])
def test_compiled_expression_baseline_equals_brute_force(mpr, tapes, name, S):
    """The tape as straight-line HIP source, compiled at run time (compiled_baseline.cpp: the
    reference's dump_tape + brute.cu comparison point), gives the image of render2D_brute."""
    tape = tapes(name)
    ctx = mpr.Context(S)
    ctx.render2D_brute(tape, view2())
    want = ctx.image.copy()
    k = mpr.CompiledTape(tape)
    assert "mpr_compiled" in k.source
    k.render2D(ctx, view2())
    got = ctx.image
    assert want.any() and np.array_equal(got, want), int((got != want).sum())
    k.close()
    ctx.close()


def test_tape_pool_sized_by_the_frames(mpr, orc, tapes):
    """The reference allocates its tape pool for the worst case (3.28 GB with BIG_SERVER, inc/parameters.hpp:18-22).  Here a
    pool the caller did not size starts at a few M clauses and doubles whenever a frame's pushes do not fit, the frame being
    rendered again — so tiles and tapes read back are what the big pool gives.  Three contexts at 2048^3 (one GPU playing three
    ranks) stay far below one reference pool."""
    tape = tapes("architecture")
    S = 2048
    ctxs = [mpr.Context(S) for _ in range(3)]
    before = sum(c.resident_bytes() for c in ctxs)
    for c in ctxs:
        for _ in range(2):
            c.render3D(tape, view3())
    after = sum(c.resident_bytes() for c in ctxs)
    print("three contexts at 2048^3: %.2f GB at creation, %.2f GB after frames of architecture" % (before / 2**30, after / 2**30))
    # (round 2: 3 x 3.28 GB for the pools alone; round 5: a pool more than 3/4 full after a frame doubles before the next one, so that no
    # later frame — whose pushes run a few per cent higher or lower with the timing of the fills — has to grow it in the middle: 3.4 GB;
    # later in round 5 architecture's last stage pushes per-tile tapes again (the faster form since the stages' atomic went) and its
    # first stage decides a little less (looser enclosures): one doubling more in each context, 4.2 GB — three reference pools: 9.8)
    assert after < 4.5 * 2**30, after            # (round 6: a pool grows by half, not by the whole)
    assert np.array_equal(ctxs[0].image, ctxs[2].image)
    # reading tiles / tapes makes a frame the reference's way: the pool grows as far as that needs, and the tapes are complete
    cnt = ctxs[0].counters()
    assert cnt["pool_overflowed"] == 0 and cnt["tape_index"] > tape.length
    for c in ctxs:
        c.close()
    # bear 1024^3 the reference's way needs over a GB of tapes: grown in steps, no overflow, the oracle's survivors
    tape = tapes("bear")
    ctx = mpr.Context(1024)
    ctx.render3D(tape, view3())
    small = ctx.resident_bytes()
    cnt = ctx.counters()
    assert cnt["pool_overflowed"] == 0
    assert ctx.resident_bytes() > small and cnt["tape_index"] > 100e6
    ctx.close()


def test_contexts_recycle_their_streams(mpr, tapes):
    """A destroyed context's streams go to the next context of the device instead of hipStreamDestroy (context.hip: acquire_stream —
    the HIP runtime's stream destruction frees its virtual device under signal handlers still to run, profiles/r06_segv_hunt.txt);
    frames on a recycled stream are the frames of a fresh one."""
    tape = tapes("sphere")
    a = mpr.Context(128)
    a.render3D(tape, view3())
    first, image = a.stream, a.image.copy()
    a.close()
    seen = set()
    for _ in range(4):
        b = mpr.Context(128)
        b.render3D(tape, view3())
        assert np.array_equal(b.image, image)
        seen.add(b.stream)
        b.close()
    assert len(seen) <= 2 and (first in seen or len(seen) == 1)
    # two contexts alive at the same time do not share a stream
    c, d = mpr.Context(64), mpr.Context(64)
    assert c.stream != d.stream
    c.close()
    d.close()
