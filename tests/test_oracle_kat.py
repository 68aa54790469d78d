"""Pins the CPU oracle to the known answers derivable from the reference's own source text
(SURVEY.md §8(c)); runs without a GPU.

  1. benchmark/brute.cu:39-61  — the compiled two-sphere kernel: its image must equal
     render2D and render2D_brute of the same model (that is what brute.cu renders three ways).
  2. the hierarchy invariant behind (1): hierarchical image == brute-force image, any model.
  3. benchmark/circle.cpp:42-103 — tile-occupancy semantics of the circle model at S = 128.
  4. benchmark/print_tape_table.cpp:29-51 — the clause table of the ring model.
  5. src/tape.cpp — tape-builder invariants on the six benchmark models.
"""
import numpy as np
import pytest

from conftest import view2, view3


def brute_cu_two_spheres(S):
    """benchmark/brute.cu:39-61 evaluated in float32 numpy, operation by operation."""
    f = np.float32
    px, py = np.meshgrid(np.arange(S, dtype=np.float32), np.arange(S, dtype=np.float32))
    x = f(2.0) * ((px + f(0.5)) / f(S) - f(0.5))
    y = f(2.0) * ((py + f(0.5)) / f(S) - f(0.5))
    z = np.zeros_like(x)
    a = x + f(0.5)
    a2 = a * a
    y2 = y * y
    z2 = z * z
    yz = y2 + z2
    s1 = np.sqrt(a2 + yz) - f(0.25)
    b = x - f(0.5)
    s2 = np.sqrt(b * b + yz) - f(0.25)
    return (np.minimum(s1, s2) < 0).astype(np.int32)


@pytest.mark.parametrize("S", [256, 512])
def test_two_spheres_matches_compiled_kernel(orc, tapes, S):
    tape = tapes("two_spheres").data
    hier = orc.Frame(tape, 2, S, view2().T.reshape(-1))
    brute = orc.Frame(tape, 2, S, view2().T.reshape(-1), brute=True)
    want = brute_cu_two_spheres(S)
    assert np.array_equal(brute.image, want)
    assert np.array_equal(hier.image, want)


@pytest.mark.parametrize("name,S", [("circle", 128), ("ring", 256), ("hello_world", 256), ("prospero", 256),
                                    ("involute_gear_2d", 256), ("architecture", 128), ("bear", 128), ("trig", 128)])
def test_hierarchy_equals_brute_force_2d(orc, tapes, name, S):
    tape = tapes(name).data
    hier = orc.Frame(tape, 2, S, view2().T.reshape(-1))
    brute = orc.Frame(tape, 2, S, view2().T.reshape(-1), brute=True)
    assert np.array_equal(hier.image, brute.image)


@pytest.mark.parametrize("name,S", [("two_spheres", 128), ("hello_world", 128), ("bear", 128), ("architecture", 128), ("involute_gear_3d", 128),
                                    ("trig", 128), ("smooth", 128), ("shape_12_0", 128), ("shape_12_3", 128), ("shape_4_9", 128), ("shape_12_21", 128),
                                    ("shape_64_5", 128)])
def test_hierarchy_equals_brute_force_3d(orc, tapes, mpr, name, S):
    """... and in 3-D, where the reference has no brute-force renderer: the oracle's (`brute=True`: every 4^3 tile of the volume
    straight to the float pass with the root tape, the normals from the root tape) draws the heights AND the normals the
    64 -> 16 -> 4 hierarchy with its shortened tapes draws — for every benchmark model and for random shapes whose partial
    functions stay inside their domains."""
    tape = tapes(name).data
    hier = orc.Frame(tape, 3, S, mpr.colmajor(view3(), 4), threads=0)
    brute = orc.Frame(tape, 3, S, mpr.colmajor(view3(), 4), threads=0, brute=True)
    assert hier.image.any()
    assert np.array_equal(hier.image, brute.image)
    assert np.array_equal(hier.normals, brute.normals)


def test_where_the_references_hierarchy_is_not_its_brute_force(orc, mpr):
    """The invariant is a property of sound intervals, and the reference's are not sound where asin / acos leave [-1, 1] inside a
    tile: inc/gpu_interval.hpp:306-324 returns a NaN end, the interval product with it is [0, 0] (:86-146: every sign test is
    false), and a tile is "filled" or a min / max "decided" on the strength of it — the hierarchy then draws what no evaluation
    voxel by voxel does (117 pixels of this shape: the very pixels where tests/golden/make_independent.py's evaluator
    disagrees).  Parity means the REFERENCE's image, so this is what the product must reproduce — and what makes its shortcut
    past the 64^3 tiles a thing to verify (tests/test_gpu_render.py:
    test_frames_that_start_at_the_16_tiles_are_verified_against_the_64_tiles)."""
    import test_gpu_fuzz_shapes
    for seed, pixels in ((14, 117), (9, 1870), (15, 14)):
        tape = test_gpu_fuzz_shapes.fuzz_tape(mpr, seed, 12).data
        hier = orc.Frame(tape, 3, 128, mpr.colmajor(view3(), 4), threads=0)
        brute = orc.Frame(tape, 3, 128, mpr.colmajor(view3(), 4), threads=0, brute=True)
        assert int((hier.image != brute.image).sum()) == pixels


def test_heightmap_equals_brute_force_column_scan(orc, tapes, mpr):
    """3-D: the heightmap is the top-most voxel with f < 0 in every column (z = 0 excluded)."""
    S = 64
    tape = tapes("two_spheres").data
    T = view3()
    f = orc.Frame(tape, 3, S, mpr.colmajor(T, 4), threads=1)
    # brute force with the oracle's float primitive, voxel by voxel
    idx = np.arange(S, dtype=np.float32)
    c = ((idx + np.float32(0.5)) * np.float32(1.0 / S) - np.float32(0.5)) * np.float32(2.0)
    Z, Y, X = np.meshgrid(c, c, c, indexing="ij")
    W = np.float32(0.3) * Z + np.float32(1.0)
    slots = {}
    h0 = int(tape[0])
    slots[(h0 >> 8) & 255] = (X / W).astype(np.float32).ravel()
    slots[(h0 >> 16) & 255] = (Y / W).astype(np.float32).ravel()
    slots[(h0 >> 24) & 255] = (Z / W).astype(np.float32).ravel()
    for cl in tape[1:-1]:
        cl = int(cl)
        op, o, l, r = cl & 255, (cl >> 8) & 255, (cl >> 16) & 255, (cl >> 24) & 255
        imm = float(np.uint32(cl >> 32).view(np.float32))
        zero = np.zeros(S ** 3, dtype=np.float32)
        slots[o] = orc.float_op(op, slots.get(l, zero), slots.get(r, zero), imm)
    val = slots[(int(tape[-1]) >> 8) & 255].reshape(S, S, S)      # [z, y, x]
    inside = val < 0
    zi = np.arange(S)[:, None, None]
    want = np.where(inside, zi, 0).max(axis=0)
    assert np.array_equal(f.image, want)
    assert np.array_equal(f.normals != 0, want != 0)


def test_circle_tile_occupancy(orc, tapes):
    """benchmark/circle.cpp: S = 128.  Every 64^2 / 8^2 tile whose position was set to -1 is
    entirely inside or entirely outside the circle of radius 1.8 about (-1, -1); every surviving
    tile really straddles it."""
    S = 128
    f = orc.Frame(tapes("circle").data, 2, S, view2().T.reshape(-1))
    img = f.image

    def d2_range(x0, x1, y0, y1):
        # squared distance range from (-1, -1) over the box; the box lies in x, y >= -1
        return (x0 + 1) ** 2 + (y0 + 1) ** 2, (x1 + 1) ** 2 + (y1 + 1) ** 2

    for stage, tile_px in ((0, 64), (2, 8)):
        tps = S // tile_px
        tiles = f.tiles[stage]
        # stage-2 list positions were overwritten by -1 for decided tiles; recover them from order
        if stage == 0:
            positions = np.arange(tiles.size)
        else:
            parents = f.tiles[0]
            act = parents["position"][parents["next"] != -1]
            order = np.argsort(parents["next"][parents["next"] != -1])
            positions = []
            for p in act[order]:
                px, py = p % 2, p // 2
                for s in range(64):
                    positions.append((px * 8 + s % 8) + (py * 8 + s // 8) * tps)
            positions = np.array(positions)
        for t, pos in zip(tiles, positions):
            x0, y0 = (pos % tps) / tps * 2 - 1, (pos // tps) / tps * 2 - 1
            x1, y1 = x0 + 2 / tps, y0 + 2 / tps
            lo, hi = d2_range(x0, x1, y0, y1)
            block = img[(pos // tps) * tile_px:(pos // tps + 1) * tile_px, (pos % tps) * tile_px:(pos % tps + 1) * tile_px]
            if t["position"] == -1:
                assert hi < 1.8 ** 2 or lo > 1.8 ** 2 or block.all() or not block.any()
                assert block.all() or not block.any()
            else:
                assert lo <= 1.8 ** 2 <= hi
    # the final image is the disc
    ys, xs = np.mgrid[0:S, 0:S]
    cx, cy = (xs + 0.5) / S * 2 - 1, (ys + 0.5) / S * 2 - 1
    d = np.sqrt((cx + 1) ** 2 + (cy + 1) ** 2) - 1.8
    sure = np.abs(d) > 1e-5
    assert np.array_equal(img[sure] != 0, d[sure] < 0)


def test_ring_clause_table(mpr, tapes):
    """benchmark/print_tape_table.cpp:29: max(sqrt(X*X+Y*Y) - 1, 0.5 - sqrt(X*X+Y*Y)).
    The front end shares the common sub-expression, X*X becomes SQUARE, constants become
    immediates, the non-constant operand of a commutative op is lhs, and `0.5 - s` is the
    IMM_RHS form (src/tape.cpp:122-180)."""
    rows = mpr.decode(tapes("ring").data)
    ops = [r[0] for r in rows]
    assert ops == ["INVALID", "SQUARE_LHS", "SQUARE_LHS", "ADD_LHS_RHS", "SQRT_LHS", "SUB_LHS_IMM", "SUB_IMM_RHS",
                   "MAX_LHS_RHS", "INVALID"]
    head, end = rows[0], rows[-1]
    assert head[1] != 0 and head[2] != 0 and head[3] == 0          # X and Y bound, Z unused
    assert rows[5][4] == 1.0 and rows[6][4] == 0.5
    assert rows[6][2] == 0 and rows[6][3] != 0                     # IMM_RHS: operand in rhs
    assert end[1] == rows[7][1]                                    # end clause names the root's slot
    # in-place reuse: slots are released before the output slot is chosen (src/tape.cpp:201-212)
    assert rows[1][1] == rows[1][2]


@pytest.mark.parametrize("name,clauses,minmax", [("prospero", 6056, 2354), ("involute_gear_2d", 1660, 374),
                                                 ("involute_gear_3d", 1735, 374), ("architecture", 1296, 488),
                                                 ("bear", 544, 27), ("hello_world", 328, 97)])
def test_benchmark_models_build(tapes, name, clauses, minmax):
    t = tapes(name)
    assert t.length == clauses + 2          # head + clauses + end (src/tape.cpp:99,213,220)
    assert t.num_choices == minmax
    assert t.flags == 0
    assert t.num_slots <= 128               # fits the reference kernels' slot files (src/context.cu:210)
    d = t.data
    assert (d[0] & 0xFF) == 0 and (d[-1] & 0xFF) == 0
    ops = d[1:-1] & 0xFF
    assert ops.min() >= 2 and ops.max() <= 26       # no JUMP / COPY in a fresh tape
    # every operand slot was written before it is read
    written = {int((d[0] >> 8) & 255), int((d[0] >> 16) & 255), int((d[0] >> 24) & 255)}
    for c in d[1:-1]:
        c = int(c)
        l, r, o = (c >> 16) & 255, (c >> 24) & 255, (c >> 8) & 255
        assert (l == 0 or l in written) and (r == 0 or r in written)
        assert o != 0
        written.add(o)


def test_rounding_formulation(orc):
    """RD(a op b) = -RU((-a) op' b): the round-up-only formulation equals true round-down."""
    assert orc.selftest_rounding(300000, 11) == 0


def test_interval_quirks(orc, mpr):
    """Quirks of inc/gpu_interval.hpp that must be reproduced."""
    OP = mpr.OP
    lo, hi, _ = orc.interval_op(OP["COS_LHS"], [0.1], [0.2])
    assert (lo[0], hi[0]) == (-1.0, 1.0)                         # :353
    lo, hi, _ = orc.interval_op(OP["SIN_LHS"], [0.1], [0.2])
    assert (lo[0], hi[0]) == (-1.0, 1.0)                         # :378-380
    lo, hi, _ = orc.interval_op(OP["LOG_LHS"], [-1.0], [2.0])
    assert lo[0] == 0.0 and abs(hi[0] - np.log(2.0)) < 1e-6     # :385-386
    lo, hi, _ = orc.interval_op(OP["SQRT_LHS"], [-2.0], [-1.0])
    assert np.isnan(lo[0]) and np.isnan(hi[0])                   # :297-298
    lo, hi, _ = orc.interval_op(OP["DIV_LHS_RHS"], [1.0], [2.0], [-1.0], [1.0])
    assert lo[0] == -np.inf and hi[0] == np.inf                  # :163-164
    lo, hi, ch = orc.interval_op(OP["MIN_LHS_RHS"], [0.0], [1.0], [2.0], [3.0])
    assert ch[0] == 1 and (lo[0], hi[0]) == (0.0, 1.0)           # :209-211
    lo, hi, ch = orc.interval_op(OP["MAX_LHS_IMM"], [0.0], [1.0], imm=5.0)
    assert ch[0] == 2 and (lo[0], hi[0]) == (5.0, 5.0)           # :246-248
    # outward rounding: 0.1 + 0.2 is not representable
    lo, hi, _ = orc.interval_op(OP["ADD_LHS_RHS"], [0.1], [0.1], [0.2], [0.2])
    assert lo[0] < hi[0] and np.nextafter(lo[0], np.float32(1)) == hi[0]


def test_pool_overflow_keeps_parent_tape(orc, tapes):
    tape = tapes("hello_world").data
    full = orc.Frame(tape, 2, 256, view2().T.reshape(-1))
    small = orc.Frame(tape, 2, 256, view2().T.reshape(-1), pool_clauses=tape.size + 64 * 40)
    assert small.counters["pool_overflowed"] == 1
    assert np.array_equal(small.image, full.image)


def test_golden_frames(orc, tapes, mpr):
    """The committed goldens (tests/golden/frames.json) still describe what the oracle computes."""
    import hashlib
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "frames.json")) as f:
        golden = json.load(f)
    for g in golden["frames"]:
        if g["size"] > 512:
            continue                       # the 1024^2 frame is checked on the GPU side only
        tape = tapes(g["model"]).data
        assert hashlib.sha256(tape.tobytes()).hexdigest() == g["tape_sha256"]
        mat = view2() if g["dim"] == 2 else view3()
        fr = orc.Frame(tape, g["dim"], g["size"], mpr.colmajor(mat, g["dim"] + 1), threads=0, keep_pool=False)
        assert hashlib.sha256(fr.image.tobytes()).hexdigest() == g["image_sha256"], g["model"]
        assert fr.counters["tiles_in"] == g["tiles_in"] and fr.counters["tiles_active"] == g["tiles_active"]
        if g["dim"] == 3:
            assert hashlib.sha256(fr.normals.tobytes()).hexdigest() == g["normals_sha256"], g["model"]
