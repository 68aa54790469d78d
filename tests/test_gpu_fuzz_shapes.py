"""Random shapes against the oracle, on the paths bear takes.  tests/test_gpu_fuzz.py's deep random trees have wide DAGs: their
frames run the level-parallel first stage and the interpreters; the fixed models exercise what their authors wrote — bear divides
by positive constants only, and a routine that took an interval's ends from the wrong side for a NEGATIVE divisor passed every one
of them (round 4: the loose division of the tile stages).  Here: seeded random CHAINS of primitives (spheres, boxes, tori, planes,
gyroid-ish waves, asin / acos that leave their domain inside the view), constants of either sign, every unary opcode, hard and
smooth (exp / log) unions — narrow DAGs of at most 24 slots like bear's: host-generated code in every stage, frames that start at
the 16^3 tiles (which is how the shapes were found whose 64^3 tiles bind their children: test_gpu_render.py:
test_frames_that_start_at_the_16_tiles_are_verified_against_the_64_tiles), loose enclosures — as the caller's default frames, as
frames somebody reads, and as instrumented frames; 2-D and 3-D; 2 to 64 primitives (up to 1200 clauses and beyond 64 min / max
clauses: off the generated paths again)."""
import zlib

import numpy as np
import pytest

from conftest import view2, view3
from helpers import check_default_path, compare_frame, compare_reader_frame

pytestmark = pytest.mark.gpu


def random_tree(mpr, rng, size):
    X, Y, Z = mpr.Tree.X(), mpr.Tree.Y(), mpr.Tree.Z()

    def const(lo=0.1, hi=1.0, signed=False):
        v = float(np.float32(rng.uniform(lo, hi)))
        return -v if signed and rng.random() < 0.5 else v

    def point():
        # a moved, scaled (by a constant of either sign: a mirror), sometimes sheared copy of the axes
        s = const(0.6, 1.6, signed=True)
        x = (X - const(-0.5, 0.5)) / s
        y = (Y - const(-0.5, 0.5)) * const(0.7, 1.5, signed=True)
        z = (Z - const(-0.4, 0.4)) / const(0.7, 1.4, signed=True)
        if rng.random() < 0.3:
            x = x + y * const(0.1, 0.4, signed=True)
        return x, y, z

    def primitive():
        x, y, z = point()
        k = rng.integers(0, 9)
        if k == 0:
            return mpr.sqrt(x * x + y * y + z * z) - const(0.2, 0.6)
        if k == 1:      # a box
            return mpr.tmax(mpr.tmax(mpr.tabs(x) - const(0.1, 0.5), mpr.tabs(y) - const(0.1, 0.5)), mpr.tabs(z) - const(0.1, 0.5))
        if k == 2:      # a torus
            q = mpr.sqrt(mpr.square(x) + mpr.square(y)) - const(0.3, 0.5)
            return mpr.sqrt(q * q + z * z) - const(0.05, 0.2)
        if k == 3:      # a slab between two planes
            return mpr.tabs(x * const(0.2, 1.0, signed=True) + y * const(0.2, 1.0, signed=True) + z * const(0.2, 1.0)) - const(0.05, 0.3)
        if k == 4:      # waves
            f = const(3.0, 9.0)
            return mpr.sin(x * f) * mpr.cos(y * f) + mpr.sin(y * f) * mpr.cos(z * f) + mpr.sin(z * f) * mpr.cos(x * f) - const(-0.5, 0.8)
        if k == 5:      # a cylinder with a reciprocal profile
            return mpr.sqrt(x * x + y * y) - const(0.1, 0.3) / (z * z + const(0.3, 1.0))
        if k == 6:      # an ellipsoid by way of exp / log: log(exp(a) * exp(b)) = a + b
            return mpr.log(mpr.exp(x * x * const(1.0, 4.0)) * mpr.exp(y * y * const(1.0, 4.0))) + z * z - const(0.1, 0.5)
        if k == 7:      # a cone by atan
            return mpr.atan(mpr.sqrt(x * x + y * y) / (mpr.tabs(z) + 0.1)) - const(0.3, 1.0)
        return mpr.asin(x * 0.6) * mpr.acos(y * 0.6) + z * z - const(0.1, 0.6)       # (arguments inside [-1, 1] over most of the view)

    def combine(a, b):
        k = rng.integers(0, 8)
        if k <= 1:
            return mpr.tmin(a, b)
        if k == 2:
            return mpr.tmax(a, b)
        if k == 3:
            return mpr.tmax(a, -b)                       # difference
        if k == 4:      # a smooth union, steepness of either sign's divisor: -log(e^(-s a) + e^(-s b)) / s
            s = const(4.0, 48.0)
            return mpr.log(mpr.exp(a * -s) + mpr.exp(b * -s)) / -s
        if k == 5:
            s = const(4.0, 24.0)                         # ... and the smooth intersection
            return mpr.log(mpr.exp(a * s) + mpr.exp(b * s)) / s
        if k == 6:
            return mpr.tmin(a, b) - const(0.0, 0.05)     # an offset union
        return mpr.tmin(a + const(0.0, 0.1), mpr.tmax(b, a - const(0.05, 0.3)))      # a shell

    t = primitive()
    for _ in range(size - 1):
        t = combine(t, primitive())
    return t


def fuzz_tape(mpr, seed, size):
    rng = np.random.default_rng(zlib.crc32(b"fuzz") + seed)
    for _ in range(50):
        try:
            return mpr.Tape(random_tree(mpr, rng, size))
        except mpr.MprError:
            continue
    raise AssertionError("no tape")


@pytest.mark.parametrize("seed", range(24))
@pytest.mark.parametrize("size", [2, 4, 12, 64])
def test_random_shapes_default_frames_match_the_oracle(mpr, orc, seed, size):
    """What a caller gets (no counters, nobody reading): fast frames from the 16^3 tiles down, host-generated or device-translated
    float pass, loose enclosures in the tile stages where the tape allows them — heights and normals of three frames in a row."""
    tape = fuzz_tape(mpr, seed, size)
    S3, S2 = (256, 512) if seed % 3 == 0 else (128, 256)
    ref = orc.Frame(tape.data, 3, S3, mpr.colmajor(view3(), 4), threads=0)
    check_default_path(mpr, ref, tape, 3, S3, view3())
    ref2 = orc.Frame(tape.data, 2, S2, mpr.colmajor(view2(), 3), z=0.05, threads=0)
    check_default_path(mpr, ref2, tape, 2, S2, view2(), z=0.05)


@pytest.mark.parametrize("seed", range(8))
@pytest.mark.parametrize("size", [2, 4, 12, 64])
def test_random_shapes_instrumented_and_read_frames_match_the_oracle(mpr, orc, seed, size):
    """... the instrumented frame (every stage's images, survivor sets, shortened tapes, work counters), and what a reader of
    tiles and tapes gets after an ordinary frame."""
    tape = fuzz_tape(mpr, 100 + seed, size)
    compare_frame(mpr, orc, tape, 3, 128, view3())
    compare_frame(mpr, orc, tape, 2, 256, view2())
    compare_reader_frame(mpr, orc, tape, 128, view3())


def random_view3(rng):
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] += rng.uniform(-0.25, 0.25, (3, 3)).astype(np.float32)
    if rng.random() < 0.5:
        T[0] *= np.float32(-1.0)                       # a mirrored axis
    T[:3, 3] = rng.uniform(-0.15, 0.15, 3).astype(np.float32)
    T[3, :3] = rng.uniform(-0.3, 0.3, 3).astype(np.float32)       # perspective along any axis
    return T


@pytest.mark.parametrize("seed", range(16))
@pytest.mark.parametrize("size", [3, 12])
def test_random_shapes_in_general_views(mpr, orc, seed, size):
    """... in views the benchmarks never use (rotated, sheared, mirrored, perspective along any axis): the tile stages' interval
    transform of the axes, the float pass's and the normals pass's own, the whole-view domain check of csrc/frame_domain.cpp and the
    64^3 tiles' walk of the verification all take the matrix."""
    tape = fuzz_tape(mpr, 200 + seed, size)
    rng = np.random.default_rng(zlib.crc32(b"view") + seed)
    T = random_view3(rng)
    ref = orc.Frame(tape.data, 3, 128, mpr.colmajor(T, 4), threads=0)
    check_default_path(mpr, ref, tape, 3, 128, T)
    T2 = np.eye(3, dtype=np.float32)
    T2[:2, :2] += rng.uniform(-0.3, 0.3, (2, 2)).astype(np.float32)
    T2[:2, 2] = rng.uniform(-0.2, 0.2, 2).astype(np.float32)
    T2[2, :2] = rng.uniform(-0.2, 0.2, 2).astype(np.float32)
    z = float(np.float32(rng.uniform(-0.3, 0.3)))
    ref2 = orc.Frame(tape.data, 2, 256, mpr.colmajor(T2, 3), z=z, threads=0)
    check_default_path(mpr, ref2, tape, 2, 256, T2, z=z)


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5, 6, 7, "fails_its_verification"])
def test_random_shapes_dealt_to_three_ranks_on_one_device(mpr, orc, seed):
    """... and as the multi-GPU pipeline renders them: three contexts on one device, each its share of the 64 x 64 columns in a
    frame that does not block (its start at the 16^3 tiles is verified BEFORE its float pass is launched: the columns are packed
    behind it on the same stream), packed, exchanged by device copies, unpacked — every rank ends up with the oracle's frame."""
    torch = pytest.importorskip("torch")
    from mpr_amd.multigpu import TileParallelRenderer
    if seed == "fails_its_verification":       # (test_gpu_render.py: ..._are_verified_against_the_64_tiles)
        tape, S = fuzz_tape(mpr, 14, 12), 128
    else:
        tape, S = fuzz_tape(mpr, 300 + seed, 12 if seed % 2 else 4), 256
    T, world = view3(), 3
    ref = orc.Frame(tape.data, 3, S, mpr.colmajor(T, 4), threads=0)
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
            assert np.array_equal(ctxs[r].image, ref.filled[3]), (r, int((ctxs[r].image != ref.filled[3]).sum()))
            assert np.array_equal(ctxs[r].normals, ref.normals), (r, int((ctxs[r].normals != ref.normals).sum()))
    if seed == "fails_its_verification":
        assert sum(c.skip0_vetoes() for c in ctxs) >= 1
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("seed", range(4))
def test_random_shapes_at_1024(mpr, orc, seed):
    """... and at the size the bench line is quoted on: 4096 parents, 262 144 tiles in the verified first stage."""
    tape = fuzz_tape(mpr, 400 + seed, 12)
    ref = orc.Frame(tape.data, 3, 1024, mpr.colmajor(view3(), 4), threads=0)
    check_default_path(mpr, ref, tape, 3, 1024, view3(), frames=2)


def sweep_view(rng):
    """scripts/fuzz_sweep.py's views: the benchmark's, or a sheared / mirrored / moved / perspective one"""
    if rng.random() < 0.6:
        return view3()
    V = np.eye(4, dtype=np.float32)
    V[:3, :3] += rng.uniform(-0.25, 0.25, (3, 3)).astype(np.float32)
    if rng.random() < 0.5:
        V[0] *= np.float32(-1.0)
    V[:3, 3] = rng.uniform(-0.15, 0.15, 3).astype(np.float32)
    V[3, :3] = rng.uniform(-0.3, 0.3, 3).astype(np.float32)
    return V


@pytest.mark.parametrize("seed,size", [(1989, 8), (2074, 3), (2435, 3), (2762, 16), (2891, 3)])
def test_shapes_the_wide_sweep_found(mpr, orc, seed, size):
    """scripts/fuzz_sweep.py over seeds 1000..3039 (round 5; profiles/r05_fuzz_sweep.txt): five of its 6120 frames had the reference's
    heights and 46 to 4096 wrong NORMALS — in round 4's code too.  A pixel's normal is evaluated one voxel above its surface, on the
    tape of the deepest tile there (reference src/context.cu:1034-1066); where that voxel's 16^3 tile was culled, the reference
    takes its ambiguous 64^3 parent's shortened tape — the parent's decisions imposed — and a frame that starts at the 16^3 tiles
    took the root tape: the same normals only where every decision is a fact about the float values.  The normals pass now imposes
    the 64^3 tiles' decisions (which the verification walks anyway) on every pixel inside an ambiguous one."""
    tape = fuzz_tape(mpr, seed, size)
    rng = np.random.default_rng(seed * 7 + size)
    S = int(rng.choice([128, 256]))
    view = sweep_view(rng)
    ref = orc.Frame(tape.data, 3, S, mpr.colmajor(view, 4), threads=0)
    ctx = mpr.Context(S)
    started_below = False
    for k in range(3):
        ctx.render3D(tape, view)
        assert np.array_equal(ctx.image, ref.filled[3]), (k, ctx.tile_stage_forms())
        assert np.array_equal(ctx.normals, ref.normals), (k, int((ctx.normals != ref.normals).sum()), ctx.tile_stage_forms(), ctx.normals_kernel())
        started_below |= ctx.tile_stage_forms().startswith("1:")
    assert started_below or ctx.skip0_vetoes() > 0      # (the path the finding is about: frames that start at the 16^3 tiles)
    ctx.close()
