"""Randomised expressions: seeded CSG-like trees over every opcode, rendered in 2-D and 3-D and
compared with the oracle bit for bit (images, normals, per-stage tile sets, shortened tapes).

The benchmark models each use a handful of opcodes in fixed patterns; these trees mix all of them
— transcendental operands of min / max, divisions by intervals that straddle zero, NaN-producing
domains (sqrt / log / asin / acos of values that leave them), long chains on one slot and wide fans —
so that the interpreters' handler tables (operands from the slot file / forwarded, results stored /
dropped, compiled routines called from inside the loops) meet in combinations the models never
produce.  Every expression is rendered twice over: instrumented frames (compiled interpreters, every tape
pushed) for the full comparison, and the default path (generated code in the float pass, level-parallel
tile stages, repeated frames without last-stage tapes) for heights / occupancy and normals."""
import numpy as np
import pytest

from conftest import view2, view3
from helpers import check_default_path, compare_frame

pytestmark = pytest.mark.gpu


def random_tree(m, seed, depth=4):
    rng = np.random.default_rng(seed)
    X, Y, Z = m.Tree.X(), m.Tree.Y(), m.Tree.Z()

    def u(lo, hi):
        return float(np.float32(rng.uniform(lo, hi)))

    def coords():
        # a random affine change of coordinates (kept mild so that the shapes stay in view)
        a, b, c = u(0.7, 1.4), u(0.7, 1.4), u(0.7, 1.4)
        s = u(-0.3, 0.3)
        return (X * a + Y * s + u(-0.4, 0.4), Y * b - X * s + u(-0.4, 0.4), Z * c + u(-0.4, 0.4))

    def leaf():
        x, y, z = coords()
        k = int(rng.integers(0, 12))
        if k == 0:      # sphere
            return m.sqrt(m.square(x) + m.square(y) + m.square(z)) - u(0.2, 0.6)
        if k == 1:      # box
            return m.tmax(m.tabs(x) - u(0.1, 0.5), m.tmax(m.tabs(y) - u(0.1, 0.5), m.tabs(z) - u(0.1, 0.5)))
        if k == 2:      # torus
            return m.sqrt(m.square(m.sqrt(x * x + y * y) - u(0.3, 0.5)) + z * z) - u(0.05, 0.2)
        if k == 3:      # wavy sheet
            return z - u(0.05, 0.3) * m.sin(x * u(2, 9)) * m.cos(y * u(2, 9))
        if k == 4:      # gaussian blob: exp of a negative square, subtracted from a level
            return u(0.2, 0.7) - m.exp(-(x * x + y * y + z * z) * u(1.0, 6.0))
        if k == 5:      # log of a sum that can reach zero or below: NaN / -inf intervals
            return m.log(x * x + y * y + u(-0.05, 0.4)) + z * z - u(-1.5, 0.5)
        if k == 6:      # atan of a quotient whose divisor straddles zero
            return m.atan(y / (x + u(-0.2, 0.2))) * u(0.1, 0.4) + z - u(-0.3, 0.3)
        if k == 7:      # asin / acos partly outside their domain
            return m.asin(x * u(0.5, 1.6)) * m.acos(y * u(0.5, 1.6)) * u(0.05, 0.3) + z * z - u(0.1, 0.5)
        if k == 8:      # sqrt of a possibly negative quantity
            return m.sqrt(u(0.1, 0.5) - x * y) - z - u(0.0, 0.6)
        if k == 9:      # rational
            return x / (y * y + u(0.2, 1.5)) - z * u(0.5, 2.0) + u(-0.3, 0.3)
        if k == 10:     # immediate-on-the-left forms: imm - rhs, imm / rhs
            return u(0.2, 0.8) - u(0.3, 1.2) / (m.square(x) + m.square(y) + u(0.3, 1.0)) - m.tabs(z)
        return m.tmin(m.tabs(x) + m.tabs(y) + m.tabs(z) - u(0.3, 0.8), m.cos(x * u(1, 5)) + z)     # octahedron / cosine wall

    def node(d):
        if d == 0 or (d < depth and rng.uniform() < 0.15):
            return leaf()
        a, b = node(d - 1), node(d - 1)
        k = int(rng.integers(0, 7))
        if k == 0:
            return m.tmin(a, b)
        if k == 1:
            return m.tmax(a, b)
        if k == 2:
            return m.tmax(a, -b)                      # difference
        if k == 3:
            return m.tmin(a, b) - u(0.0, 0.1)         # union, inflated
        if k == 4:
            return a + b - m.sqrt(a * a + b * b + u(0.001, 0.05))      # smooth union
        if k == 5:
            return m.tmax(m.tmax(m.tmin(a, u(-0.2, 0.6)), u(-0.8, -0.1)), m.tmin(b / u(0.5, 2.0), a * u(0.5, 1.5)))   # immediates
        return m.tmin(a * u(0.5, 2.0), b + u(-0.2, 0.2))

    return node(depth)


@pytest.mark.parametrize("seed", range(40))
def test_random_expression_2d(mpr, orc, seed):
    tape = mpr.Tape(random_tree(mpr, 1000 + seed))
    assert tape.length > 20
    z = float(np.float32(0.05 * (seed % 5) - 0.1))
    cnt, ref = compare_frame(mpr, orc, tape, 2, 256, view2(), z=z)
    check_default_path(mpr, ref, tape, 2, 256, view2(), z=z)     # generated code, repeated frames


@pytest.mark.parametrize("seed", range(40))
def test_random_expression_3d(mpr, orc, seed):
    tape = mpr.Tape(random_tree(mpr, 2000 + seed))
    cnt, ref = compare_frame(mpr, orc, tape, 3, 128, view3())
    check_default_path(mpr, ref, tape, 3, 128, view3())


@pytest.mark.parametrize("seed", range(12))
def test_random_expression_3d_general_view(mpr, orc, seed):
    rng = np.random.default_rng(3000 + seed)
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] += rng.uniform(-0.15, 0.15, (3, 3)).astype(np.float32)
    T[:3, 3] = rng.uniform(-0.1, 0.1, 3).astype(np.float32)
    T[3, 2] = 0.25
    tape = mpr.Tape(random_tree(mpr, 3000 + seed, depth=3))
    cnt, ref = compare_frame(mpr, orc, tape, 3, 128, T)
    check_default_path(mpr, ref, tape, 3, 128, T)


@pytest.mark.parametrize("seed", range(24))
def test_random_expression_3d_on_group_tapes(mpr, orc, seed, monkeypatch):
    """At this size the last tile stage would run level-parallel (few tiles) and the float pass on per-tile tapes; with
    the serial last stage the float pass takes the group form where the tapes allow it and repeated frames push no
    tapes in the last stage: the decisions of every tile applied in generated code (float pass) and by the normals
    interpreter, on expressions with min / max of everything."""
    monkeypatch.setenv("MPR_WIDE_LATER", "0")
    tape = mpr.Tape(random_tree(mpr, 2000 + seed))
    ref = orc.Frame(tape.data, 3, 128, mpr.colmajor(view3(), 4), threads=0)
    check_default_path(mpr, ref, tape, 3, 128, view3(), frames=3)
    monkeypatch.setenv("MPR_VOXEL_GROUPS", "2")          # ... and whatever the tapes' lengths
    check_default_path(mpr, ref, tape, 3, 128, view3(), frames=2)
