"""Shared fixtures.  `-m "not gpu"` runs here (no GPU); `-m gpu` runs on an MI355X box."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure); built on demand with gcc."""
    from oracle import orc as _orc
    _orc.lib()
    return _orc


@pytest.fixture(scope="session")
def mpr():
    """The product library.  Built in-tree when missing (hipcc cross-compiles without a GPU)."""
    import mpr_amd
    mpr_amd.build()
    mpr_amd.lib()
    return mpr_amd


@pytest.fixture(scope="session")
def tapes(mpr):
    """Tapes of the reference's benchmark models + the in-code expressions of its benchmarks."""
    cache = {}

    def get(name):
        if name in cache:
            return cache[name]
        X, Y, Z = mpr.Tree.X(), mpr.Tree.Y(), mpr.Tree.Z()
        if name == "circle":          # benchmark/circle.cpp:22-24
            t = mpr.sqrt((X + 1) * (X + 1) + (Y + 1) * (Y + 1)) - 1.8
        elif name == "two_spheres":   # benchmark/render_2d_table.cpp:44-45, brute.cu:87-91
            t = mpr.tmin(mpr.sqrt((X + 0.5) * (X + 0.5) + Y * Y + Z * Z) - 0.25,
                         mpr.sqrt((X - 0.5) * (X - 0.5) + Y * Y + Z * Z) - 0.25)
        elif name == "ring":          # benchmark/print_tape_table.cpp:29
            t = mpr.tmax(mpr.sqrt(X * X + Y * Y) - 1, 0.5 - mpr.sqrt(X * X + Y * Y))
        elif name == "sphere":
            t = mpr.sqrt(X * X + Y * Y + Z * Z) - 0.7
        elif name == "trig":          # exercises every transcendental opcode
            t = mpr.tmin(mpr.sin(X * 3) + mpr.cos(Y * 2) * 0.5 + mpr.atan(Z + X) * 0.3 - 0.2,
                         mpr.tmax(mpr.exp(X) * 0.2 - mpr.log(Y * Y + 1.5) + mpr.asin(X * 0.5) * mpr.acos(Y * 0.5) * 0.1,
                                  mpr.tabs(Z) - 0.8 + X / (Y * Y + 2.0)))
        elif name == "smooth":        # exp / log blends the way bear has them, steep enough that exp underflows far from the surface
            d1 = mpr.sqrt((X + 0.3) * (X + 0.3) + Y * Y + Z * Z) - 0.35
            d2 = mpr.sqrt((X - 0.3) * (X - 0.3) + (Y - 0.1) * (Y - 0.1) + Z * Z) - 0.3
            blend = mpr.log(mpr.exp(d1 * -64.0) + mpr.exp(d2 * -64.0)) / -64.0
            t = mpr.tmax(mpr.tmin(blend, mpr.sqrt(X * X + (Y + 0.6) * (Y + 0.6) + Z * Z) / 3.0 - 0.1), Z - 0.25)
        elif name == "trig_blend":    # sin / cos the way bear has them (a rotation by an angle that depends on the position) + a ripple, min / max, an exp / log blend
            ang = mpr.exp(-mpr.sqrt(X * X + Y * Y + Z * Z)) * 2.5
            xr = X * mpr.cos(ang) - Y * mpr.sin(ang)
            yr = X * mpr.sin(ang) + Y * mpr.cos(ang)
            bar = mpr.tmax(mpr.tmax(mpr.tabs(xr) - 0.55, mpr.tabs(yr) - 0.2), mpr.tabs(Z) - 0.35)
            ball = mpr.sqrt((X - 0.3) * (X - 0.3) + (Y + 0.4) * (Y + 0.4) + Z * Z) - 0.3 + mpr.sin(X * 14.0) * mpr.cos(Y * 9.0) * 0.04
            blend = mpr.log(mpr.exp(bar * -24.0) + mpr.exp(ball * -24.0)) / -24.0
            t = mpr.tmin(blend, mpr.sqrt(X * X + (Y - 0.6) * (Y - 0.6) + (Z - 0.2) * (Z - 0.2)) - 0.15)
        elif name == "many_slots":    # > 128 simultaneously live values: every s_i is used by a product and, later, a sum
            terms = [(X - (i % 13) * 0.11 + 0.6) * (Y + (i % 7) * 0.13 - 0.4) + Z * (0.01 * i) for i in range(150)]
            prod = terms[0]
            for s_ in terms[1:]:
                prod = mpr.tmax(prod * 0.5, s_)
            total = terms[0]
            for s_ in terms[1:]:
                total = total + s_
            t = mpr.tmin(prod - 0.2, total * 0.01 - 0.05)
        elif name.startswith("shape_"):   # tests/golden/shapes: random shapes of test_gpu_fuzz_shapes.py (tests/golden/make_shapes.py)
            t = mpr.Tree.from_frep(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "shapes", name + ".frep"))
        else:
            t = mpr.model(name)
        cache[name] = mpr.Tape(t)
        return cache[name]

    return get


def view3(perspective=0.3):
    """The 3-D benchmark view: identity with T(3,2) = 0.3 (benchmark/render_3d_table.cpp:48-49)."""
    T = np.eye(4, dtype=np.float32)
    T[3, 2] = perspective
    return T


def view2():
    return np.eye(3, dtype=np.float32)
