"""The oracle and the HIP path against an evaluator neither of them shares code with.

tests/golden/independent_*.npz come from tests/golden/make_independent.py: its own .frep parser,
the expression DAG evaluated node by node in numpy float64 at every voxel centre (no tape, no
slots, no tiles), with a running bound on what a faithful float32 evaluation may differ by.  Per
pixel that yields the interval [hmin, hmax] the heightmap must lie in — equal wherever no voxel
of the column is within the bound of zero; the others are the fragility mask of SURVEY.md
section 8(c) — and a float64 normal.  Assertions:
  * outside the mask the heightmap is EQUAL to the independent one, inside hmin <= h <= hmax;
  * the mask is small (it is reported by the generator and bounded here);
  * normals agree within 1 LSB per channel wherever the gradient is well defined in float32.
The CPU half (oracle) runs everywhere; the GPU half is marked gpu and also covers the BASELINE
sizes through sampled pixel columns (bear 1024^3, architecture 2048^3, gears 4096^2).
"""
import os

import numpy as np
import pytest

from conftest import view2, view3

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, "independent_" + name + ".npz"))
    d = {k: z[k] for k in z.files}
    d["model"], d["dim"], d["size"] = str(d["model"]), int(d["dim"]), int(d["size"])
    n = d["pixels"].size or d["size"] ** 2
    if d["dim"] == 3:
        d["nfrag"] = np.unpackbits(d["nfrag"])[:n].astype(bool)
    else:
        d["lo"] = np.unpackbits(d["lo"])[:n]
        d["hi"] = np.unpackbits(d["hi"])[:n]
    return d


def pick(img, d):
    flat = np.asarray(img).reshape(-1)
    return flat[d["pixels"]] if d["pixels"].size else flat


def check_heights(h, d, max_fragile=0.01):
    h = pick(h, d).astype(np.int64)
    hmin, hmax = d["hmin"].astype(np.int64), d["hmax"].astype(np.int64)
    firm = hmin == hmax
    assert firm.mean() > 1 - max_fragile, "fragility mask covers %.3f%% of the pixels" % (100 * (1 - firm.mean()))
    bad = np.flatnonzero(firm & (h != hmin))
    assert bad.size == 0, "heightmap differs from the independent evaluation at %d firm pixels, e.g. %s" % (
        bad.size, [(int(i), int(h[i]), int(hmin[i])) for i in bad[:5]])
    assert ((h >= hmin) & (h <= hmax)).all(), "heightmap leaves [hmin, hmax] inside the fragility mask"
    assert (hmin > 0).sum() > 100


def check_normals(n, h, d, max_fragile=0.12, identical=0.97):
    n = pick(n, d).astype(np.int64)
    h = pick(h, d)
    want = d["normal"].astype(np.int64)
    use = (~d["nfrag"]) & (d["hmin"] > 0) & (h == d["hmin"])
    filled = d["hmin"] > 0
    assert use.sum() > (1 - max_fragile) * filled.sum(), "only %d of %d filled pixels have a float32-stable gradient" % (use.sum(), filled.sum())
    assert ((n[use] >> 24) == 0xFF).all()
    for shift in (0, 8, 16):
        a, b = (n[use] >> shift) & 0xFF, (want[use] >> shift) & 0xFF
        worst = np.abs(a - b).max() if use.any() else 0
        assert worst <= 1, "normal channel %d differs by %d LSB" % (shift // 8, worst)
    # and they are not all off by one: the vast majority is identical
    assert (n[use] == want[use]).mean() > identical


def check_image2d(img, d, max_fragile=0.002):
    v = pick(img, d)
    lo, hi = d["lo"], d["hi"]
    firm = lo == hi
    assert firm.mean() > 1 - max_fragile
    assert np.array_equal(v[firm] != 0, lo[firm] != 0)
    assert ((v != 0) >= (lo != 0)).all() and ((v != 0) <= (hi != 0)).all()
    assert 0 < lo.sum() < lo.size


# ---------------------------------------------------------------------------------------------
# CPU: the oracle
# shape_*: random shapes of tests/test_gpu_fuzz_shapes.py (divisions by negative constants, steep exp / log blends, asin / acos inside
# their domain), tests/golden/make_shapes.py.  (Shapes whose asin / acos leave the domain inside the view are not here: the
# reference's hierarchy then draws what no voxel-by-voxel evaluation predicts — tests/golden/make_independent.py.)
SHAPES_3D = ["shape_12_0_3d_128", "shape_12_3_3d_128", "shape_4_15_3d_128", "shape_4_9_3d_128", "shape_12_21_3d_128", "shape_64_5_3d_128"]
SHAPES_2D = ["shape_4_15_2d_256", "shape_4_9_2d_256", "shape_12_21_2d_256", "shape_64_5_2d_256"]
CPU_3D = ["two_spheres_3d_128", "hello_world_3d_128", "bear_3d_128", "architecture_3d_128", "involute_gear_3d_128"] + SHAPES_3D
CPU_2D = ["circle_2d_256", "hello_world_2d_256", "prospero_2d_512", "involute_gear_2d_2d_512"] + SHAPES_2D
# (blends: exp and log of this package against numpy's; more normals one unit apart than in the models, none further)
NORMALS_IDENTICAL = {n: 0.9 for n in SHAPES_3D}


@pytest.mark.parametrize("name", CPU_3D)
def test_oracle_3d_against_independent_evaluator(mpr, orc, tapes, name):
    d = load(name)
    ref = orc.Frame(tapes(d["model"]).data, 3, d["size"], mpr.colmajor(view3(), 4), threads=0, keep_pool=False)
    check_heights(ref.image, d)
    check_normals(ref.normals, ref.image, d, identical=NORMALS_IDENTICAL.get(name, 0.97))


@pytest.mark.parametrize("name", CPU_2D)
def test_oracle_2d_against_independent_evaluator(mpr, orc, tapes, name):
    d = load(name)
    ref = orc.Frame(tapes(d["model"]).data, 2, d["size"], mpr.colmajor(view2(), 3), threads=0, keep_pool=False)
    check_image2d(ref.image, d)


# ---------------------------------------------------------------------------------------------
# GPU: the HIP path, including the BASELINE sizes through sampled columns
GPU_3D = CPU_3D + ["bear_3d_256", "architecture_3d_256", "bear_3d_1024_sample", "architecture_3d_2048_sample"]
GPU_2D = CPU_2D + ["prospero_2d_1024", "involute_gear_2d_2d_4096_sample"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", GPU_3D)
def test_gpu_3d_against_independent_evaluator(mpr, tapes, name):
    d = load(name)
    ctx = mpr.Context(d["size"])
    ctx.render3D(tapes(d["model"]), view3())
    h, n = ctx.image, ctx.normals
    ctx.close()
    check_heights(h, d)
    check_normals(n, h, d, identical=NORMALS_IDENTICAL.get(name, 0.97))


@pytest.mark.gpu
@pytest.mark.parametrize("name", GPU_2D)
def test_gpu_2d_against_independent_evaluator(mpr, tapes, name):
    d = load(name)
    ctx = mpr.Context(d["size"])
    ctx.render2D(tapes(d["model"]), view2())
    img = ctx.image
    ctx.close()
    check_image2d(img, d)
