"""MPR_CTX_PARANOID: a context that renders every frame that took a shortcut a second time, the reference's way, and compares the two
frames' heights and normals on the device (csrc/context.hip: render_checked) — the equivalence argument of the fast frames checked
frame by frame instead of trusted; helpers.check_default_path runs it for every shape and model the suite renders the default way."""
import numpy as np
import pytest

from conftest import view2, view3

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,dim,S", [("bear", 3, 256), ("bear", 3, 512), ("architecture", 3, 256), ("hello_world", 3, 256), ("involute_gear_3d", 3, 256),
                                        ("prospero", 2, 512), ("involute_gear_2d", 2, 1024), ("trig", 3, 128), ("smooth", 3, 256)])
def test_both_renderings_of_every_frame_agree(mpr, orc, tapes, name, dim, S):
    tape = tapes(name)
    mat = view3() if dim == 3 else view2()
    ref = orc.Frame(tape.data, dim, S, mpr.colmajor(mat, dim + 1), threads=0, keep_pool=False)
    ctx = mpr.Context(S, flags=mpr.CTX_PARANOID)
    for k in range(4):
        if dim == 3:
            ctx.render3D(tape, mat)
            assert np.array_equal(ctx.normals, ref.normals)
        else:
            ctx.render2D(tape, mat)
        assert np.array_equal(ctx.image, ref.filled[3])
        # the context is left holding the reference's frame: a reader needs no third rendering
        assert ctx.last_stage_pushed()
    frames, again, cells = ctx.paranoid_stats()
    assert frames == 4 and cells == 0, (frames, again, cells)
    if name == "bear":
        assert again == 4, again            # (generated code, no tapes from the last stage: every frame of bear takes a shortcut)
    if name == "smooth":
        assert again >= 1, again
    ctx.close()


def test_a_shortcut_that_is_not_the_references_procedure_is_caught(mpr, orc, monkeypatch):
    """The shape of test_gpu_render.py::test_frames_that_start_at_the_16_tiles_are_verified_against_the_64_tiles with the verification
    switched off (development switch): its frame from the 16^3 tiles down is NOT the reference's, and the second rendering says so."""
    import test_gpu_fuzz_shapes
    tape = test_gpu_fuzz_shapes.fuzz_tape(mpr, 14, 12)
    ref = orc.Frame(tape.data, 3, 128, mpr.colmajor(view3(), 4), threads=0)
    monkeypatch.setenv("MPR_SKIP0_CHECK", "0")
    ctx = mpr.Context(128, flags=mpr.CTX_PARANOID)
    ctx.render3D(tape, view3())
    frames, again, cells = ctx.paranoid_stats()
    assert (frames, again) == (1, 1) and cells > 0, (frames, again, cells)
    # ... and what is left behind is the second rendering: the oracle's frame
    assert np.array_equal(ctx.image, ref.filled[3]) and np.array_equal(ctx.normals, ref.normals)
    ctx.close()
    monkeypatch.delenv("MPR_SKIP0_CHECK")
    ctx = mpr.Context(128, flags=mpr.CTX_PARANOID)
    for _ in range(2):
        ctx.render3D(tape, view3())
    assert ctx.paranoid_stats()[2] == 0
    ctx.close()


def test_frames_that_do_not_block(mpr, orc, tapes):
    tape = tapes("bear")
    ref = orc.Frame(tape.data, 3, 256, mpr.colmajor(view3(), 4), threads=0, keep_pool=False)
    ctx = mpr.Context(256, flags=mpr.CTX_PARANOID)
    for _ in range(2):
        ctx.render3D(tape, view3(), blocking=False)
        ctx.sync()
        assert np.array_equal(ctx.image, ref.filled[3]) and np.array_equal(ctx.normals, ref.normals)
    assert ctx.paranoid_stats()[2] == 0 and ctx.paranoid_stats()[0] == 2
    ctx.close()
