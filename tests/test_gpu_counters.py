"""The work counters of mpr_get_counters (F, R, W, lane-clauses: the numerators of every roofline figure bench.py prints,
SURVEY.md 8(d)) against the oracle's.  2-D frames are deterministic: every counter is equal.  In 3-D a tile can be culled
by a neighbour's fill before or after it pushed a tape, and a voxel pair is skipped when the heightmap it reads is already
above it (src/context.cu:299-305, :852-864), upstream as here: the counters lie between the oracle's order-free bounds."""
import numpy as np
import pytest

from conftest import view2, view3

pytestmark = pytest.mark.gpu


def frames(mpr, orc, tape, dim, S, mat):
    ctx = mpr.Context(S, flags=mpr.CTX_COUNTERS | mpr.CTX_SERIAL_STAGES)
    if dim == 2:
        ctx.render2D(tape, mat)
    else:
        ctx.render3D(tape, mat)
    got = ctx.counters()
    ctx.close()
    ref = orc.Frame(tape.data, dim, S, mpr.colmajor(mat, dim + 1), threads=0, keep_pool=False).counters
    return got, ref


@pytest.mark.parametrize("name,S", [("circle", 256), ("ring", 256), ("hello_world", 256), ("prospero", 512), ("prospero", 1024),
                                    ("involute_gear_2d", 512), ("architecture", 256), ("trig", 256)])
def test_counters_2d_equal_the_oracle(mpr, orc, tapes, name, S):
    got, ref = frames(mpr, orc, tapes(name), 2, S, view2())
    assert got["tiles_in"][:2] == ref["tiles_in"][:2] and got["tiles_active"][:2] == ref["tiles_active"][:2]
    assert got["voxel_tiles"] == ref["voxel_tiles"]
    assert got["clauses_fwd_voxels"] == ref["clauses_fwd_voxels"]
    assert got["clauses_fwd"] - got["clauses_fwd_voxels"] - got["clauses_fwd_normals"] == ref["clauses_fwd_tiles"]
    assert got["clauses_bwd"] == ref["clauses_bwd"]
    assert got["clauses_written"] == ref["clauses_written"]
    assert got["lane_clauses"] == ref["lane_clauses"]


@pytest.mark.parametrize("name,S", [("two_spheres", 256), ("hello_world", 256), ("bear", 256), ("architecture", 256),
                                    ("involute_gear_3d", 256), ("bear", 1024)])
def test_counters_3d_inside_the_oracles_bounds(mpr, orc, tapes, name, S):
    got, ref = frames(mpr, orc, tapes(name), 3, S, view3())
    assert got["tiles_in"] == ref["tiles_in"] and got["tiles_active"] == ref["tiles_active"]
    assert got["voxel_tiles"] == ref["voxel_tiles"]
    # the float pass walks a smallest tile unless all its voxel pairs are hidden when it runs
    assert ref["clauses_fwd_voxels_min"] <= got["clauses_fwd_voxels"] <= ref["clauses_fwd_voxels_max"]
    # every surviving tile that shortened its tape wrote it; tiles culled later in their stage may have as well
    evaluated = sum(ref["tiles_in"])
    assert ref["clauses_written_survivors"] <= got["clauses_written"]
    slack = ref["clauses_written"] - ref["clauses_written_survivors"]        # what the oracle's own order wrote for tiles culled later
    assert got["clauses_written"] <= ref["clauses_written_survivors"] + 4 * slack + evaluated // 100 + 64
    # forward / backward words of the tile stages: a group is walked when one of its tiles is still visible as its wave starts.
    # The oracle culls before a stage with the fills of EARLIER stages only (mask_filled_tiles as its own pass, :1335); here that
    # test is fused into the evaluation and also sees fills of waves of the same launch, so fewer groups may be walked, never more
    ft = got["clauses_fwd"] - got["clauses_fwd_voxels"] - got["clauses_fwd_normals"]
    assert 0.8 * ref["clauses_fwd_tiles"] <= ft <= ref["clauses_fwd_tiles"]
    # (... never more — up to the tiles that the oracle's own threads, in their order, found hidden between evaluating them and
    # classifying them, :312, and this run did not: a fraction of a percent either way)
    assert 0.8 * ref["clauses_bwd"] - 64 <= got["clauses_bwd"] <= 1.02 * ref["clauses_bwd"] + 64
    assert got["normal_pixels"] == ref["normal_pixels"]
