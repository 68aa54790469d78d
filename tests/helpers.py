"""Frame comparison shared by the GPU-vs-oracle tests."""
import numpy as np


def active_positions(tiles):
    return np.sort(tiles["position"][tiles["next"] != -1])


def compare_frame(mpr, orc, tape, dim, S, mat, z=0.0, check_tapes=True):
    ctx = mpr.Context(S, flags=mpr.CTX_COUNTERS)
    ref = orc.Frame(tape.data, dim, S, mpr.colmajor(mat, dim + 1), z=z, threads=0)
    if dim == 2:
        ctx.render2D(tape, mat, z)
    else:
        ctx.render3D(tape, mat)
    cnt = ctx.counters()
    assert cnt["pool_overflowed"] == 0 and ref.counters["pool_overflowed"] == 0
    stages = [0, 1, 2, 3] if dim == 3 else [0, 2, 3]
    for s in stages:
        got = ctx.stages[s].filled
        assert np.array_equal(got, ref.filled[s]), "filled image of stage %d differs (%d cells)" % (
            s, int((got != ref.filled[s]).sum()))
    if dim == 3:
        gn, rn = ctx.normals, ref.normals
        assert np.array_equal(gn, rn), "normals differ at %d pixels" % int((gn != rn).sum())
    # per-stage tile occupancy: the SET of surviving tiles (order is timing dependent)
    tile_stages = [0, 1, 2] if dim == 3 else [0, 2]
    pool = ctx.tape_data if check_tapes else None
    for k, s in enumerate(tile_stages):
        gt, rt = ctx.stages[s].tiles, ref.tiles[s]
        assert gt.size == rt.size
        last = (k == len(tile_stages) - 1)
        nxt = s + 1 if dim == 3 else (3 if s else 2)
        g_next, r_next = ctx.stages[nxt].tiles, ref.tiles[nxt]
        assert g_next.size == r_next.size, "stage %d hands %d tiles on, oracle %d" % (s, g_next.size, r_next.size)
        assert cnt["tiles_in"][k] == ref.counters["tiles_in"][k]
        assert cnt["tiles_active"][k] == ref.counters["tiles_active"][k]
        if not last:
            assert np.array_equal(active_positions(gt), active_positions(rt))
        # The list handed on has been evaluated in place by the NEXT stage (position = -1 for
        # tiles that died there, tape = the tile's own shortened tape if it pushed one).  Only
        # survivors are comparable: whether a tile that ends up occluded pushed a tape first is
        # timing dependent in the reference too (src/context.cu:299-305 vs :312).
        g_live, r_live = g_next[g_next["position"] != -1], r_next[r_next["position"] != -1]
        go, ro = np.argsort(g_live["position"]), np.argsort(r_live["position"])
        assert np.array_equal(g_live["position"][go], r_live["position"][ro]), "survivor sets differ after stage %d" % s
        if check_tapes and g_live.size:
            # the shortened tape every surviving tile carries: same clause sequence
            glen, ghash = orc.tiles_digest(pool, g_live[go])
            rlen, rhash = orc.tiles_digest(ref.pool, r_live[ro])
            assert np.array_equal(glen, rlen), "shortened tape lengths differ at stage %d" % s
            assert np.array_equal(ghash, rhash), "shortened tape contents differ at stage %d" % s
    # the float pass evaluates exactly the voxels/pixels of the surviving smallest tiles
    assert cnt["voxel_tiles"] == ref.counters["voxel_tiles"]
    ctx.close()
    return cnt, ref


def check_default_path(mpr, ref, tape, dim, S, mat, z=0.0, frames=3):
    """compare_frame runs instrumented frames (work counters on: compiled interpreters, every tape pushed).  This is the
    path a caller gets — generated code in the float pass, no tapes from the last tile stage where that pays, 3-D frames up to
    1024^3 of narrow DAGs from the 16^3 tiles down — against the same oracle frame: heights / occupancy and normals."""
    ctx = mpr.Context(S)
    kinds = []
    for _ in range(frames):
        if dim == 2:
            ctx.render2D(tape, mat, z)
        else:
            ctx.render3D(tape, mat)
        kinds.append((ctx.float_kernel(), ctx.last_stage_pushed()))
        assert np.array_equal(ctx.image, ref.filled[3]), "image of the default path differs (%d cells)" % int((ctx.image != ref.filled[3]).sum())
        if dim == 3:
            bad = int((ctx.normals != ref.normals).sum())
            assert bad == 0, "normals of the default path differ at %d pixels (%s)" % (bad, kinds[-1])
    ctx.close()
    # ... and the same frames in a context that renders every frame that took a shortcut a second time, the reference's way, and
    # compares the two on the device (MPR_CTX_PARANOID): what a caller without an oracle can ask for
    ctx = mpr.Context(S, flags=mpr.CTX_PARANOID)
    for _ in range(2):
        if dim == 2:
            ctx.render2D(tape, mat, z)
        else:
            ctx.render3D(tape, mat)
        assert np.array_equal(ctx.image, ref.filled[3])
    frames_seen, again, cells = ctx.paranoid_stats()
    assert frames_seen == 2 and cells == 0, (frames_seen, again, cells)
    ctx.close()
    return kinds


def compare_reader_frame(mpr, orc, tape, S, mat, ref=None, frames=1, read_first="tape_data"):
    """What a reader of `stages[k].tiles` / `tape_data` gets from a context WITHOUT work counters — compare_frame's contexts carry
    MPR_CTX_COUNTERS, which keeps every tile stage on the instrumented walks; this is the path callers of mpr.hpp are on (for
    tapes the host generates code for: every stage on that code, tapes pushed by TileGen::bwd_full).  `frames` 3-D frames are
    rendered first; reading then returns the reference's state (a frame that took a shortcut is rendered again the reference's
    way).  Checked against the oracle, stage by stage: stage images, survivor sets, and the clause sequence of the shortened
    tape every surviving tile carries; heights and normals.  Returns (ctx, ref); the caller closes ctx."""
    if ref is None:
        ref = orc.Frame(tape.data, 3, S, mpr.colmajor(mat, 4), threads=0)
    ctx = mpr.Context(S)
    for _ in range(frames):
        ctx.render3D(tape, mat)
        assert np.array_equal(ctx.image, ref.filled[3]), "heights differ (%d cells)" % int((ctx.image != ref.filled[3]).sum())
        assert np.array_equal(ctx.normals, ref.normals), "normals differ at %d pixels" % int((ctx.normals != ref.normals).sum())
    # a reader: from here on the context holds the reference's state (whatever is read first; the re-render runs the tile stages
    # only and must leave the frame's heights and normals as they are)
    if read_first == "filled0":
        assert np.array_equal(ctx.stages[0].filled, ref.filled[0]), "stages[0].filled read first differs"
    elif read_first == "tiles":
        assert ctx.stages[1].tiles.size == ref.tiles[1].size
    pool = ctx.tape_data
    assert ctx.last_stage_pushed()
    assert np.array_equal(ctx.image, ref.filled[3]), "heights changed by the reader's re-render"
    for s in (0, 1, 2, 3):
        assert np.array_equal(ctx.stages[s].filled, ref.filled[s]), "filled image of stage %d differs" % s
    assert np.array_equal(ctx.normals, ref.normals)
    for s in (0, 1, 2):
        gt, rt = ctx.stages[s].tiles, ref.tiles[s]
        assert gt.size == rt.size, "stage %d evaluates %d tiles, oracle %d" % (s, gt.size, rt.size)
        if s < 2:
            assert np.array_equal(active_positions(gt), active_positions(rt)), "tiles stage %d subdivides differ" % s
        g_next, r_next = ctx.stages[s + 1].tiles, ref.tiles[s + 1]
        assert g_next.size == r_next.size, "stage %d hands %d tiles on, oracle %d" % (s, g_next.size, r_next.size)
        g_live, r_live = g_next[g_next["position"] != -1], r_next[r_next["position"] != -1]
        go, ro = np.argsort(g_live["position"]), np.argsort(r_live["position"])
        assert np.array_equal(g_live["position"][go], r_live["position"][ro]), "survivor sets differ after stage %d" % s
        if g_live.size:
            glen, ghash = orc.tiles_digest(pool, g_live[go])
            rlen, rhash = orc.tiles_digest(ref.pool, r_live[ro])
            assert np.array_equal(glen, rlen), "shortened tape lengths differ after stage %d (%d tiles)" % (s, int((glen != rlen).sum()))
            assert np.array_equal(ghash, rhash), "shortened tape contents differ after stage %d (%d tiles)" % (s, int((ghash != rhash).sum()))
    return ctx, ref


# ---- the oracle's interval routines over tiles, clause by clause (test_frame_domain.py, scripts/isotone_study.py) ----
def oracle_axes_of_tiles(mpr, orc, pos, tps, mat):
    """the tile stages' axis intervals (reference src/context.cu:91-113): corners in round-to-nearest float, then interval arithmetic"""
    t = np.float32(tps)
    two, half = np.float32(2.0), np.float32(0.5)
    iv = []
    for k in range(3):
        p = pos[:, k].astype(np.float32)
        iv.append(((p / t - half) * two, ((p + np.float32(1)) / t - half) * two))
    rows = []
    for i in range(4):
        acc = None
        for k in range(3):
            lo, hi, _ = orc.interval_op(mpr.OP["MUL_LHS_IMM"], iv[k][0], iv[k][1], imm=float(mat[i + 4 * k]))
            acc = (lo, hi) if acc is None else orc.interval_op(mpr.OP["ADD_LHS_RHS"], acc[0], acc[1], lo, hi)[:2]
        rows.append(orc.interval_op(mpr.OP["ADD_LHS_IMM"], acc[0], acc[1], imm=float(mat[i + 12]))[:2])
    return [orc.interval_op(mpr.OP["DIV_LHS_RHS"], rows[k][0], rows[k][1], rows[3][0], rows[3][1])[:2] for k in range(3)]


def oracle_walk_tiles(mpr, orc, clauses, axes):
    """every clause's interval for every tile: (lo[n_clauses, n_tiles], hi)"""
    d = mpr.decode(clauses)
    n = axes[0][0].size
    slots = {}
    head = d[0]
    slots[head[1]], slots[head[2]], slots[head[3]] = axes[0], axes[1], axes[2]
    zero = (np.zeros(n, np.float32), np.zeros(n, np.float32))
    lo_all = np.full((len(d), n), np.nan, np.float32)
    hi_all = np.full((len(d), n), np.nan, np.float32)
    for i in range(1, len(d) - 1):
        name, out, lhs, rhs, imm = d[i]
        a = slots.get(lhs, zero)
        b = slots.get(rhs, zero)
        lo, hi, _ = orc.interval_op(mpr.OP[name], a[0], a[1], b[0], b[1], imm)
        slots[out] = (lo, hi)
        lo_all[i], hi_all[i] = lo, hi
    return d, lo_all, hi_all
