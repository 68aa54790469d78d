#!/usr/bin/env python
"""How much of the float pass works on voxels that end up behind the final heightmap?

For every smallest tile handed to the float pass: is it entirely behind the finished heightmap
(every column's height >= the tile's top voxel)?  Such a tile could be skipped if the tiles in
front of it had been evaluated first.  Also: per-lane fraction of voxel pairs behind the heightmap.
    python tests/tools/occlusion_potential.py bear 1024
"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import mpr_amd as m

name = sys.argv[1] if len(sys.argv) > 1 else "bear"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
tape = m.Tape(m.model(name))
ctx = m.Context(S)
ctx.render3D(tape, T)
img = ctx.stages[3].filled            # [y, x] heights
tiles = ctx.stages[3].tiles
pos = tiles["position"][: ]
pos = pos[pos >= 0]
tps = S // 4
x, y, z = pos % tps, (pos // tps) % tps, pos // (tps * tps)
colmin = img.reshape(tps, 4, tps, 4).min(axis=(1, 3))     # min height per 4x4 column block [ty, tx]
hidden = colmin[y, x] >= z * 4 + 3
print("%s %d^3: %d float-pass tiles, %d (%.1f%%) entirely behind the final heightmap" % (
    name, S, pos.size, int(hidden.sum()), 100.0 * hidden.mean()))
# tiles that contain the final surface voxel of at least one of their columns
colmax = img.reshape(tps, 4, tps, 4).max(axis=(1, 3))
front = (colmax[y, x] >= z * 4) & ~hidden
print("tiles touching the visible surface: %d (%.1f%%); in front of it (empty after all): %d (%.1f%%)" % (
    int(front.sum()), 100.0 * front.mean(), int((~front & ~hidden).sum()), 100.0 * (~front & ~hidden).mean()))
# tape length statistics of hidden vs other tiles would need the pool; report z histogram instead
print("tiles per occupied 4x4 column: %.2f" % (pos.size / max(1, np.unique(x + y * tps).size)))

# ---- work actually done by the float pass vs the work of the tiles that are not hidden ----
from oracle import orc
cctx = m.Context(S, flags=m.CTX_COUNTERS)
cctx.render3D(tape, T)
done = cctx.counters()["clauses_fwd_voxels"]
t3 = cctx.stages[3].tiles
t3 = t3[t3["position"] >= 0]
ln, _ = orc.tiles_digest(cctx.tape_data, t3)
p3 = t3["position"]
x3, y3, z3 = p3 % tps, (p3 // tps) % tps, p3 // (tps * tps)
hid3 = colmin[y3, x3] >= z3 * 4 + 3
tot = int(ln.astype(np.int64).sum())
print("MPR_ZSORT=%s: float pass walked %d words; all tiles %d clauses, not-hidden tiles %d (%.1f%%), hidden %d" % (
    os.environ.get("MPR_ZSORT", "default"), done, tot, int(ln[~hid3].sum()), 100.0 * ln[~hid3].sum() / tot, int(ln[hid3].sum())))
