#!/usr/bin/env python
"""How many distinct tapes does a 4x4 pixel footprint of the normals pass walk?  (One wave walks every
distinct tape of its footprint with all 64 lanes; only the pixels on that tape use the result.)
    python tests/tools/normals_grouping.py bear 1024"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import mpr_amd as m
from oracle import orc

name = sys.argv[1] if len(sys.argv) > 1 else "bear"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
tape = m.Tape(m.model(name))
ctx = m.Context(S)
ctx.render3D(tape, T)
img = ctx.stages[3].filled.astype(np.int64)
t0, t1, t2 = ctx.stages[0].tiles, ctx.stages[1].tiles, ctx.stages[2].tiles
py, px = np.nonzero(img)
pz = img[py, px]
pz = np.where(pz < S - 1, pz + 1, pz)
t64 = S // 64
tile = px // 64 + (py // 64) * t64 + (pz // 64) * t64 * t64
tp = t0["tape"][tile].astype(np.int64); nx = t0["next"][tile].astype(np.int64)
deeper = nx != -1
sub = nx * 64 + (px % 64) // 16 + ((py % 64) // 16) * 4 + ((pz % 64) // 16) * 16
sub = np.where(deeper, sub, 0)
tp = np.where(deeper, t1["tape"][sub], tp); nx2 = np.where(deeper, t1["next"][sub], -1)
deeper2 = nx2 != -1
mic = np.where(deeper2, nx2 * 64 + (px % 16) // 4 + ((py % 16) // 4) * 4 + ((pz % 16) // 4) * 16, 0)
tp = np.where(deeper2, t2["tape"][mic], tp)
fp = (px // 4) + (py // 4) * (S // 4)
key = fp * (1 << 32) + tp
uk = np.unique(key)
walks = uk.size
fps = np.unique(fp).size
# clause-weighted
heads = (uk & 0xFFFFFFFF).astype(np.int64)
uh, inv = np.unique(heads, return_inverse=True)
tl = np.zeros(uh.size, dtype=orc.TILE_DTYPE)
tl["tape"] = uh
ln, _ = orc.tiles_digest(ctx.tape_data, tl)
walked = ln[inv].astype(np.int64).sum()
# ideal: pixels grouped by tape, 16 per wave
cnt = np.unique(tp, return_counts=True)
hl = dict(zip(uh.tolist(), ln.tolist()))
ideal = sum(((c + 15) // 16) * hl[int(h)] for h, c in zip(*cnt))
print("%s %d^3: %d filled pixels in %d footprints; %d tape walks (%.2f per footprint); %.1f M clauses walked, %.1f M if pixels were grouped by tape (x%.2f); %d distinct tapes" % (
    name, S, px.size, fps, walks, walks / fps, walked / 1e6, ideal / 1e6, walked / max(ideal, 1), uh.size))
