import sys, os, time; sys.path.insert(0, ".")
import numpy as np, torch, mpr_amd as m
from mpr_amd.multigpu import TileParallelRenderer, column_weights
name, S, world = "bear", 1024, 8
tape = m.Tape(m.model(name)); T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
ctx = m.Context(S, flags=m.CTX_TIMING)
ctx.render3D(tape, T)
w = column_weights(ctx.stages[3].tiles, S, 3)
owner = m.partition_columns((S // 64) ** 2, world, w)
print("weights per rank", [float(w[owner == r].sum()) for r in range(world)])
for rank in (0, 3, 7):
    for _ in range(3): ctx.render3D_part(tape, T, owner, rank)
    acc = {}
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.render3D_part(tape, T, owner, rank)
        for k, v in ctx.timings(): acc[k] = acc.get(k, 0) + v / 10
    print("rank", rank, "%.3f ms/frame" % ((time.perf_counter() - t0) / 10 * 1e3), " ".join("%s=%.3f" % kv for kv in acc.items()))
