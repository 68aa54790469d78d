"""Per-kernel times of ONE rank's part of a frame dealt to `world` ranks (development aid: where the per-rank floor of the
tile-parallel loop sits).   python scripts/part_breakdown.py bear:1024:8"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mpr_amd as m

for spec in (sys.argv[1:] or ["bear:1024:8"]):
    name, S, world = spec.split(":"); S = int(S); world = int(world)
    tape = m.Tape(m.model(name)); T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
    ctx = m.Context(S, flags=m.CTX_TIMING)
    w = ctx.column_weights(tape, T, 0.0, 3)
    owner = np.ascontiguousarray(m.partition_columns((S // 64) ** 2, world, w), dtype=np.int32)
    for rank in (0, world - 1):
        for _ in range(4): ctx.render3D_part(tape, T, owner, rank)
        acc, n = {}, 10
        t0 = time.perf_counter()
        for _ in range(n):
            ctx.render3D_part(tape, T, owner, rank)
            for k, v in ctx.timings(): acc[k] = acc.get(k, 0) + v / n
        dt = (time.perf_counter() - t0) / n * 1e3
        print("%s %d^3 rank %d of %d: %.3f ms/frame (with events)  " % (name, S, rank, world, dt) + " ".join("%s=%.3f" % kv for kv in acc.items()))
        print("   launches: " + " ".join("%s=%.3f" % kv for kv in ctx.timings()), flush=True)
    ctx.close()
    ctx = m.Context(S)
    for _ in range(5): ctx.render3D_part(tape, T, owner, 0)
    t0 = time.perf_counter()
    for _ in range(50): ctx.render3D_part(tape, T, owner, 0)
    print("   rank 0 without events: %.3f ms/frame" % ((time.perf_counter() - t0) / 50 * 1e3), flush=True)
    ctx.close()
