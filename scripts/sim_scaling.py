"""Per-rank frame time of the tile-parallel loop for world sizes 1..8, measured on ONE GPU by playing
each rank in turn without the collective (development aid: what scaling the compute side allows)."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mpr_amd as m
from mpr_amd.multigpu import TileParallelRenderer
name, S = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("bear", 1024)
tape = m.Tape(m.model(name)); T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
ctx = m.Context(S)
def mk(n):
    t = torch.zeros(n, dtype=torch.int32, device="cuda"); torch.cuda.synchronize(); return t, t.data_ptr()
base = None
for world in (1, 2, 4, 8):
    worst = 0.0
    for rank in range(world):
        tpr = TileParallelRenderer(ctx, m, rank, world, mk, lambda o, i: None, dim=3)
        tpr.plan(tape, T)
        for _ in range(5): tpr.render(tape, T)
        t0 = time.perf_counter()
        for _ in range(20): tpr.render(tape, T)
        worst = max(worst, (time.perf_counter() - t0) / 20 * 1e3)
    base = base or worst
    print("%s %d^3 world %d: slowest rank %.3f ms/frame (without the collective) -> %.2fx" % (name, S, world, worst, base / worst), flush=True)
