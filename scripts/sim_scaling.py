"""Per-rank frame time of the tile-parallel loop for world sizes 1..8, measured on ONE GPU by playing each rank in turn
without the collective (development aid: what scaling the compute side allows), for the default column deal (the first tile
stage's ambiguous tiles per column: no frame in advance) and for the deal on a previous frame's measured work (feedback).

    python scripts/sim_scaling.py bear:1024 bear:2048 architecture:2048
"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mpr_amd as m
from mpr_amd.multigpu import TileParallelRenderer, column_weights


def mk(n):
    t = torch.zeros(n, dtype=torch.int32, device="cuda"); torch.cuda.synchronize(); return t, t.data_ptr()


for spec in (sys.argv[1:] or ["bear:1024"]):
    name, S = spec.split(":"); S = int(S)
    tape = m.Tape(m.model(name)); T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
    ctx = m.Context(S)
    ctx.render3D(tape, T)
    true_w = column_weights(ctx.stages[3].tiles, S, 3)            # smallest tiles per column: the float pass's work
    for feedback in (False, True):
        base = None
        for world in (1, 2, 4, 8):
            worst, times = 0.0, []
            for rank in range(world):
                tpr = TileParallelRenderer(ctx, m, rank, world, mk, lambda o, i: None, dim=3)
                tpr.plan(tape, T, feedback=feedback)
                for _ in range(5): tpr.render(tape, T)
                per = []
                for _ in range(20):
                    t0 = time.perf_counter()
                    tpr.render(tape, T)
                    per.append((time.perf_counter() - t0) * 1e3)
                times.append(float(np.median(per)))              # (median: a box shared with other jobs throws in a stray slow frame now and then)
                if os.environ.get("MPR_SIM_VERBOSE"):
                    print("   rank %d: median %.3f ms, max %.3f, mean %.3f  %s  last stage pushed: %s  resident %.0f MB" % (rank, times[-1], max(per), float(np.mean(per)), ctx.float_kernel(), ctx.last_stage_pushed(), ctx.resident_bytes() / 2**20), flush=True)
            worst = max(times)
            share = np.array([true_w[tpr.owner == r].sum() for r in range(world)]) / max(true_w.sum(), 1)
            base = base or worst
            print("%s %d^3 %s world %d: slowest rank %.3f ms/frame, mean %.3f (without the collective) -> %.2fx; largest share of the "
                  "smallest tiles %.3f (ideal %.3f)" % (name, S, "feedback deal" if feedback else "stage-0 proxy deal", world, worst,
                                                        float(np.mean(times)), base / worst, float(share.max()), 1.0 / world), flush=True)
    ctx.close()
