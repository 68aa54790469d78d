"""How much of the float pass's work is on tapes shared by several voxel tiles (development aid)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, mpr_amd as m
from oracle import orc
for name, dim, S in (("bear", 3, 1024), ("architecture", 3, 1024), ("involute_gear_3d", 3, 1024)):
    c = m.Context(S); T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
    t = m.Tape(m.model(name))
    (c.render3D(t, T) if dim == 3 else c.render2D(t))
    tl = c.stages[3].tiles
    pool = c.tape_data
    u, inv, cnt = np.unique(tl["tape"], return_inverse=True, return_counts=True)
    rep = np.zeros(u.size, dtype=tl.dtype); rep["tape"] = u; rep["position"] = 0
    lens, _ = orc.tiles_digest(pool, rep)
    per_tile = lens[inv]
    shared = cnt[inv] > 1
    print(name, "tiles", tl.size, "clause visits", int(per_tile.sum()), "in shared tapes %.1f%%" % (100 * per_tile[shared].sum() / per_tile.sum()),
          "mean len shared %.0f own %.0f" % (per_tile[shared].mean() if shared.any() else 0, per_tile[~shared].mean()),
          "mean sharers %.1f" % (cnt[cnt > 1].mean() if (cnt > 1).any() else 0))
    # how many packs of K would run
    for K in (2, 4):
        packs = int(np.ceil(cnt[cnt > 1] / K).sum()); singles = int((cnt == 1).sum())
        work = (np.ceil(cnt / K) * lens).sum()
        print("   K=%d: wave-walks %d (vs %d), clause visits %.0f%% of now" % (K, packs + singles, tl.size, 100 * work / per_tile.sum()))
    c.close()
