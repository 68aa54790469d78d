"""Cycle breakdown of the tile stages (MPR_DEBUG_TILES=4; development aid)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mpr_amd as m
specs = [a.split(":") for a in sys.argv[1:]] or [("bear", "3", "1024"), ("architecture", "3", "1024"), ("prospero", "2", "1024")]
for name, dim, S in specs:
    dim, S = int(dim), int(S)
    c = m.Context(S, flags=m.CTX_COUNTERS); T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
    t = m.Tape(m.model(name))
    (c.render3D(t, T) if dim == 3 else c.render2D(t))
    print(name, dim, S, flush=True); c.counters(); c.close()
