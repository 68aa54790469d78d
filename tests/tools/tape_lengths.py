#!/usr/bin/env python
"""Length distribution of the tapes each tile stage walks (one wave walks one tape for the 64
children of a surviving parent; a stage lasts as long as its slowest wave when the machine is
not full).   python tests/tools/tape_lengths.py prospero:2:1024 involute_gear_3d:3:1024"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import mpr_amd as m
from oracle import orc

for spec in sys.argv[1:]:
    name, dim, S = spec.split(":"); dim = int(dim); S = int(S)
    T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
    tape = m.Tape(m.model(name))
    ctx = m.Context(S)
    (ctx.render3D(tape, T) if dim == 3 else ctx.render2D(tape))
    pool = ctx.tape_data
    print("%s %dD %d: root tape %d clauses" % (name, dim, S, tape.length - 2))
    for st in ([0, 1] if dim == 3 else [0]):
        t = ctx.stages[st].tiles
        t = t[t["next"] != -1]                  # survivors: their tapes are walked by the next stage
        ln, _ = orc.tiles_digest(pool, t)
        q = np.percentile(ln, [0, 25, 50, 75, 90, 99, 100]).astype(int)
        print("  stage after %d: %d waves; tape clauses min/25/50/75/90/99/max = %s, mean %.0f" % (st, t.size, q.tolist(), ln.mean()))
    ctx.close()
