#!/usr/bin/env python
"""What the tapes a stage walks actually need: highest slot index and number of min/max clauses,
against the root tape's (which size the LDS slot planes and the choice array today).
    python scripts/tape_needs.py architecture:3:1024 prospero:2:1024"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import mpr_amd as m

def walk(pool, head):
    i = head + 1
    mx, ch, n = 0, 0, 0
    used = set()
    while True:
        d = int(pool[i]); op = d & 0xFF
        if op == 0:
            mx = max(mx, (d >> 8) & 0xFF); used.add((d >> 8) & 0xFF); break
        if op == 1:
            j = (d >> 32) & 0xFFFFFFFF
            if j >= 1 << 31: j -= 1 << 32
            i += j + 1; continue
        mx = max(mx, (d >> 8) & 0xFF, (d >> 16) & 0xFF, (d >> 24) & 0xFF)
        used.update(((d >> 8) & 0xFF, (d >> 16) & 0xFF, (d >> 24) & 0xFF))
        ch += 17 <= op <= 20
        n += 1; i += 1
    used.discard(0)
    return mx, ch, n, len(used)

for spec in sys.argv[1:]:
    name, dim, S = spec.split(":"); dim = int(dim); S = int(S)
    T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
    tape = m.Tape(m.model(name))
    ctx = m.Context(S)
    (ctx.render3D(tape, T) if dim == 3 else ctx.render2D(tape))
    pool = ctx.tape_data
    print("%s %dD %d: root tape %d clauses, %d slots, %d min/max" % (name, dim, S, tape.length - 2, tape.num_slots, tape.num_choices))
    for st in ([0, 1] if dim == 3 else [0]):
        t = ctx.stages[st].tiles
        t = t[t["next"] != -1]
        heads = np.unique(t["tape"])
        if heads.size > 3000: heads = heads[:: heads.size // 3000]
        r = np.array([walk(pool, int(h)) for h in heads])
        print("  tapes walked by the stage after %d (%d sampled): max slot index %d (mean %.0f), DISTINCT slots max %d (mean %.0f), min/max clauses max %d (mean %.0f), clauses max %d (mean %.0f)" % (
            st, heads.size, r[:, 0].max(), r[:, 0].mean(), r[:, 3].max(), r[:, 3].mean(), r[:, 1].max(), r[:, 1].mean(), r[:, 2].max(), r[:, 2].mean()))
    ctx.close()
