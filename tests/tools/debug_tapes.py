import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import mpr_amd as m
from oracle import orc
X, Y = m.Tree.X(), m.Tree.Y()
t = m.tmax(m.sqrt(X * X + Y * Y) - 1, 0.5 - m.sqrt(X * X + Y * Y))
tape = m.Tape(t)
print("root:", m.decode(tape.data))
S = 256
ctx = m.Context(S, flags=m.CTX_COUNTERS)
ctx.render2D(tape)
ref = orc.Frame(tape.data, 2, S, m.colmajor(np.eye(3), 3))
pool = ctx.tape_data
def walk(pool, head):
    out = []; p = head
    while True:
        p += 1; d = int(pool[p]); op = d & 255
        if op == 1:
            p += np.int32(np.uint32(d >> 32)); continue
        out.append(d)
        if op == 0: break
    return out
for s, nxt in ((0, 2), (2, 3)):
    g, r = ctx.stages[nxt].tiles, ref.tiles[nxt]
    g, r = g[g["position"] != -1], r[r["position"] != -1]
    g, r = g[np.argsort(g["position"])], r[np.argsort(r["position"])]
    bad = 0
    for a, b in zip(g, r):
        wa, wb = walk(pool, a["tape"]), walk(ref.pool, b["tape"])
        if wa != wb:
            bad += 1
            if bad <= 3:
                print("stage", s, "pos", a["position"], "gpu tape@", a["tape"], m.decode(wa), "\n   oracle", m.decode(wb))
    print("stage", s, "tiles", len(g), "mismatching", bad)
print(ctx.counters())
print("---- stage-0 parents (after the frame) ----")
g0, r0 = ctx.stages[0].tiles, ref.tiles[0]
for i in range(16):
    wa = walk(pool, g0[i]["tape"]) if g0[i]["tape"] else []
    wb = walk(ref.pool, r0[i]["tape"]) if r0[i]["tape"] else []
    print(i, "gpu", g0[i], [x[0] for x in m.decode(wa)][-4:], "| orc", r0[i], [x[0] for x in m.decode(wb)][-4:], "SAME" if wa == wb else "DIFF")
g2 = ctx.stages[2].tiles
print("stage2 list entries 0..15 tape fields:", g2["tape"][:16], "positions", g2["position"][:16])
print("---- chunk dumps for mismatching stage-2 survivors ----")
g, r = ctx.stages[3].tiles, ref.tiles[3]
g, r = g[np.argsort(g["position"])], r[np.argsort(r["position"])]
shown = 0
lanes = []
g2all = ctx.stages[2].tiles
for a, b in zip(g, r):
    wa, wb = walk(pool, a["tape"]), walk(ref.pool, b["tape"])
    if wa != wb:
        idx = int(np.flatnonzero(g2all["position"] == a["position"])[0])
        lanes.append(idx % 64)
        if shown < 2:
            shown += 1
            t0 = int(a["tape"])
            cb = (t0 - 9) // 64 * 64 + 9
            print("pos", a["position"], "list index", idx, "lane", idx % 64, "tape", t0, "chunk base", cb)
            for k in range(max(t0 - 2, cb), cb + 64):
                print("   ", k - cb, m.decode([pool[k]])[0])
print("mismatching lanes:", sorted(lanes))
ok_lanes = []
for a, b in zip(g, r):
    if walk(pool, a["tape"]) == walk(ref.pool, b["tape"]):
        idx = int(np.flatnonzero(g2all["position"] == a["position"])[0]); ok_lanes.append(idx % 64)
print("matching lanes:", sorted(set(ok_lanes)))
