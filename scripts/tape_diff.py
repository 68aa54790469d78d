"""Development aid: the tapes the 16^3 tiles carry after a reference-mode frame, generated code against the interpreter."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mpr_amd as m
OPN = {v: k for k, v in m.OP.items()}
def walk(pool, head):
    out = []; i = head + 1
    while True:
        w = int(pool[i]); op = w & 0xff
        if op == 0: out.append(w); break
        if op == 1:
            d = (w >> 32); d = d - (1 << 32) if d >= (1 << 31) else d
            i = i + d + 1; continue
        out.append(w); i += 1
    return out
def frame(S, env):
    for k in ("MPR_TILE_GEN_LAST", "MPR_TILE_GEN"): os.environ.pop(k, None)
    os.environ.update(env)
    tape = m.Tape(m.model("bear")); T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
    ctx = m.Context(S, flags=m.CTX_COUNTERS)
    ctx.render3D(tape, T)
    tiles = np.array(ctx.stages[1].tiles); pool = np.array(ctx.tape_data)
    ctx.close()
    live = tiles[tiles["position"] != -1]
    return {int(t["position"]): walk(pool, int(t["tape"])) for t in live}
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
a = frame(S, {}); b = frame(S, {"MPR_TILE_GEN_LAST": "0"})
print(len(a), len(b), "tiles")
shown = 0
for pos in sorted(a):
    if a[pos] != b.get(pos):
        x, y = a[pos], b[pos]
        print("tile", pos, "gen", len(x), "interp", len(y))
        i = 0
        while i < min(len(x), len(y)) and x[i] == y[i]: i += 1
        for j in range(max(0, i - 2), i + 4):
            f = lambda t: "%s o%d l%d r%d %08x" % (OPN.get(t & 0xff, t & 0xff), (t >> 8) & 0xff, (t >> 16) & 0xff, (t >> 24) & 0xff, t >> 32)
            print("   %3d  gen %-34s interp %-34s" % (j, f(x[j]) if j < len(x) else "-", f(y[j]) if j < len(y) else "-"))
        shown += 1
        if shown >= 3: break
print("differing tiles:", sum(1 for p in a if a[p] != b.get(p)))
