"""Development aid: first tile whose shortened tape differs between the level-parallel later stages and the oracle."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, mpr_amd as m
from oracle import orc
name, dim, S = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
tape = m.Tape(m.model(name))
T = np.eye(dim + 1, dtype=np.float32)
if dim == 3: T[3, 2] = 0.3
ctx = m.Context(S, flags=m.CTX_COUNTERS)
ref = orc.Frame(tape.data, dim, S, m.colmajor(T, dim + 1), threads=0)
(ctx.render2D(tape, T) if dim == 2 else ctx.render3D(tape, T))
pool, rpool = ctx.tape_data, ref.pool
def walk(pool, head):
    out = []; p = head
    while True:
        p += 1
        w = int(pool[p]); op = w & 0xff
        if op == 1:
            p += int(np.array([(w >> 32) & 0xffffffff], dtype=np.uint32).view(np.int32)[0]); continue
        if op == 0: break
        out.append(w)
    return out
stages = [0, 1, 2, 3] if dim == 3 else [0, 2, 3]
for k in range(len(stages) - 1):
    nxt = stages[k + 1]
    g, r = ctx.stages[nxt].tiles, ref.tiles[nxt]
    g, r = g[g["position"] != -1], r[r["position"] != -1]
    g, r = g[np.argsort(g["position"])], r[np.argsort(r["position"])]
    glen, gh = orc.tiles_digest(pool, g); rlen, rh = orc.tiles_digest(rpool, r)
    bad = np.flatnonzero((glen != rlen) | (gh != rh))
    print("list %d: %d survivors, %d differ" % (nxt, g.size, bad.size))
    if bad.size:
        b = bad[0]
        print(" tile position %d: gpu tape %d (len %d), oracle tape %d (len %d)" % (g["position"][b], g["tape"][b], glen[b], r["tape"][b], rlen[b]))
        gw, rw = walk(pool, int(g["tape"][b])), walk(rpool, int(r["tape"][b]))
        import difflib
        def show(w):
            return "op%d o%d l%d r%d%s" % (w & 0xff, (w >> 8) & 0xff, (w >> 16) & 0xff, (w >> 24) & 0xff, " imm" if w >> 32 else "")
        sm = difflib.SequenceMatcher(None, gw, rw)
        missing = set()
        for tag, i1, i2, j1, j2 in sm.get_opcodes():
            if tag != "equal":
                print("  ", tag, [show(x) for x in gw[i1:i2]][:8], "->", [show(x) for x in rw[j1:j2]][:8])
                if tag == "insert": missing.update(range(j1, j2))
        # oracle tape in evaluation order (the walk above goes from the head: that IS evaluation order): who reads a missing clause's result?
        for j in sorted(missing)[:6]:
            o = (rw[j] >> 8) & 0xff
            for k in range(j + 1, len(rw)):
                w = rw[k]
                if ((w >> 16) & 0xff) == o or ((w >> 24) & 0xff) == o:
                    print("   oracle clause %d (%s) is read by clause %d (%s)%s" % (j, show(rw[j]), k, show(w), " [also missing]" if k in missing else " [present on the GPU]"))
                    break
                if ((w >> 8) & 0xff) == o:
                    print("   oracle clause %d (%s): slot overwritten by clause %d (%s) before any read" % (j, show(rw[j]), k, show(w)))
                    break
        break
