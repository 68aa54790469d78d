"""How much of a tile's shortened tape can the ROOT tape's code skip with scalar branches?

Test / measurement infrastructure (uses the CPU oracle): for every smallest tile of an oracle frame, compare
 - the length of the tile's own shortened tape (what the reference's float pass walks),
 - the clauses the root tape's straight-line code would run when every (min / max clause, side) whose operand's exclusive
   sub-DAG is a run of >= MINRUN consecutive clauses is guarded by one scalar branch on the tile's decision bits.
usage: skip_study.py [model] [size] [minrun]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mpr_amd as m
from oracle import orc

name = sys.argv[1] if len(sys.argv) > 1 else "bear"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
MINRUN = int(sys.argv[3]) if len(sys.argv) > 3 else 4
tape = m.Tape(m.model(name))
root = tape.data
dec = m.decode(root)
n = len(dec)
MM = {"MIN_LHS_RHS", "MAX_LHS_RHS", "MIN_LHS_IMM", "MAX_LHS_IMM"}
UN = {"SQUARE_LHS", "SQRT_LHS", "NEG_LHS", "SIN_LHS", "COS_LHS", "ASIN_LHS", "ACOS_LHS", "ATAN_LHS", "EXP_LHS", "ABS_LHS", "LOG_LHS", "COPY_LHS"}
def uses(op):
    l = op.endswith("LHS") or "LHS_" in op
    r = op.endswith("RHS") or "_RHS" in op
    if op in UN: return True, False
    return l, r
# SSA: defs
cur = {}
h = dec[0]
for ax, s in zip("xyz", h[1:4]): cur[s] = -1 - "xyz".index(ax)
ldef, rdef = [None] * n, [None] * n
for i in range(1, n - 1):
    op, o, l, r, imm = dec[i]
    ul, ur = uses(op)
    if ul: ldef[i] = cur[l]
    if ur: rdef[i] = cur[r]
    cur[o] = i
rootdef = cur[dec[n - 1][1]]
body = list(range(1, n - 1))
def infer(decs, present):
    """complete the decisions: a LIVE min / max clause that is absent from the tile's tape was decided for the operand that
    already sits in its out slot (the reference drops such a clause instead of writing a COPY onto itself)"""
    d = dict(decs)
    while True:
        lv = live_set(decisions=d)
        add = {}
        for c in lv:
            if dec[c][0] in MM and c not in present and c not in d:
                o, l, r = dec[c][1:4]
                add[c] = 'l' if l == o else 'r'
        if not add: return d, lv
        d.update(add)
def live_set(dead_edge=None, decisions=None):
    """clauses reachable from the result; dead_edge=(m, 'l'|'r') removed; decisions: {m: 'l'|'r'} keeps only that side"""
    seen = set()
    st = [rootdef]
    while st:
        c = st.pop()
        if c is None or c < 0 or c in seen: continue
        seen.add(c)
        sides = []
        if ldef[c] is not None: sides.append(('l', ldef[c]))
        if rdef[c] is not None: sides.append(('r', rdef[c]))
        for sd, d in sides:
            if dead_edge == (c, sd): continue
            if decisions and c in decisions and decisions[c] != sd: continue
            st.append(d)
    return seen
full = live_set()
mm = [i for i in body if dec[i][0] in MM]
guards = []   # (start, end_exclusive, m, side_removed)
for c in mm:
    for sd in "lr":
        if (sd == 'l' and ldef[c] is None) or (sd == 'r' and rdef[c] is None): continue
        dead = sorted(full - live_set((c, sd)))
        # runs
        k = 0
        while k < len(dead):
            j = k
            while j + 1 < len(dead) and dead[j + 1] == dead[j] + 1: j += 1
            if j - k + 1 >= MINRUN: guards.append((dead[k], dead[j] + 1, c, sd))
            k = j + 1
print("%s: %d clauses, %d min/max, %d guarded runs (>= %d), total guarded clauses %d" % (name, n - 2, len(mm), len(guards), MINRUN, sum(e - s for s, e, _, _ in guards)))
for g in sorted(guards): print("   run [%d, %d) len %d  dead when clause %d (%s) drops its %s" % (g[0], g[1], g[1] - g[0], g[2], dec[g[2]][0], g[3]))

T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
ref = orc.Frame(root, 3, S, m.colmajor(T, 4), threads=0)
tiles = ref.tiles[3]
tiles = tiles[tiles["position"] != -1]
pool = ref.pool
def walk(t):
    out = []
    i = t + 1
    while True:
        w = int(pool[i]); op = w & 0xFF
        if op == 0: return out
        if op == 1:
            d = (w >> 32) & 0xFFFFFFFF
            if d >= 1 << 31: d -= 1 << 32
            i = i + d + 1
            continue
        out.append(w); i += 1
rootw = [int(x) for x in root]
key = lambda w: w >> 8
def match(words):
    """-> decisions {m: side}, present set"""
    present = set(); decs = {}
    k = 1
    for w in words:
        op = w & 0xFF
        while True:
            rw = rootw[k]
            if rw == w: break
            rop = m.OP_NAMES[rw & 0xFF]
            if rop in MM and key(rw) == key(w) and m.OP_NAMES[op] in ("COPY_LHS", "COPY_RHS", "COPY_IMM"): break
            k += 1
        present.add(k)
        if (rootw[k] & 0xFF) != op:
            decs[k] = 'l' if m.OP_NAMES[op] == "COPY_LHS" else 'r'
        k += 1
    return decs, present
cache = {}
tot_own = tot_run = tot_live = 0
patterns = {}
for t in np.unique(tiles["tape"]):
    words = walk(int(t))
    decs, present = match(words)
    decs, lvs = infer(decs, present)
    cnt = int((tiles["tape"] == t).sum())
    # decisions known to the code: the min/max clauses turned into copies.  min/max clauses that are ABSENT are dead anyway.
    skipped = set()
    for s, e, c, sd in guards:
        # the run is dead when clause c does not use side sd: decided for the other side, or c itself absent (dead)
        if c in decs and decs[c] != sd and c in lvs:
            skipped.update(range(s, e))
    run = (n - 2) - len(skipped)
    lv = len(lvs)
    tot_own += len(words) * cnt; tot_run += run * cnt; tot_live += lv * cnt
    patterns[frozenset(decs.items())] = patterns.get(frozenset(decs.items()), 0) + cnt
N = tiles.size
print("%d smallest tiles, %d distinct tapes, %d distinct decision patterns" % (N, np.unique(tiles['tape']).size, len(patterns)))
print("mean clauses: own tape %.1f | true liveness under its decisions %.1f | root code with guarded runs %.1f | root %d" % (tot_own / N, tot_live / N, tot_run / N, n - 2))
if os.environ.get("SKIP_DEBUG"):
    t = int(np.unique(tiles["tape"])[len(np.unique(tiles["tape"])) // 2])
    words = walk(t); decs, present = match(words)
    lv = live_set(decisions=decs)
    print("tile tape", t, "own", len(words), "live", len(lv), "decs", decs)
    print("live but absent:", sorted(lv - present)[:60])
    print("present but not live:", sorted(present - lv)[:60])
    print("mm present:", [(c, dec[c][0], decs.get(c)) for c in mm if c in present], "absent mm:", [c for c in mm if c not in present])
