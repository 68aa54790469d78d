import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, mpr_amd as m
tape = m.Tape(m.model("architecture"))
T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
ctx = m.Context(2048)
per = []
for k in range(140):
    t = time.perf_counter(); ctx.render3D(tape, T); per.append((time.perf_counter() - t) * 1e3)
print([(k, round(p, 1)) for k, p in enumerate(per) if p > 3.0], ctx.tile_stage_forms())
