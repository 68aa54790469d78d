import sys, time; sys.path.insert(0, ".")
import numpy as np, mpr_amd as m
tape = m.Tape(m.model("bear")); T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
ctx = m.Context(1024)
for _ in range(5): ctx.render3D(tape, T); a = ctx.image; b = ctx.normals
t0 = time.perf_counter()
for _ in range(50): ctx.render3D(tape, T)
t1 = time.perf_counter()
for _ in range(50): ctx.render3D(tape, T); a = ctx.image; b = ctx.normals
t2 = time.perf_counter()
print("render3D only %.3f ms/frame; with heightmap + normals copied to the host (8 MiB) %.3f ms/frame" % ((t1 - t0) / 50 * 1e3, (t2 - t1) / 50 * 1e3))
