"""Render a few frames of one model (for rocprofv3 runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mpr_amd as m
model, dim, S, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
tape = m.Tape(m.model(model))
ctx = m.Context(S)
T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
for _ in range(n):
    if dim == 3: ctx.render3D(tape, T)
    else: ctx.render2D(tape)
print("done", ctx.counters()["voxel_tiles"])
