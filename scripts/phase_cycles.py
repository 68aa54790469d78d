import sys, os; sys.path.insert(0, "/root/repo")
import numpy as np, mpr_amd as m
name, S = sys.argv[1], int(sys.argv[2])
tape = m.Tape(m.model(name)); T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
ctx = m.Context(S, flags=m.CTX_COUNTERS)
ctx.render3D(tape, T); ctx.counters()
