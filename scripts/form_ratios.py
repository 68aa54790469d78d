"""Development aid (gpurun): the last stage's sample — clauses of the tapes handed on against clauses of the tapes walked by the
group form — frame after frame (MPR_DEBUG_CHOICES prints it and keeps every frame measuring)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MPR_DEBUG_CHOICES"] = "1"
import mpr_amd as mpr
T = np.eye(4, dtype=np.float32)
T[3, 2] = 0.3

for name, S in (("architecture", 512), ("architecture", 1024), ("architecture", 2048), ("involute_gear_3d", 1024), ("hello_world", 1024), ("bear", 1024)):
    tape = mpr.Tape(mpr.model(name))
    ctx = mpr.Context(S)
    print(name, S, flush=True)
    for _ in range(6):
        ctx.render3D(tape, T)
    ctx.close()
