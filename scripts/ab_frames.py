"""Median frame time of a few 3-D models with the float pass's form forced (MPR_VOXEL_GROUPS=2: always the group form, 0: never)
and as the context chooses (the last stage's sample of tape lengths: context.hip, `pays`).  Development aid."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mpr_amd as m
T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
MODELS = (("architecture", 3, 512), ("architecture", 3, 1024), ("architecture", 3, 2048), ("involute_gear_3d", 3, 1024), ("hello_world", 3, 1024), ("bear", 3, 1024),
          ("prospero", 2, 1024), ("involute_gear_2d", 2, 4096), ("prospero", 2, 256), ("involute_gear_2d", 2, 1024))
for name, dim, S in MODELS:
    tape = m.Tape(m.model(name))
    for groups in ("", "2", "0"):
        if groups: os.environ["MPR_VOXEL_GROUPS"] = groups
        else: os.environ.pop("MPR_VOXEL_GROUPS", None)
        ctx = m.Context(S)
        render = (lambda: ctx.render3D(tape, T)) if dim == 3 else (lambda: ctx.render2D(tape))
        for _ in range(40): render()
        per, kinds = [], {}
        for _ in range(200):
            t0 = time.perf_counter(); render(); per.append((time.perf_counter() - t0) * 1e3)
            k = ctx.float_kernel()
            kinds[k] = kinds.get(k, 0) + 1
        print(name, S, "MPR_VOXEL_GROUPS=%s" % (groups or "-"), "median %.4f" % np.median(per), kinds, flush=True)
        ctx.close()
