"""Per-kernel times (HIP events: CTX_TIMING), stage forms, the share of wavefronts whose loose walk asked for the exact one
(MPR_DEBUG_REDO) and the frame time without any of that, for a few 3-D models / sizes.  One JSON line per configuration.
usage: stage_probe.py [MODEL:SIZE ...]"""
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(model, S, frames):
    os.environ["MPR_DEBUG_REDO"] = "1"
    import mpr_amd as m
    T = np.eye(4, dtype=np.float32)
    T[3, 2] = 0.3
    tape = m.Tape(m.model(model))
    ctx = m.Context(S, flags=m.CTX_TIMING)
    for _ in range(5):
        ctx.render3D(tape, T)
    per = {}
    for _ in range(frames):
        ctx.render3D(tape, T)
        for k, v in ctx.timings():
            per[k] = per.get(k, 0.0) + v / frames
    forms = ctx.tile_stage_forms()
    out = (ctypes.c_uint32 * 2)()
    m.lib().mpr_debug_redo_counts.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    m.lib().mpr_debug_redo_counts(ctx._h, out)
    ctx.close()
    os.environ.pop("MPR_DEBUG_REDO", None)          # (its counter is one word every wavefront adds to: 9 ns each, one after the other)
    ctx = m.Context(S)
    for _ in range(10):
        ctx.render3D(tape, T)
    t0 = time.perf_counter()
    for _ in range(frames):
        ctx.render3D(tape, T)
    ms = (time.perf_counter() - t0) / frames * 1e3
    tiles = ctx.frame_tiles()
    ctx.close()
    print(json.dumps({"model": model, "S": S, "ms_per_frame": round(ms, 4), "forms": forms,
                      "kernel_ms": {k: round(v, 4) for k, v in per.items()}, "gen_waves": int(out[0]), "redone": int(out[1]),
                      "tiles": tiles}), flush=True)


if __name__ == "__main__":
    cfgs = [("bear", 1024), ("bear", 512), ("bear", 2048)]
    if len(sys.argv) > 1:
        cfgs = [(a.split(":")[0], int(a.split(":")[1])) for a in sys.argv[1:]]
    for model, S in cfgs:
        run(model, S, 30 if S <= 1024 else 8)
