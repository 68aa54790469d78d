"""Round 5 probe: stage forms, per-kernel times and frame times of the 2-D configurations by size (the stage-form cliffs)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpr_amd as m

def run(model, S, frames=40):
    tape = m.Tape(m.model(model))
    ctx = m.Context(S, flags=m.CTX_TIMING)
    for _ in range(5): ctx.render2D(tape)
    per = {}
    for _ in range(frames):
        ctx.render2D(tape)
        for k, v in ctx.timings(): per[k] = per.get(k, 0.0) + v / frames
    forms = ctx.tile_stage_forms(); fk = ctx.float_kernel(); tiles = ctx.frame_tiles()
    ctx.close()
    ctx = m.Context(S)
    for _ in range(10): ctx.render2D(tape)
    t0 = time.perf_counter()
    for _ in range(frames * 3): ctx.render2D(tape)
    ms = (time.perf_counter() - t0) / (frames * 3) * 1e3
    ctx.close()
    print(json.dumps({"model": model, "S": S, "ms": round(ms, 4), "forms": forms, "float": fk, "kernel_ms": {k: round(v, 4) for k, v in per.items()}, "tiles": tiles}), flush=True)

for model, sizes in (("prospero", (256, 512, 1024, 2048)), ("involute_gear_2d", (512, 1024, 2048, 4096)), ("hello_world", (256, 1024))):
    for S in sizes:
        run(model, S)
