"""Development aid (gpurun): first stages on the tape's loose forward walk (mpr_tape::big_fwd; MPR_TILE_GEN_BIG=0 switches it off,
MPR_TILE_GEN_BIG_TILES = the first-stage tile count from which it replaces the level-parallel kernel): frame times, forms, the share of
wavefronts that fell back on the interpreter, and both renderings of the frame compared (MPR_CTX_PARANOID)."""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpr_amd as m

T = np.eye(4, dtype=np.float32)
T[3, 2] = 0.3
cfgs = [a.split(":") for a in sys.argv[1:]] or [("architecture", "2048"), ("architecture", "1024"), ("architecture", "1536"), ("architecture", "512")]
for name, S in cfgs:
    S = int(S)
    tape = m.Tape(m.model(name))
    for env in ({"MPR_TILE_GEN_BIG": "0"}, {}, {"MPR_TILE_GEN_BIG_TILES": "1"}, {"MPR_TILE_GEN_BIG_TILES": "100000000"}):
        for k in ("MPR_TILE_GEN_BIG", "MPR_TILE_GEN_BIG_TILES"):
            os.environ.pop(k, None)
        os.environ.update(env)
        os.environ["MPR_DEBUG_REDO"] = "1"
        ctx = m.Context(S, flags=m.CTX_PARANOID)
        for _ in range(2):
            ctx.render3D(tape, T)
        st = ctx.paranoid_stats()
        ctx.close()
        ctx = m.Context(S)
        for _ in range(3):
            ctx.render3D(tape, T)
        out = (ctypes.c_uint32 * 2)()
        m.lib().mpr_debug_redo_counts.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        m.lib().mpr_debug_redo_counts(ctx._h, out)
        forms = ctx.tile_stage_forms()
        ctx.close()
        os.environ.pop("MPR_DEBUG_REDO")
        ctx = m.Context(S)
        for _ in range(30):
            ctx.render3D(tape, T)
        per = []
        for _ in range(100):
            t0 = time.perf_counter(); ctx.render3D(tape, T); per.append((time.perf_counter() - t0) * 1e3)
        print(name, S, env or "default", "median %.4f ms" % np.median(per), forms, "| walks %d, fell back %d | paranoid %s" % (out[0], out[1], st), flush=True)
        ctx.close()
