"""Development aid (gpurun): wavefronts of the tile stages' generated walks that ran loose code and those of them redone on the exact
code (MPR_DEBUG_REDO=1), per frame.  usage: redo_probe.py MODEL SIZE"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mpr_amd as mpr
name, S = sys.argv[1], int(sys.argv[2])
T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
tape = mpr.Tape(mpr.model(name))
os.environ["MPR_DEBUG_REDO"] = "1"
ctx = mpr.Context(S)
import ctypes
def redo_counts(ctx):
    out = (ctypes.c_uint32 * 2)()
    mpr.lib().mpr_debug_redo_counts.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    mpr.lib().mpr_debug_redo_counts(ctx._h, out)
    return int(out[0]), int(out[1])
prev = (0, 0)
for k in range(5):
    ctx.render3D(tape, T)
    now = redo_counts(ctx)
    print("frame %d: %s | %d wavefronts on loose code, %d of them redone | tiles %s" % (k, ctx.tile_stage_forms(), now[0] - prev[0], now[1] - prev[1], ctx.frame_tiles()), flush=True)
    prev = now
