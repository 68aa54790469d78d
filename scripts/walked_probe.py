"""Development aid (gpurun): tiles the float pass lists and walks, and its duration, under environment settings.
usage: walked_probe.py MODEL SIZE [ENV=VALUE[,ENV=VALUE..] ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpr_amd as mpr
name, S = sys.argv[1], int(sys.argv[2])
T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
tape = mpr.Tape(mpr.model(name))
for spec in [""] + sys.argv[3:]:
    env = dict(kv.split("=") for kv in spec.split(",") if kv)
    os.environ.update(env)
    os.environ["MPR_DEBUG_WALKED"] = "1"
    ctx = mpr.Context(S)
    for _ in range(6):
        ctx.render3D(tape, T)
    listed, walked = ctx.frame_tiles()[2], ctx.tiles_walked()
    ctx.close()
    del os.environ["MPR_DEBUG_WALKED"]
    tctx = mpr.Context(S, flags=mpr.CTX_TIMING)
    acc = {}
    for k in range(23):
        tctx.render3D(tape, T)
        if k >= 3:
            for n, ms in tctx.timings():
                acc[n] = acc.get(n, 0.0) + ms / 20
    kern = tctx.float_kernel()
    tctx.close()
    ctx = mpr.Context(S)
    for _ in range(20):
        ctx.render3D(tape, T)
    t0 = time.perf_counter()
    for _ in range(100):
        ctx.render3D(tape, T)
    ms = (time.perf_counter() - t0) * 10
    ctx.close()
    print("%s %d %-40s %s listed %d walked %d | frame %.4f ms | %s" % (name, S, spec, kern, listed, walked, ms, " ".join("%s %.1f" % (k, v * 1000) for k, v in acc.items())), flush=True)
    for k in env:
        del os.environ[k]
