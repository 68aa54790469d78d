"""Pipelined frames against frames with MPR_PIPELINE=0 (development aid): same images, timing of both."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mpr_amd as m

T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
for spec in sys.argv[1:]:
    model, S = spec.split(":"); S = int(S)
    tape = m.Tape(m.model(model))
    ref = m.Context(S)
    os.environ["MPR_PIPELINE"] = "1"
    ctx = m.Context(S, flags=m.CTX_TIMING)
    for _ in range(3):
        ref.render3D(tape, T)
    bad = 0
    for k in range(12):
        ctx.render3D(tape, T)
        same = np.array_equal(ctx.image, ref.image) and np.array_equal(ctx.normals, ref.normals)
        bad += not same
        if k in (0, 2, 11):
            print(model, S, "frame", k, ctx.float_kernel(), "same" if same else "DIFFERENT h=%d n=%d" % (int((ctx.image != ref.image).sum()), int((ctx.normals != ref.normals).sum())),
                  " ".join("%s=%.3f" % kv for kv in ctx.timings()), flush=True)
    ctx.close()
    ctx = m.Context(S)
    del os.environ["MPR_PIPELINE"]
    for c, name in ((ctx, "pipelined (MPR_PIPELINE=1)"), (ref, "default")):
        for _ in range(10): c.render3D(tape, T)
        t0 = time.perf_counter()
        for _ in range(50): c.render3D(tape, T)
        print("  %s: %.3f ms/frame (%s)" % (name, (time.perf_counter() - t0) / 50 * 1e3, c.float_kernel()), flush=True)
    print("  frames that differed:", bad)
    ctx.close(); ref.close()
