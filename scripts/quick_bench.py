"""Quick per-kernel timing of one model/size (development aid; not the judged bench)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mpr_amd as m

def run(model, dim, S, n=10):
    tape = m.Tape(m.model(model))
    ctx = m.Context(S, flags=m.CTX_TIMING)
    T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
    f = (lambda: ctx.render3D(tape, T)) if dim == 3 else (lambda: ctx.render2D(tape))
    for _ in range(3): f()
    acc = {}; t0 = time.perf_counter()
    for _ in range(n):
        f()
        for k, v in ctx.timings(): acc[k] = acc.get(k, 0) + v / n
    dt = (time.perf_counter() - t0) / n * 1e3
    print("%s %dD %d: %.3f ms/frame  " % (model, dim, S, dt) + " ".join("%s=%.3f" % kv for kv in acc.items()), flush=True)
    if os.environ.get("MPR_QB_STAGES") == "1":       # every launch of the last frame, and the frame without events
        print("   launches: " + " ".join("%s=%.3f" % kv for kv in ctx.timings()), flush=True)
        ctx.close()
        ctx = m.Context(S)
        f = (lambda: ctx.render3D(tape, T)) if dim == 3 else (lambda: ctx.render2D(tape))
        for _ in range(5): f()
        t0 = time.perf_counter()
        for _ in range(50): f()
        print("   without events: %.3f ms/frame" % ((time.perf_counter() - t0) / 50 * 1e3), flush=True)
    ctx.close()

if __name__ == "__main__":
    for spec in sys.argv[1:]:
        model, dim, S = spec.split(":")
        run(model, int(dim), int(S))
