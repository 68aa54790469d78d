"""Development aid: frames with the first tile stage on generated code (MPR_TILE_GEN=1: both walks, 2: forward only) against the
interpreter's (0): heights, normals and the stages' tile lists.   python scripts/gen_check.py bear:3:256 ..."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mpr_amd as m

def frame(name, dim, S, mode, flags=0):
    os.environ["MPR_TILE_GEN"] = str(mode)
    tape = m.Tape(m.model(name)); T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
    ctx = m.Context(S, flags=flags)
    f = (lambda: ctx.render3D(tape, T)) if dim == 3 else (lambda: ctx.render2D(tape))
    f(); f()
    out = {"image": np.array(ctx.image), "normals": np.array(ctx.normals) if dim == 3 else None}
    t0 = time.perf_counter()
    for _ in range(20): f()
    out["ms"] = (time.perf_counter() - t0) / 20 * 1e3
    ctx.close()
    # a reference-mode frame: tile lists of every stage
    os.environ["MPR_SKIP_STAGE0"] = "0"
    ctx = m.Context(S, flags=m.CTX_COUNTERS | m.CTX_SERIAL_STAGES)
    f = (lambda: ctx.render3D(tape, T)) if dim == 3 else (lambda: ctx.render2D(tape))
    f()
    out["counters"] = ctx.counters()
    out["image_ref"] = np.array(ctx.image)
    ctx.close()
    del os.environ["MPR_SKIP_STAGE0"]
    return out

for spec in sys.argv[1:]:
    name, dim, S = spec.split(":"); dim = int(dim); S = int(S)
    base = frame(name, dim, S, 0)
    for mode in (2, 1):
        got = frame(name, dim, S, mode)
        ok = np.array_equal(got["image"], base["image"]) and np.array_equal(got["image_ref"], base["image_ref"])
        if dim == 3: ok = ok and np.array_equal(got["normals"], base["normals"])
        cd = {k: (base["counters"][k], got["counters"][k]) for k in base["counters"] if base["counters"][k] != got["counters"][k]}
        print("%s %dD %d mode %d: %s  %.3f ms (interpreter %.3f)  counters that differ: %s" % (name, dim, S, mode, "SAME" if ok else "DIFFERENT", got["ms"], base["ms"], cd), flush=True)
