"""2000 consecutive blocking frames of every BASELINE configuration (round 5): per-frame wall time with the
context's pool capacity, pool growths, restarted frames and vetoes (mpr_debug_frame_stats) — every frame slower than 1.5 x the median
is printed with what changed in it.  usage: outlier_probe.py [FRAMES]"""
import ctypes, gc, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpr_amd as m

gc.disable()          # (a 0.2 ms frame feels the collector)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
f = m.lib().mpr_debug_frame_stats
f.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3


def stats(ctx):
    out = (ctypes.c_int64 * 4)()
    f(ctx._h, out)
    return list(out)


for model, dim, S in (("prospero", 2, 1024), ("involute_gear_2d", 2, 4096), ("bear", 3, 1024), ("architecture", 3, 2048), ("architecture", 3, 1024), ("bear", 3, 2048)):
    tape = m.Tape(m.model(model))
    ctx = m.Context(S)
    render = (lambda: ctx.render3D(tape, T)) if dim == 3 else (lambda: ctx.render2D(tape))
    per, st = [], []
    for k in range(N):
        t0 = time.perf_counter()
        render()
        per.append((time.perf_counter() - t0) * 1e3)
        st.append(stats(ctx))
    per = np.array(per)
    warm = per[20:]
    med = float(np.median(warm))
    slow = [k for k in range(N) if per[k] > 1.5 * med]
    rec = {"workload": "%s render%dD %d" % (model, dim, S), "frames": N, "median_ms": round(med, 4), "mean_after_20": round(float(warm.mean()), 4),
           "max_after_20_ms": round(float(warm.max()), 3), "max_over_median_after_20": round(float(warm.max()) / med, 3),
           "pool_clauses_first_last": [st[0][0], st[-1][0]], "pool_growths": st[-1][1], "frames_restarted": st[-1][2], "skip0_vetoes": st[-1][3],
           "slow_frames": [{"frame": k, "ms": round(float(per[k]), 3), "pool": st[k][0], "growths": st[k][1] - (st[k - 1][1] if k else 0),
                            "restarts": st[k][2] - (st[k - 1][2] if k else 0), "vetoes": st[k][3] - (st[k - 1][3] if k else 0)} for k in slow[:12]]}
    print(json.dumps(rec), flush=True)
    ctx.close()
