"""Is a sporadic slow frame the host's or the GPU's?  bear render3D 1024^3, N blocking frames, twice: as an ordinary process and under
SCHED_FIFO (nothing preempts the thread that spins on the survivor counts), each frame timed on the host (perf_counter) and on the
device (HIP events on the context's stream around the frame).  usage: outlier_host_or_gpu.py [FRAMES]"""
import gc, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mpr_amd as m

gc.disable()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
tape = m.Tape(m.model("bear"))
for mode in ("ordinary", "SCHED_FIFO"):
    if mode == "SCHED_FIFO":
        try:
            os.sched_setscheduler(0, os.SCHED_FIFO, os.sched_param(50))
        except Exception as e:
            print(json.dumps({"mode": mode, "error": str(e)}))
            continue
    ctx = m.Context(1024)
    stream = torch.cuda.ExternalStream(ctx.stream)
    for _ in range(30):
        ctx.render3D(tape, T)
    host, dev = [], []
    for k in range(N):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(stream)
        ctx.render3D(tape, T)
        e1.record(stream)
        host.append((time.perf_counter() - t0) * 1e3)
        e1.synchronize()
        dev.append(e0.elapsed_time(e1))
    host, dev = np.array(host), np.array(dev)
    slow = np.flatnonzero(host > 1.5 * np.median(host))
    print(json.dumps({"mode": mode, "frames": N, "host_median_ms": round(float(np.median(host)), 4), "host_max_ms": round(float(host.max()), 3),
                      "device_median_ms": round(float(np.median(dev)), 4), "device_max_ms": round(float(dev.max()), 3),
                      "slow_frames": [{"frame": int(k), "host_ms": round(float(host[k]), 3), "device_ms": round(float(dev[k]), 3)} for k in slow[:10]]}), flush=True)
    ctx.close()
