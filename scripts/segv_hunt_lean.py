"""VERDICT r5 next-2, the lean leg: scripts/fuzz_sweep.py's loop WITHOUT the oracle (tape, context, three 3-D frames, close, context, two
2-D frames, close) — the crash of round 5's sweep was in the tape's creation (faulthandler, scripts/segv_hunt.sh: mpr_amd/__init__.py
Tape.__init__ <- fuzz_tape), i.e. a heap that something before it had damaged.  usage: segv_hunt_lean.py FIRST COUNT [ROUNDS]"""
import faulthandler, os, sys
faulthandler.enable()
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mpr_amd as mpr
src = open(os.path.join(ROOT, "tests", "test_gpu_fuzz_shapes.py")).read().split("@pytest.mark.parametrize")[0]
src = src.replace("from conftest import view2, view3", "").replace("from helpers import check_default_path, compare_frame, compare_reader_frame", "")
ns = {}
exec(src, ns)


def random_view3(rng):
    V = np.eye(4, dtype=np.float32)
    V[:3, :3] += rng.uniform(-0.25, 0.25, (3, 3)).astype(np.float32)
    if rng.random() < 0.5:
        V[0] *= np.float32(-1.0)
    V[:3, 3] = rng.uniform(-0.15, 0.15, 3).astype(np.float32)
    V[3, :3] = rng.uniform(-0.3, 0.3, 3).astype(np.float32)
    return V


first, count = int(sys.argv[1]), int(sys.argv[2])
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 1
# HUNT_CHURN=1: what the oracle did to the process besides computing — large host buffers allocated, written and freed every iteration
# (its frame, pool and images copied out), and an OpenMP team of 16 that has run (torch's CPU kernels use libgomp as the oracle does)
CHURN = os.environ.get("HUNT_CHURN") == "1"
if CHURN:
    import torch
    torch.set_num_threads(16)
T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
T2 = np.eye(3, dtype=np.float32)
n = 0
for r in range(rounds):
    for seed in range(first, first + count):
        for size in (3, 8, 16):
            tape = ns["fuzz_tape"](mpr, seed, size)
            rng = np.random.default_rng(seed * 7 + size)
            S = int(rng.choice([128, 256]))
            view = T if rng.random() < 0.6 else random_view3(rng)
            if CHURN:
                bufs = [np.full(int(rng.integers(1, 40)) << 20, 7, dtype=np.uint8) for _ in range(4)]
                a = torch.rand(256, 256)
                (a @ a).sum().item()
                del bufs
            ctx = mpr.Context(S)
            for k in range(3):
                ctx.render3D(tape, view)
            ctx.close()
            if CHURN:
                bufs = [np.full(int(rng.integers(1, 40)) << 20, 9, dtype=np.uint8) for _ in range(2)]
                del bufs
            ctx = mpr.Context(256)
            for k in range(2):
                ctx.render2D(tape, T2, 0.1)
            ctx.close()
            n += 1
    print("round %d done: %d shapes" % (r, n), flush=True)
print("seeds %d..%d x 3 sizes x %d rounds: no crash" % (first, first + count - 1, rounds))
