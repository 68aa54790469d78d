"""Development aid (gpurun): tests/test_gpu_fuzz_shapes.py's random shapes in contexts that render every frame twice — the way a caller
gets it (a start at the 16^3 tiles, loose enclosures, generated code, no tapes from the last stage) and the reference's way — and compare
heights and normals on the device (MPR_CTX_PARANOID).  No oracle on the host: about a hundred frames a second, so the sweep can be two
orders of magnitude wider than scripts/fuzz_sweep.py (which holds the same shapes against the oracle; the reference's way on the chip is
what the test suite holds against the oracle, shape by shape).
usage: paranoid_sweep.py FIRST COUNT"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
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
T = np.eye(4, dtype=np.float32)
T[3, 2] = 0.3
T2 = np.eye(3, dtype=np.float32)
ctxs = {}


def context(S):
    if S not in ctxs:
        ctxs[S] = mpr.Context(S, flags=mpr.CTX_PARANOID)
    return ctxs[S]


t0 = time.time()
frames = bad = 0
for seed in range(first, first + count):
    for size in (3, 8, 16, 40):
        tape = ns["fuzz_tape"](mpr, seed, size)
        rng = np.random.default_rng(seed * 7 + size)
        S = int(rng.choice([128, 256]))
        view = T if rng.random() < 0.5 else random_view3(rng)
        ctx = context(S)
        before = ctx.paranoid_stats()
        for k in range(3):
            ctx.render3D(tape, view)
        ctx2 = context(256)
        z = float(np.float32(rng.uniform(-0.3, 0.3)))
        b2 = ctx2.paranoid_stats() if ctx2 is not ctx else None
        for k in range(2):
            ctx2.render2D(tape, T2, z)
        after = ctx.paranoid_stats()
        frames += 5
        if after[2] != before[2] or (b2 is not None and ctx2.paranoid_stats()[2] != b2[2]):
            bad += 1
            print("DIFFERS seed %d size %d S %d: 3-D/2-D cells %s -> %s (%s)" % (seed, size, S, before, after, ctx.tile_stage_forms()), flush=True)
    if (seed - first) % 200 == 199:
        print("... through seed %d: %d shapes differ so far, %.0f s" % (seed, bad, time.time() - t0), flush=True)
tot = [0, 0, 0]
for c in ctxs.values():
    st = c.paranoid_stats()
    tot = [a + b for a, b in zip(tot, st)]
print("seeds %d..%d x 4 sizes: %d frames rendered, %d of them a second time the reference's way, %d cells differ (%d shapes); %.0f s"
      % (first, first + count - 1, tot[0], tot[1], tot[2], bad, time.time() - t0))
