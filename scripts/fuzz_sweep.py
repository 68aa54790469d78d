"""Development aid (gpurun): a wider one-off sweep of tests/test_gpu_fuzz_shapes.py's random shapes than the suite keeps —
default frames (3 in a row) in 3-D and 2-D against the oracle; prints the seeds that differ (as it goes: about two seeds a second,\nmost of it the oracle on the host).  usage: fuzz_sweep.py FIRST COUNT [ORACLE_THREADS]   (scripts/fuzz_sweep_parallel.sh runs a dozen of these side by side)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mpr_amd as mpr
from oracle import orc

orc.lib()
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
THREADS = int(sys.argv[3]) if len(sys.argv) > 3 else 0
T = np.eye(4, dtype=np.float32)
T[3, 2] = 0.3
T2 = np.eye(3, dtype=np.float32)
bad = 0
vetoes = 0
for seed in range(first, first + count):
    for size in (3, 8, 16):
        tape = ns["fuzz_tape"](mpr, seed, size)
        rng = np.random.default_rng(seed * 7 + size)
        S = int(rng.choice([128, 256]))
        view = T if rng.random() < 0.6 else random_view3(rng)
        ref = orc.Frame(tape.data, 3, S, mpr.colmajor(view, 4), threads=THREADS)
        ctx = mpr.Context(S)
        for k in range(3):
            ctx.render3D(tape, view)
            dh, dn = int((ctx.image != ref.filled[3]).sum()), int((ctx.normals != ref.normals).sum())
            if dh or dn:
                bad += 1
                print("DIFFERS 3-D seed %d size %d S %d frame %d: heights %d normals %d (%s)" % (seed, size, S, k, dh, dn, ctx.tile_stage_forms()), flush=True)
                break
        vetoes += ctx.skip0_vetoes()
        ctx.close()
        ref2 = orc.Frame(tape.data, 2, 256, mpr.colmajor(T2, 3), z=0.1, threads=THREADS)
        ctx = mpr.Context(256)
        for k in range(2):
            ctx.render2D(tape, T2, 0.1)
            d2 = int((ctx.image != ref2.filled[3]).sum())
            if d2:
                bad += 1
                print("DIFFERS 2-D seed %d size %d frame %d: %d (%s)" % (seed, size, k, d2, ctx.tile_stage_forms()), flush=True)
                break
        ctx.close()
    if (seed - first) % 10 == 9:
        print("... through seed %d: %d frames differ so far" % (seed, bad), flush=True)
print("seeds %d..%d x 3 sizes: %d frames differ; %d frames failed their verification and were rendered again" % (first, first + count - 1, bad, vetoes))
