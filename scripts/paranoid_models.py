"""Development aid (gpurun): the benchmark's models under random views — rotated, sheared, mirrored, zoomed, perspective along any
axis — in contexts that render every frame both ways and compare on the device (MPR_CTX_PARANOID): the fast frames' equivalence on the
tapes the numbers are quoted on, in views the benchmarks never use (frames that start at the 16^3 tiles, loose stages, the generated
first-stage walk of many-slot tapes, either form of the float pass, all three forms of the normals pass).
usage: paranoid_models.py [VIEWS_PER_CONFIGURATION]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpr_amd as mpr

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60


def view3(rng):
    V = np.eye(4, dtype=np.float32)
    if rng.random() < 0.8:
        V[:3, :3] += rng.uniform(-0.35, 0.35, (3, 3)).astype(np.float32)
    V[:3, :3] *= np.float32(rng.choice([0.6, 1.0, 1.0, 1.7]))
    if rng.random() < 0.3:
        V[int(rng.integers(0, 3))] *= np.float32(-1.0)
    V[:3, 3] = rng.uniform(-0.3, 0.3, 3).astype(np.float32)
    V[3, :3] = rng.uniform(-0.3, 0.3, 3).astype(np.float32)
    return V


def view2(rng):
    V = np.eye(3, dtype=np.float32)
    V[:2, :2] += rng.uniform(-0.35, 0.35, (2, 2)).astype(np.float32)
    V[:2, :2] *= np.float32(rng.choice([0.6, 1.0, 1.7]))
    V[:2, 2] = rng.uniform(-0.3, 0.3, 2).astype(np.float32)
    V[2, :2] = rng.uniform(-0.2, 0.2, 2).astype(np.float32)
    return V


CONFIGS = [("bear", 3, (256, 512, 1024)), ("architecture", 3, (256, 1024, 1536, 2048)), ("involute_gear_3d", 3, (256, 1024)), ("hello_world", 3, (512, 1024)),
           ("prospero", 2, (256, 1024, 2048)), ("involute_gear_2d", 2, (1024, 4096)), ("hello_world", 2, (1024,))]
t0 = time.time()
grand = [0, 0, 0]
for name, dim, sizes in CONFIGS:
    tape = mpr.Tape(mpr.model(name))
    for S in sizes:
        n = N if S <= 1024 else max(N // 4, 8)
        rng = np.random.default_rng(S * 31 + dim + len(name))
        ctx = mpr.Context(S, flags=mpr.CTX_PARANOID)
        forms = set()
        nonempty = 0
        for k in range(n):
            if dim == 3:
                V = view3(rng)
                for _ in range(2):
                    ctx.render3D(tape, V)
            else:
                V, z = view2(rng), float(np.float32(rng.uniform(-0.3, 0.3)))
                for _ in range(2):
                    ctx.render2D(tape, V, z)
            forms.add(ctx.tile_stage_forms())
            if k % 8 == 0:
                nonempty += int(ctx.image.any())
        st = ctx.paranoid_stats()
        ctx.close()
        grand = [a + b for a, b in zip(grand, st)]
        print("%-17s %dD %4d: %3d views x 2 frames, %4d rendered twice, %d cells differ; %d of %d sampled images not empty; %d forms; %.0f s"
              % (name, dim, S, n, st[1], st[2], nonempty, (n + 7) // 8, len(forms), time.time() - t0), flush=True)
print("total: %d frames, %d rendered twice, %d cells differ" % tuple(grand))
