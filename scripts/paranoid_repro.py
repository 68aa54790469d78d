"""Seeds scripts/paranoid_sweep.py reports: the same frames, 3-D and 2-D apart, against the oracle and under MPR_CTX_PARANOID, with the
last stage's second verdict on and off.  usage: paranoid_repro.py SEED:SIZE ..."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
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


T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
T2 = np.eye(3, dtype=np.float32)
for seed, size in [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]]:
    tape = ns["fuzz_tape"](mpr, seed, size)
    rng = np.random.default_rng(seed * 7 + size)
    S = int(rng.choice([128, 256]))
    view = T if rng.random() < 0.5 else random_view3(rng)
    z = float(np.float32(rng.uniform(-0.3, 0.3)))
    ops = sorted({c[0] for c in mpr.decode(tape.data)})
    print("seed %d size %d S %d identity %s clauses %d slots %d choices %d ops %s" % (seed, size, S, view is T, len(tape.data), tape.num_slots, tape.num_choices, ops), flush=True)
    ref = orc.Frame(tape.data, 3, S, mpr.colmajor(view, 4), threads=8)
    ref2 = orc.Frame(tape.data, 2, 256, mpr.colmajor(T2, 3), z=z, threads=8)
    for env in ({}, {"MPR_TILE_TIGHT": "0"}, {"MPR_VOXEL_COLS": "0"}, {"MPR_TILE_GEN_LOOSE": "0"}):
        os.environ.update(env)
        for flags, tag in ((0, "plain"), (mpr.CTX_PARANOID, "paranoid")):
            ctx = mpr.Context(S, flags=flags)
            out = []
            for k in range(3):
                ctx.render3D(tape, view)
                out.append((int((ctx.image != ref.filled[3]).sum()), int((ctx.normals != ref.normals).sum()), ctx.tile_stage_forms(), ctx.float_kernel()))
            st = ctx.paranoid_stats() if flags else None
            ctx.close()
            ctx = mpr.Context(256, flags=flags)
            out2 = []
            for k in range(2):
                ctx.render2D(tape, T2, z)
                out2.append((int((ctx.image != ref2.filled[3]).sum()), ctx.tile_stage_forms(), ctx.float_kernel()))
            st2 = ctx.paranoid_stats() if flags else None
            ctx.close()
            print("  %s %s: 3-D (heights, normals differ from the oracle) %s stats %s | 2-D %s stats %s" % (env, tag, out, st, out2, st2), flush=True)
        for k in env:
            del os.environ[k]
print("done")
