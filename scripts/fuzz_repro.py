"""Seeds scripts/fuzz_sweep.py reports, frame by frame, with the loose enclosures on and off.  usage: fuzz_repro.py SEED:SIZE ..."""
import faulthandler, os, sys
faulthandler.enable()
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
pairs = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]]
for seed, size in pairs:
    tape = ns["fuzz_tape"](mpr, seed, size)
    rng = np.random.default_rng(seed * 7 + size)
    S = int(rng.choice([128, 256]))
    view = T if rng.random() < 0.6 else random_view3(rng)
    ref = orc.Frame(tape.data, 3, S, mpr.colmajor(view, 4), threads=16)
    for sched in ("-",):
        for loose in ("1", "0"):
            os.environ["MPR_TILE_GEN_LOOSE"] = loose
            ctx = mpr.Context(S)
            out = []
            for k in range(3):
                ctx.render3D(tape, view)
                dh, dn = int((ctx.image != ref.filled[3]).sum()), int((ctx.normals != ref.normals).sum())
                out.append((dh, dn, ctx.tile_stage_forms(), ctx.normals_kernel()))
            print("seed %d size %d S %d identity_view %s clauses %d choices %d | sched %s loose %s:" % (seed, size, S, view is T, len(tape.data), tape.num_choices, sched, loose), out, "vetoes", ctx.skip0_vetoes(), flush=True)
            ctx.close()
    print("closing done; 2-D", flush=True)
    ref2 = orc.Frame(tape.data, 2, 256, mpr.colmajor(T2, 3), z=0.1, threads=16)
    ctx = mpr.Context(256)
    for k in range(2):
        ctx.render2D(tape, T2, 0.1)
        print("  2-D frame", k, int((ctx.image != ref2.filled[3]).sum()), ctx.tile_stage_forms(), flush=True)
    ctx.close()
print("done")
