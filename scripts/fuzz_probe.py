"""Development aid: one of tests/test_gpu_fuzz_shapes.py's random shapes (size:seed[:S]) under a list of switch settings — which of this
library's paths give the oracle's frame, and does the oracle's hierarchy give its own brute-force image (if not, the reference's
interval proofs are not facts for this shape: a partial function left its domain)."""
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

T = np.eye(4, dtype=np.float32)
T[3, 2] = 0.3
T2 = np.eye(3, dtype=np.float32)
settings = [{}, {"MPR_TILE_GEN_LOOSE": "0"}, {"MPR_TILE_GEN_LOOSE": "0", "MPR_TILE_GEN_GUARDS": "0"},
            {"MPR_TILE_GEN_LOOSE": "0", "MPR_VOXEL_GEN": "0"},
            {"MPR_TILE_GEN_LOOSE": "0", "MPR_TILE_GEN_LAST": "0"}, {"MPR_TILE_GEN_LOOSE": "0", "MPR_SKIP_STAGE0": "0"},
            {"MPR_LAST_STAGE_PUSH": "1"}, {"MPR_LAST_STAGE_PUSH": "1", "MPR_VOXEL_GEN": "0"},
            {"MPR_LAST_STAGE_PUSH": "1", "MPR_VOXEL_GEN": "0", "MPR_VOXEL_GROUPS": "0"},
            {"MPR_LAST_STAGE_PUSH": "1", "MPR_TILE_GEN": "0", "MPR_VOXEL_GEN": "0", "MPR_VOXEL_GROUPS": "0", "MPR_NORMALS_GEN": "0"}]
for arg in sys.argv[1:]:
    parts = arg.split(":")
    size, seed = int(parts[0]), int(parts[1])
    S = int(parts[2]) if len(parts) > 2 else 128
    tape = ns["fuzz_tape"](mpr, seed, size)
    print("%s: %d clauses, %d slots, %d min / max" % (arg, tape.length, tape.num_slots, tape.num_choices))
    ref = orc.Frame(tape.data, 3, S, mpr.colmajor(T, 4), threads=0)
    ref2 = orc.Frame(tape.data, 2, 2 * S, mpr.colmajor(T2, 3), z=0.05, threads=0)
    # (the reference has a brute-force renderer in 2-D only, and so has the oracle: render2D_brute, src/context.cu:1461-1508)
    brute2 = orc.Frame(tape.data, 2, 2 * S, mpr.colmajor(T2, 3), z=0.05, threads=0, brute=True)
    print("  the oracle's 2-D hierarchy against its brute force: %d pixels differ" % int((brute2.filled[3] != ref2.filled[3]).sum()))
    for env in settings:
        for key in list(os.environ):
            if key.startswith("MPR_"):
                del os.environ[key]
        os.environ.update(env)
        ctx = mpr.Context(S)
        out = []
        for _ in range(2):
            ctx.render3D(tape, T)
            out.append((int((ctx.image != ref.filled[3]).sum()), int((ctx.normals != ref.normals).sum())))
        forms, fk = ctx.tile_stage_forms(), ctx.float_kernel()
        ctx.close()
        ctx = mpr.Context(2 * S)
        ctx.render2D(tape, T2, 0.05)
        d2 = int((ctx.image != ref2.filled[3]).sum())
        print("  %s: 3-D (heights, normals) differing %s (%s, %s); 2-D %d (%s)" % (env, out, forms, fk, d2, ctx.tile_stage_forms()))
        ctx.close()
