"""Development aid (gpurun): frames that leave the reference's tiles and tapes behind (MPR_LAST_STAGE_PUSH=1) WITH the second verdict in a
launch of its own and the float pass by segments made beside the reference's list ('+verdict', k_eval_voxels_gen_fp) against the same
frames WITHOUT either (MPR_TILE_TIGHT=0, MPR_VOXEL_FP=0: every listed tile walked, tile by tile): heights, normals and the list of
smallest tiles must be the same.  bear under random views at 256^3 .. 1024^3, then scripts/tight_sweep.py's random shapes.
usage: full_frames_sweep.py [VIEWS_OF_BEAR] [FIRST_SEED COUNT]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpr_amd as mpr

NV = int(sys.argv[1]) if len(sys.argv) > 1 else 60
FIRST = int(sys.argv[2]) if len(sys.argv) > 2 else 0
COUNT = int(sys.argv[3]) if len(sys.argv) > 3 else 300

src = open(os.path.join(ROOT, "scripts", "tight_sweep.py")).read()
ns = {"np": np, "mpr": mpr}
exec(src[src.index("def view3"):src.index("t0 = time.time()")], ns)
view3, shape = ns["view3"], ns["shape"]


def contexts(S):
    os.environ["MPR_LAST_STAGE_PUSH"] = "1"
    a = mpr.Context(S)
    os.environ["MPR_TILE_TIGHT"] = "0"
    os.environ["MPR_VOXEL_FP"] = "0"
    b = mpr.Context(S)
    for k in ("MPR_LAST_STAGE_PUSH", "MPR_TILE_TIGHT", "MPR_VOXEL_FP"):
        del os.environ[k]
    return a, b


def same(a, b, tape, V, what):
    bad = 0
    for k in range(2):
        a.render3D(tape, V)
        b.render3D(tape, V)
        dh, dn = int((a.image != b.image).sum()), int((a.normals != b.normals).sum())
        ta, tb = a.stages[3].tiles, b.stages[3].tiles
        pa, pb = np.sort(ta["position"][ta["position"] != -1]), np.sort(tb["position"][tb["position"] != -1])
        dt = 0 if np.array_equal(pa, pb) else 1
        if dh or dn or dt:
            bad += 1
            print("DIFFERS %s frame %d: heights %d normals %d tiles %d (%s | %s)" % (what, k, dh, dn, dt, a.tile_stage_forms(), a.float_kernel()), flush=True)
    return bad


t0 = time.time()
T = np.eye(4, dtype=np.float32)
T[3, 2] = 0.3
tape = mpr.Tape(mpr.model("bear"))
total = bad = verdicts = 0
for S in (256, 512, 1024):
    rng = np.random.default_rng(S * 31 + 5)
    a, b = contexts(S)
    n = NV if S < 1024 else max(NV // 4, 4)
    for k in range(n):
        V = view3(rng) if k else T
        bad += same(a, b, tape, V, "bear %d view %d" % (S, k))
        total += 2
        verdicts += "+verdict" in a.tile_stage_forms() and a.float_kernel() == "k_eval_voxels_gen_fp<3>"
    a.close()
    b.close()
    print("bear %4d: %d views; %d frames so far, %d differ, %d views with '+verdict' and segments; %.0f s" % (S, n, total, bad, verdicts, time.time() - t0), flush=True)
ctxs = {S: contexts(S) for S in (256, 512)}
for seed in range(FIRST, FIRST + COUNT):
    rng = np.random.default_rng(77000 + seed)
    tape = mpr.Tape(shape(rng))
    S = int(rng.choice([256, 512]))
    a, b = ctxs[S]
    for V in (T, view3(rng)):
        bad += same(a, b, tape, V, "shape %d S %d" % (seed, S))
        total += 2
        verdicts += "+verdict" in a.tile_stage_forms() and a.float_kernel() == "k_eval_voxels_gen_fp<3>"
print("total: %d frames compared, %d differ; %d views ran '+verdict' with the float pass by segments; %.0f s" % (total, bad, verdicts, time.time() - t0))
