import sys, os; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, mpr_amd as m
from oracle import orc
from test_gpu_fuzz import random_tree
from conftest import view3
seed = int(sys.argv[1])
tape = m.Tape(random_tree(m, 2000 + seed))
d = np.asarray(tape.data); print("len", len(d), "slots", tape.num_slots, "choices", tape.num_choices)
ref = orc.Frame(tape.data, 3, 128, m.colmajor(view3(), 4), threads=0, keep_pool=False)
for env in ({}, {"MPR_TILE_GEN_LAST": "0"}, {"MPR_TILE_GEN_LAST": "0", "MPR_NORMALS_GEN": "0"}, {"MPR_TILE_GEN": "0"}):
    for k in ("MPR_TILE_GEN_LAST", "MPR_NORMALS_GEN", "MPR_TILE_GEN"): os.environ.pop(k, None)
    os.environ.update(env)
    ctx = m.Context(128)
    out = []
    for f in range(3):
        ctx.render3D(tape, view3())
        out.append((int((np.array(ctx.image) != ref.filled[3]).sum()), int((np.array(ctx.normals) != ref.normals).sum()), ctx.normals_kernel(), ctx.float_kernel()))
    print(env, out)
    ctx.close()
