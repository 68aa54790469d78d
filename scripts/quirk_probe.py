"""Development aid: a model whose exact interval log takes its unsound branch (x.lo <= 0 -> lower bound 0, reference
inc/gpu_interval.hpp:382-390) — which frames of this library still give the oracle's image?  One line per switch setting."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import mpr_amd as mpr
from oracle import orc

orc.lib()
X, Y, Z = mpr.Tree.X(), mpr.Tree.Y(), mpr.Tree.Z()


def smooth(k):
    d1 = mpr.sqrt((X + 0.3) * (X + 0.3) + Y * Y + Z * Z) - 0.35
    d2 = mpr.sqrt((X - 0.3) * (X - 0.3) + (Y - 0.1) * (Y - 0.1) + Z * Z) - 0.3
    blend = mpr.log(mpr.exp(d1 * -k) + mpr.exp(d2 * -k)) / -k
    return mpr.tmax(mpr.tmin(blend, mpr.sqrt(X * X + (Y + 0.6) * (Y + 0.6) + Z * Z) / 3.0 - 0.1), Z - 0.25)


T = np.eye(4, dtype=np.float32)
T[3, 2] = 0.3
T2 = np.eye(3, dtype=np.float32)
settings = [{}, {"MPR_TILE_GEN_LOOSE": "0"}, {"MPR_LAST_STAGE_PUSH": "1"}, {"MPR_SKIP_STAGE0": "0", "MPR_TILE_GEN_LOOSE": "0"},
            {"MPR_SKIP_STAGE0": "0", "MPR_TILE_GEN_LOOSE": "0", "MPR_TILE_GEN_LAST": "0"},
            {"MPR_LAST_STAGE_PUSH": "1", "MPR_VOXEL_GEN": "0"}, {"MPR_LAST_STAGE_PUSH": "1", "MPR_VOXEL_GEN": "0", "MPR_VOXEL_GROUPS": "0"},
            {"MPR_LAST_STAGE_PUSH": "1", "MPR_TILE_GEN": "0", "MPR_VOXEL_GEN": "0", "MPR_VOXEL_GROUPS": "0"},
            {"MPR_LAST_STAGE_PUSH": "1", "MPR_TILE_GEN": "0", "MPR_VOXEL_GEN": "0", "MPR_VOXEL_GROUPS": "0", "MPR_NORMALS_GEN": "0",
             "MPR_WIDE_STAGE0": "0"}]
for k in (8.0, 64.0):
    tape = mpr.Tape(smooth(k))
    for S in (128, 512):
        ref = orc.Frame(tape.data, 3, S, mpr.colmajor(T, 4), threads=0)
        ref2 = orc.Frame(tape.data, 2, S, mpr.colmajor(T2, 3), threads=0)
        brute2 = orc.Frame(tape.data, 2, S, mpr.colmajor(T2, 3), threads=0, brute=True)      # (2-D only: the reference's render2D_brute)
        print("k=%g S=%d: the oracle's own 2-D hierarchy against its brute force: %d pixels differ" % (k, S, int((brute2.filled[3] != ref2.filled[3]).sum())))
        for env in settings:
            for key in list(os.environ):
                if key.startswith("MPR_"):
                    del os.environ[key]
            os.environ.update(env)
            ctx = mpr.Context(S)
            ctx.render3D(tape, T)
            d3 = int((ctx.image != ref.filled[3]).sum())
            dn = int((ctx.normals != ref.normals).sum())
            forms = ctx.tile_stage_forms()
            ctx.render2D(tape, T2)
            d2 = int((ctx.image != ref2.filled[3]).sum())
            print("k=%g S=%d %s: 3-D %d heights %d normals differ (%s, %s); 2-D %d" % (k, S, env, d3, dn, forms, ctx.float_kernel(), d2))
            ctx.close()
