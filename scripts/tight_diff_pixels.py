"""Development aid (gpurun): where a default frame's heights differ from the oracle's, for seeds scripts/paranoid_sweep.py reports.
usage: tight_diff_pixels.py SEED:SIZE ..."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import mpr_amd as mpr
from oracle import orc
src = open(os.path.join(ROOT, "tests", "test_gpu_fuzz_shapes.py")).read().split("@pytest.mark.parametrize")[0]
src = src.replace("from conftest import view2, view3", "").replace("from helpers import check_default_path, compare_frame, compare_reader_frame", "")
ns = {}
exec(src, ns)
T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
for seed, size in [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]]:
    tape = ns["fuzz_tape"](mpr, seed, size)
    rng = np.random.default_rng(seed * 7 + size)
    S = int(rng.choice([128, 256]))
    assert rng.random() < 0.5
    ref = orc.Frame(tape.data, 3, S, mpr.colmajor(T, 4), threads=8)
    ctx = mpr.Context(S)
    for k in range(3):
        ctx.render3D(tape, T)
    img = ctx.image
    ys, xs = np.nonzero(img != ref.filled[3])
    print("seed %d size %d S %d: %d pixels" % (seed, size, S, len(xs)))
    for x, y in list(zip(xs, ys))[:12]:
        print("   pixel (%d, %d): gpu %d oracle %d" % (x, y, img[y, x], ref.filled[3][y, x]))
    ctx.close()
