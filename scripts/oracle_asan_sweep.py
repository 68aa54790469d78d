"""VERDICT r5 next-2, the leg that needs no GPU: the ORACLE's frames of scripts/fuzz_sweep.py's shapes (seeds FIRST..FIRST+COUNT-1 x sizes 3 / 8 / 16,
3-D and 2-D, 16 OpenMP threads as the sweep ran them) under AddressSanitizer:
    make -C oracle asan; MPR_ORACLE_ASAN=1 LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python scripts/oracle_asan_sweep.py FIRST COUNT"""
import os, sys, faulthandler
faulthandler.enable()
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
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
T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
T2 = np.eye(3, dtype=np.float32)
frames = 0
for seed in range(first, first + count):
    for size in (3, 8, 16):
        tape = ns["fuzz_tape"](mpr, seed, size)
        rng = np.random.default_rng(seed * 7 + size)
        S = int(rng.choice([128, 256]))
        view = T if rng.random() < 0.6 else random_view3(rng)
        orc.Frame(tape.data, 3, S, mpr.colmajor(view, 4), threads=16)
        orc.Frame(tape.data, 2, 256, mpr.colmajor(T2, 3), z=0.1, threads=16)
        frames += 2
print("seeds %d..%d x 3 sizes: %d oracle frames under AddressSanitizer, no report" % (first, first + count - 1, frames))
