"""One-off wider fuzz (development aid): 60 deeper random trees at 256^3 and 512^2 against the oracle.
    python tests/tools/fuzz_more.py    (on the GPU box)"""
import sys; import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, mpr_amd as m
from oracle import orc
from test_gpu_fuzz import random_tree
from helpers import check_default_path, compare_frame
from conftest import view2, view3
bad = 0
for seed in range(5000, 5060):
    tape = m.Tape(random_tree(m, seed, depth=5))
    try:
        cnt, ref = compare_frame(m, orc, tape, 3, 256, view3())
        check_default_path(m, ref, tape, 3, 256, view3())          # generated code, repeated frames
        cnt, ref = compare_frame(m, orc, tape, 2, 512, view2(), z=0.1)
        check_default_path(m, ref, tape, 2, 512, view2(), z=0.1)
    except AssertionError as e:
        bad += 1; print("seed", seed, "FAILED", str(e)[:200])
print("extra fuzz: 60 trees x (3-D 256^3, 2-D 512^2), instrumented frames and the default path over 3 frames:", "all bit-exact" if bad == 0 else "%d failures" % bad)
