"""Development aid: host time of mpr::Tape's construction (tree -> clauses -> the tape's walks as gfx950 code), per model.
usage: tape_times.py [MODEL ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpr_amd as m
for name in (sys.argv[1:] or ["bear", "architecture", "prospero", "hello_world", "involute_gear_3d", "involute_gear_2d"]):
    tree = m.model(name)
    ts = []
    for _ in range(8):
        a = time.perf_counter()
        t = m.Tape(tree)
        ts.append((time.perf_counter() - a) * 1e3)
    print("%-18s %5d clauses: %.1f ms (min of 8; median %.1f)" % (name, t.length, min(ts), sorted(ts)[4]))
