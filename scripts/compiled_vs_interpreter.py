"""The paper's comparison: hierarchical interpreter vs brute-force interpreter vs the expression
compiled to machine code (reference benchmark/brute.cu, dump_tape.cpp), all on this GPU."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mpr_amd as m

def t(f, n=20, w=3):
    for _ in range(w): f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3

for name, S in (("prospero", 1024), ("involute_gear_2d", 1024), ("hello_world", 1024)):
    tape = m.Tape(m.model(name)); ctx = m.Context(S)
    t0 = time.perf_counter(); k = m.CompiledTape(tape); tc = time.perf_counter() - t0
    hier = t(lambda: ctx.render2D(tape))
    a = ctx.image.copy()
    brute = t(lambda: ctx.render2D_brute(tape), n=5, w=1)
    comp = t(lambda: k.render2D(ctx), n=5, w=1)
    assert np.array_equal(ctx.image, a)
    print("%-18s %d^2 (%d clauses): hierarchical %.3f ms  brute-force interpreter %.3f ms  compiled %.3f ms (+ %.1f s to compile)"
          % (name, S, tape.length - 2, hier, brute, comp, tc), flush=True)
    k.close(); ctx.close()
