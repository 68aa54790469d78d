"""Cycles per clause of the tile stages' assembly forward walk, one wavefront on an idle chip
(development aid)."""
import sys, os, ctypes; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mpr_amd as m
L = m.lib()
L.mpr_debug_interp_cycles.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
OP = m.OP
def cl(op, out, lhs, rhs, imm=0.0):
    return m.clause(OP[op], out, lhs, rhs, imm) if hasattr(m, "clause") else None
def run(name, body, waves=1):
    tape = [m.clause(0, 1, 2, 3, 0.0)] + body + [m.clause(0, body[-1] >> 8 & 0xff if False else 4, 0, 0, 0.0)]
    arr = np.array(tape, dtype=np.uint64)
    out = np.zeros(8, dtype=np.int64)
    rc = L.mpr_debug_interp_cycles(0, arr.ctypes.data, arr.size, out.size, waves, out.ctypes.data)
    assert rc == 0
    n = len(body)
    print("%-34s %3d clauses: %6d cycles  -> %5.1f per clause (min of %d)" % (name, n, out[1:].min(), out[1:].min() / n, out.size - 1))
    return out[1:].min()
N = 60
base = run("empty-ish (1 neg)", [m.clause(OP["NEG_LHS"], 4, 1, 0, 0.0)])
run("add chain, forwarded", [m.clause(OP["ADD_LHS_RHS"], 4, 4 if i else 1, 2, 0.0) for i in range(N)])
run("add, operands from LDS", [m.clause(OP["ADD_LHS_RHS"], 4 + (i & 1), 1, 2, 0.0) for i in range(N)])
run("add imm chain, forwarded", [m.clause(OP["ADD_LHS_IMM"], 4, 4 if i else 1, 0, 0.5) for i in range(N)])
run("add imm, from LDS", [m.clause(OP["ADD_LHS_IMM"], 4 + (i & 1), 1, 0, 0.5) for i in range(N)])
run("neg chain, forwarded", [m.clause(OP["NEG_LHS"], 4, 4 if i else 1, 0, 0.0) for i in range(N)])
run("mul, from LDS", [m.clause(OP["MUL_LHS_RHS"], 4 + (i & 1), 1, 2, 0.0) for i in range(N)])
run("min, from LDS", [m.clause(OP["MIN_LHS_RHS"], 4 + (i & 1), 1, 2, 0.0) for i in range(N)])
run("sqrt (leaves the block), LDS", [m.clause(OP["SQRT_LHS"], 4 + (i & 1), 1, 0, 0.0) for i in range(N)])
run("exp (leaves the block), LDS", [m.clause(OP["EXP_LHS"], 4 + (i & 1), 1, 0, 0.0) for i in range(N)])
run("div (leaves the block), LDS", [m.clause(OP["DIV_LHS_RHS"], 4 + (i & 1), 1, 2, 0.0) for i in range(N)])
run("not an opcode (exit overhead only)", [m.clause(30, 4 + (i & 1), 1, 0, 0.0) for i in range(N)])
run("log (leaves the block), LDS", [m.clause(OP["LOG_LHS"], 4 + (i & 1), 1, 0, 0.0) for i in range(N)])
run("atan (leaves the block), LDS", [m.clause(OP["ATAN_LHS"], 4 + (i & 1), 1, 0, 0.0) for i in range(N)])
