"""Cycles of one scheduled interval forward walk (csrc/interval_gen.cpp) on the chip, alone and with four wavefronts per SIMD, for the
scheduler's windows — development aid (mpr_debug_walk_cycles)."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpr_amd as m

L = m.lib()
f = L.mpr_debug_walk_cycles
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
              ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]


def cycles(words, kind, loose, window, waves, empty=0, reps=8):
    arr = np.asarray(words, dtype=np.uint64)
    out = np.zeros(waves, dtype=np.int64)
    red = ctypes.c_uint32(0)
    info = (ctypes.c_int32 * 4)()
    rc = f(0, arr.ctypes.data, len(arr), kind, loose, window, empty, reps, waves, out.ctypes.data, ctypes.byref(red), info)
    if rc != 0:
        return None
    return float(np.median(out)), float(out.max()), red.value, list(info)


if __name__ == "__main__":
    names = sys.argv[1:] or ["bear"]
    os.makedirs("gpurun_out/r05b", exist_ok=True)
    for name in names:
        words = m.Tape(m.model(name)).data
        print(name, "harness alone: 1 wave %s | 4096 waves %s" % (cycles(words, 0, 0, 0, 1, empty=1)[:2], cycles(words, 0, 0, 0, 4096, empty=1)[:2]), flush=True)
        for kind in (0, 2):
            for loose in (1, 0):
                for window in ((1, 8) if loose else (1, 2)):
                    a = cycles(words, kind, loose, window, 1)
                    if a is None:
                        print(name, "kind", kind, "loose", loose, "window", window, "not generated")
                        continue
                    b = cycles(words, kind, loose, window, 1024)
                    c = cycles(words, kind, loose, window, 4096)
                    print(name, "kind", kind, "loose", loose, "window", window, "instr/w/vgprs/est", a[3], "| 1 wave: %.0f | 1/SIMD: med %.0f max %.0f | 4/SIMD: med %.0f max %.0f redone %d" %
                          (a[0], b[0], b[1], c[0], c[1], c[2]), flush=True)
