"""Regenerate tests/golden/frames.json with the CPU oracle.

The reference stores no expected outputs for this path and cannot be run
here, so these goldens are produced by the oracle (oracle/mpr_oracle.c) after it has been
pinned by the known-answer tests in tests/test_oracle_kat.py.  They freeze the oracle's
results so that a later change to oracle, tape builder or kernels is noticed.

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import mpr_amd  # noqa: E402  (host front end only: .frep reader + tape builder)
from oracle import orc  # noqa: E402

FRAMES = [("circle", 2, 128), ("circle", 2, 256), ("two_spheres", 2, 256), ("ring", 2, 256),
          ("hello_world", 2, 256), ("prospero", 2, 256), ("prospero", 2, 1024), ("involute_gear_2d", 2, 512),
          ("two_spheres", 3, 256), ("hello_world", 3, 256), ("bear", 3, 256), ("architecture", 3, 256),
          ("involute_gear_3d", 3, 256)]


def tape_of(name):
    X, Y, Z = mpr_amd.Tree.X(), mpr_amd.Tree.Y(), mpr_amd.Tree.Z()
    if name == "circle":
        t = mpr_amd.sqrt((X + 1) * (X + 1) + (Y + 1) * (Y + 1)) - 1.8
    elif name == "two_spheres":
        t = mpr_amd.tmin(mpr_amd.sqrt((X + 0.5) * (X + 0.5) + Y * Y + Z * Z) - 0.25,
                         mpr_amd.sqrt((X - 0.5) * (X - 0.5) + Y * Y + Z * Z) - 0.25)
    elif name == "ring":
        t = mpr_amd.tmax(mpr_amd.sqrt(X * X + Y * Y) - 1, 0.5 - mpr_amd.sqrt(X * X + Y * Y))
    else:
        t = mpr_amd.model(name)
    return mpr_amd.Tape(t)


def main():
    mpr_amd.build()
    out = {"generator": "tests/golden/make_golden.py (oracle/mpr_oracle.c)", "frames": []}
    for name, dim, size in FRAMES:
        tape = tape_of(name).data
        if dim == 2:
            mat = np.eye(3, dtype=np.float32)
        else:
            mat = np.eye(4, dtype=np.float32)
            mat[3, 2] = 0.3
        f = orc.Frame(tape, dim, size, mpr_amd.colmajor(mat, dim + 1), threads=1, keep_pool=False)
        rec = {"model": name, "dim": dim, "size": size,
               "tape_sha256": hashlib.sha256(tape.tobytes()).hexdigest(),
               "image_sha256": hashlib.sha256(f.image.tobytes()).hexdigest(),
               "filled_cells": int((f.image != 0).sum()),
               "tiles_in": f.counters["tiles_in"], "tiles_active": f.counters["tiles_active"],
               "tiles_empty": f.counters["tiles_empty"], "tiles_filled": f.counters["tiles_filled"],
               "voxel_tiles": f.counters["voxel_tiles"]}
        if dim == 3:
            rec["normals_sha256"] = hashlib.sha256(f.normals.tobytes()).hexdigest()
            rec["max_height"] = int(f.image.max())
        out["frames"].append(rec)
        print(rec)
    with open(os.path.join(HERE, "frames.json"), "w") as fp:
        json.dump(out, fp, indent=1)


if __name__ == "__main__":
    main()
