"""Writes tests/golden/shapes/shape_<primitives>_<seed>.frep: a few of tests/test_gpu_fuzz_shapes.py's random shapes as .frep files
(this package's own serialisation of its own trees: mpr_amd.Tree.to_frep), so that tests/golden/make_independent.py — which shares
no code with the oracle or the product — can evaluate them: the oracle is pinned on shapes with divisions by negative constants,
steep exp / log blends and asin / acos, not only on the reference's six models."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import mpr_amd as mpr   # noqa: E402

SHAPES = [(12, 0), (12, 3), (4, 15), (4, 9), (12, 21), (64, 5)]


def main():
    src = open(os.path.join(os.path.dirname(HERE), "test_gpu_fuzz_shapes.py")).read().split("@pytest.mark.parametrize")[0]
    src = src.replace("from conftest import view2, view3", "").replace("from helpers import check_default_path, compare_frame, compare_reader_frame", "")
    ns = {}
    exec(src, ns)
    import numpy as np
    import zlib
    os.makedirs(os.path.join(HERE, "shapes"), exist_ok=True)
    for size, seed in SHAPES:
        rng = np.random.default_rng(zlib.crc32(b"fuzz") + seed)
        tree = None
        for _ in range(50):            # (as fuzz_tape: the first tree that makes a tape)
            t = ns["random_tree"](mpr, rng, size)
            try:
                mpr.Tape(t)
            except mpr.MprError:
                continue
            tree = t
            break
        data = tree.to_frep()
        path = os.path.join(HERE, "shapes", "shape_%d_%d.frep" % (size, seed))
        open(path, "wb").write(data)
        print(path, len(data), "bytes")


if __name__ == "__main__":
    main()
