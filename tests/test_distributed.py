"""World-size-2 test of the tile-parallel frame (mpr_amd/multigpu.py) over gloo on CPU.

The GPU context is replaced by a stand-in that renders a rank's columns with the oracle and
implements pack / unpack in numpy; everything else — planning, the column deal (the product's
mpr_partition_columns), the gather and the unpack loop — is the code bench.py runs with
--gpus N.  The assembled frame must equal the single-rank frame bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleContext:
    """mpr_amd.Context look-alike backed by the oracle (test double)."""

    def __init__(self, size, mpr, orc):
        self.image_size_px = size
        self.mpr, self.orc = mpr, orc
        self.image = np.zeros((size, size), dtype=np.int32)
        self.normals = np.zeros((size, size), dtype=np.uint32)
        self.stages = [None, None, None, self]
        self.tiles = None

    def _run(self, tape, mat, owner=None, rank=0):
        f = self.orc.Frame(tape.data, 3, self.image_size_px, self.mpr.colmajor(mat, 4), threads=2, owner=owner,
                           rank=rank, keep_pool=False)
        self.image, self.normals, self.tiles = f.image.copy(), f.normals.copy(), f.tiles[3]

    def render3D(self, tape, mat):
        self._run(tape, mat)

    def render3D_part(self, tape, mat, owner, rank):
        self._run(tape, mat, owner, rank)

    def _cols(self, owner, rank):
        return np.flatnonzero(np.asarray(owner) == rank)

    def pack_columns(self, owner, rank, capacity, with_normals, ptr):
        buf = _as_array(ptr, capacity * 4096 * (2 if with_normals else 1))
        cols = self.image_size_px // 64
        for i, c in enumerate(self._cols(owner, rank)):
            x0, y0 = (c % cols) * 64, (c // cols) * 64
            buf[i * 4096:(i + 1) * 4096] = self.image[y0:y0 + 64, x0:x0 + 64].reshape(-1)
            if with_normals:
                buf[(capacity + i) * 4096:(capacity + i + 1) * 4096] = \
                    self.normals[y0:y0 + 64, x0:x0 + 64].reshape(-1).view(np.int32)

    def unpack_columns(self, owner, rank, capacity, with_normals, ptr):
        buf = _as_array(ptr, capacity * 4096 * (2 if with_normals else 1))
        cols = self.image_size_px // 64
        for i, c in enumerate(self._cols(owner, rank)):
            x0, y0 = (c % cols) * 64, (c // cols) * 64
            self.image[y0:y0 + 64, x0:x0 + 64] = buf[i * 4096:(i + 1) * 4096].reshape(64, 64)
            if with_normals:
                self.normals[y0:y0 + 64, x0:x0 + 64] = \
                    buf[(capacity + i) * 4096:(capacity + i + 1) * 4096].view(np.uint32).reshape(64, 64)


def _as_array(ptr, n):
    import ctypes
    return np.ctypeslib.as_array((ctypes.c_int32 * n).from_address(ptr))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import mpr_amd
    from mpr_amd.multigpu import TileParallelRenderer
    from oracle import orc
    from conftest import view3

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S = 256
    tape = mpr_amd.Tape(mpr_amd.model("hello_world"))
    ctx = OracleContext(S, mpr_amd, orc)

    def make_buffer(n):
        t = torch.zeros(n, dtype=torch.int32)
        return t, t.data_ptr()

    def all_gather(out, inp):
        dist.all_gather_into_tensor(out, inp)

    tpr = TileParallelRenderer(ctx, mpr_amd, rank, world, make_buffer, all_gather, dim=3)
    owner = tpr.plan(tape, view3())
    assert set(owner.tolist()) == set(range(world))
    tpr.render(tape, view3())
    np.save(os.path.join(out_dir, "img%d.npy" % rank), ctx.image)
    np.save(os.path.join(out_dir, "nrm%d.npy" % rank), ctx.normals)
    np.save(os.path.join(out_dir, "own%d.npy" % rank), owner)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_frame_equals_single_rank(tmp_path, mpr, orc):
    import torch.multiprocessing as mp
    from conftest import view3
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    tape = mpr.Tape(mpr.model("hello_world"))
    ref = orc.Frame(tape.data, 3, 256, mpr.colmajor(view3(), 4), threads=0, keep_pool=False)
    own0 = np.load(tmp_path / "own0.npy")
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / ("own%d.npy" % r)), own0)      # same deal on every rank
        assert np.array_equal(np.load(tmp_path / ("img%d.npy" % r)), ref.image)
        assert np.array_equal(np.load(tmp_path / ("nrm%d.npy" % r)), ref.normals)
    load = np.bincount(own0, minlength=world)
    assert load.min() > 0


def test_partitioned_oracle_frames_are_disjoint_and_complete(mpr, orc):
    """Rendering only the owned columns leaves every other pixel 0 and reproduces the owned ones."""
    from conftest import view3
    tape = mpr.Tape(mpr.model("bear"))
    S = 128
    full = orc.Frame(tape.data, 3, S, mpr.colmajor(view3(), 4), keep_pool=False)
    owner = mpr.partition_columns((S // 64) ** 2, 2)
    acc = np.zeros_like(full.image)
    for r in range(2):
        part = orc.Frame(tape.data, 3, S, mpr.colmajor(view3(), 4), owner=owner, rank=r, keep_pool=False)
        mask = np.kron((owner.reshape(S // 64, S // 64) == r), np.ones((64, 64), dtype=bool))
        assert not part.image[~mask].any()
        assert np.array_equal(part.image[mask], full.image[mask])
        acc += part.image
    assert np.array_equal(acc, full.image)
