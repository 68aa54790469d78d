"""Tile-parallel rendering of ONE frame on N GPUs.

The unit of partitioning is a top-level xy column (a 64x64 pixel footprint with all of its
S/64 z tiles): columns never interact (3-D tiles only talk to the heightmap of their own
column, reference src/context.cu:299-316, :489-494, :861, :941-947), so a rank that renders
only its columns produces exactly the pixels the single-GPU frame has there.  Per frame:

    render the owned columns  ->  pack their 64x64 blocks  ->  ONE all-gather (RCCL over
    xGMI; point-to-point links, every rank pushes its pack to every peer)  ->  unpack

No reduction is involved, so the result is bit-identical to the single-GPU frame.

The column -> rank deal is the longest-processing-time-first heuristic on a per-column work proxy
every rank computes for itself, identically, without communication: by default the number of
first-stage (64 px) tiles of a column that the interval evaluation leaves ambiguous (one cheap
stage over the whole frame); optionally (plan(feedback=True)) the smallest tiles
per column of a full frame rendered beforehand.

The class is written against a small "context" protocol (render3D / render3D_part /
pack_columns / unpack_columns / stages[3].tiles) so that the world-size-2 gloo test in
tests/test_distributed.py can drive the same code with a CPU stand-in.
"""
import numpy as np


def column_weights(tiles, size, dim=3):
    """Per-column work proxy: number of smallest tiles (4^3 voxels / 8^2 pixels) per 64x64 column."""
    cols = size // 64
    sub = 4 if dim == 3 else 8
    tps = size // sub
    pos = np.asarray(tiles["position"], dtype=np.int64)
    pos = pos[pos >= 0]
    x = pos % tps
    y = (pos // tps) % tps
    col = (x * sub // 64) + (y * sub // 64) * cols
    return np.bincount(col, minlength=cols * cols).astype(np.float32)


class TileParallelRenderer:
    def __init__(self, ctx, mpr, rank, world, make_buffer, all_gather, dim=3):
        """ctx: an mpr_amd.Context (or stand-in) on this rank's device.
        make_buffer(n_int32) -> (handle, device_pointer): gather staging memory.
        all_gather(out_handle, in_handle): collective over equal-size int32 buffers."""
        self.ctx, self.mpr = ctx, mpr
        self.rank, self.world = rank, world
        self.dim = dim
        self.size = ctx.image_size_px
        self.cols = (self.size // 64) ** 2
        self.make_buffer, self.all_gather = make_buffer, all_gather
        self.owner = None
        self.capacity = 0
        self.with_normals = dim == 3

    def plan(self, tape, mat, z=0.0, feedback=False):
        """Deal the columns (identical on all ranks, no communication).  Default: by the first tile stage's own verdict — every
        rank runs the 64 px stage over the whole frame (a fraction of a millisecond) and weighs a column by the tiles it leaves
        ambiguous (the stage-0 proxy).  feedback=True: by the smallest tiles per column of one full frame
        rendered first — a better balance for a view that repeats (what a benchmark does; say so when you use it)."""
        if not feedback and hasattr(self.ctx, "column_weights"):
            w = self.ctx.column_weights(tape, mat, z, self.dim)
        else:
            if self.dim == 3:
                self.ctx.render3D(tape, mat)
            else:
                self.ctx.render2D(tape, mat, z)
            w = column_weights(self.ctx.stages[3].tiles, self.size, self.dim)
        self.set_owner(self.mpr.partition_columns(self.cols, self.world, w))
        return self.owner

    def set_owner(self, owner):
        self.owner = np.ascontiguousarray(owner, dtype=np.int32)
        counts = np.bincount(self.owner, minlength=self.world)
        self.capacity = int(counts.max())
        per_rank = self.capacity * 4096 * (2 if self.with_normals else 1)
        self.send, self.send_ptr = self.make_buffer(per_rank)
        self.recv, self.recv_ptr = self.make_buffer(per_rank * self.world)
        self.per_rank = per_rank
        # steady-state path: resident plan, one pack and one unpack launch, no host waits in between
        self.planned = hasattr(self.ctx, "gather_plan")
        if self.planned:
            self.ctx.gather_plan(self.owner, self.rank, self.world, self.capacity, self.with_normals)

    def render(self, tape, mat, z=0.0):
        """One frame: afterwards ctx.image / ctx.normals hold the complete result on every rank."""
        if self.world == 1:
            if self.dim == 3:
                self.ctx.render3D(tape, mat)
            else:
                self.ctx.render2D(tape, mat, z)
            return
        if self.planned:
            # everything is enqueued on the context's stream; `all_gather` must order the collective
            # after that stream (bench.py runs it under torch.cuda.stream(ExternalStream(ctx.stream)))
            if self.dim == 3:
                self.ctx.render3D_part(tape, mat, self.owner, self.rank, blocking=False)
            else:
                self.ctx.render2D_part(tape, mat, z, self.owner, self.rank, blocking=False)
            self.ctx.pack_planned(self.send_ptr)
            self.all_gather(self.recv, self.send)
            self.ctx.unpack_planned(self.recv_ptr)
            self.ctx.sync()
            return
        if self.dim == 3:
            self.ctx.render3D_part(tape, mat, self.owner, self.rank)
        else:
            self.ctx.render2D_part(tape, mat, z, self.owner, self.rank)
        self.ctx.pack_columns(self.owner, self.rank, self.capacity, self.with_normals, self.send_ptr)
        self.all_gather(self.recv, self.send)
        for r in range(self.world):
            if r == self.rank:
                continue
            self.ctx.unpack_columns(self.owner, r, self.capacity, self.with_normals,
                                    self.recv_ptr + r * self.per_rank * 4)
