"""ctypes loader for the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing in the mpr_amd package imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Counters(ctypes.Structure):
    _fields_ = [
        ("tiles_in", ctypes.c_int64 * 3),
        ("tiles_empty", ctypes.c_int64 * 3),
        ("tiles_filled", ctypes.c_int64 * 3),
        ("tiles_masked", ctypes.c_int64 * 3),
        ("tiles_active", ctypes.c_int64 * 3),
        ("tiles_pushed", ctypes.c_int64 * 3),
        ("voxel_tiles", ctypes.c_int64),
        ("clauses_fwd", ctypes.c_int64),
        ("clauses_fwd_tiles", ctypes.c_int64),
        ("clauses_fwd_voxels", ctypes.c_int64),
        ("clauses_fwd_normals", ctypes.c_int64),
        ("clauses_bwd", ctypes.c_int64),
        ("clauses_written", ctypes.c_int64),
        ("lane_clauses", ctypes.c_int64),
        ("normal_pixels", ctypes.c_int64),
        ("tape_index", ctypes.c_int32),
        ("pool_overflowed", ctypes.c_int32),
        ("slots_exceeded", ctypes.c_int32),
        ("threads", ctypes.c_int32),
        ("clauses_fwd_voxels_min", ctypes.c_int64),
        ("clauses_fwd_voxels_max", ctypes.c_int64),
        ("clauses_written_survivors", ctypes.c_int64),
    ]

    def as_dict(self):
        out = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            out[name] = list(v) if hasattr(v, "__len__") else v
        return out


TILE_DTYPE = np.dtype([("position", "<i4"), ("tape", "<i4"), ("next", "<i4")])


def build():
    """Compile liboracle.so from oracle/mpr_oracle.c (gcc)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "liboracle.so")
    if os.environ.get("MPR_ORACLE_ASAN") == "1":       # scripts/segv_hunt.sh: the build with AddressSanitizer (make -C oracle asan)
        path = os.path.join(_HERE, "liboracle_asan.so")
    if not os.path.exists(path):
        build()
    L = ctypes.CDLL(path)
    P = ctypes.POINTER
    L.orc_render.restype = ctypes.c_void_p
    L.orc_render.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                             ctypes.c_void_p, ctypes.c_float, ctypes.c_int64, ctypes.c_int32,
                             ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32]
    L.orc_frame_free.argtypes = [ctypes.c_void_p]
    L.orc_filled.restype = P(ctypes.c_int32)
    L.orc_filled.argtypes = [ctypes.c_void_p, ctypes.c_int32, P(ctypes.c_size_t)]
    L.orc_normals.restype = P(ctypes.c_uint32)
    L.orc_normals.argtypes = [ctypes.c_void_p, P(ctypes.c_size_t)]
    L.orc_heatmap.restype = P(ctypes.c_float)
    L.orc_heatmap.argtypes = [ctypes.c_void_p, P(ctypes.c_size_t)]
    L.orc_tiles.restype = ctypes.c_void_p
    L.orc_tiles.argtypes = [ctypes.c_void_p, ctypes.c_int32, P(ctypes.c_size_t)]
    L.orc_tape_pool.restype = P(ctypes.c_uint64)
    L.orc_tape_pool.argtypes = [ctypes.c_void_p, P(ctypes.c_int32)]
    L.orc_get_counters.argtypes = [ctypes.c_void_p, P(Counters)]
    L.orc_tape_digest.restype = ctypes.c_int32
    L.orc_tape_digest.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, P(ctypes.c_uint64)]
    L.orc_tiles_digest.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_size_t,
                                   ctypes.c_void_p, ctypes.c_void_p]
    L.orc_interval_op_n.argtypes = [ctypes.c_int32, ctypes.c_int32] + [ctypes.c_void_p] * 4 + \
                                   [ctypes.c_float] + [ctypes.c_void_p] * 3
    L.orc_float_op_n.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                 ctypes.c_float, ctypes.c_void_p]
    L.orc_deriv_op_n.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                 ctypes.c_float, ctypes.c_void_p]
    L.orc_selftest_rounding.restype = ctypes.c_int64
    L.orc_selftest_rounding.argtypes = [ctypes.c_int64, ctypes.c_uint64]
    L.orc_fmath_n.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    L.orc_effects.argtypes = [ctypes.c_int32, ctypes.c_int32] + [ctypes.c_void_p] * 4
    L.orc_effects_tables.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.orc_glibc_rand.argtypes = [ctypes.c_uint32, ctypes.c_int32, ctypes.c_void_p]
    _LIB = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class Frame:
    """Result of one oracle frame; arrays are copied out, the C frame is freed."""

    def __init__(self, tape, dim, size, mat, z=0.0, pool_clauses=0, threads=1, owner=None, rank=0,
                 brute=False, skip_normals=False, keep_pool=True, heatmap=False):
        L = lib()
        tape = np.ascontiguousarray(tape, dtype=np.uint64)
        mat = np.ascontiguousarray(mat, dtype=np.float32).reshape(-1)
        assert mat.size == (9 if dim == 2 else 16)
        own = None if owner is None else np.ascontiguousarray(owner, dtype=np.int32)
        # heatmap: True = as executed; "lower" / "upper" = bounds over the timing-dependent parts
        hbit = {False: 0, None: 0, True: 4, "lower": 8, "upper": 16}[heatmap]
        flags = (1 if brute else 0) | (2 if skip_normals else 0) | hbit
        f = L.orc_render(_ptr(tape), tape.size, dim, size, _ptr(mat), z, pool_clauses, threads,
                         _ptr(own), rank, flags)
        if not f:
            raise ValueError("orc_render rejected its arguments")
        try:
            n = ctypes.c_size_t()
            self.size, self.dim = size, dim
            self.filled = []
            for s in range(4):
                p = L.orc_filled(f, s, ctypes.byref(n))
                side = int(round(n.value ** 0.5))
                self.filled.append(np.ctypeslib.as_array(p, (n.value,)).copy().reshape(side, side))
            p = L.orc_normals(f, ctypes.byref(n))
            self.normals = np.ctypeslib.as_array(p, (n.value,)).copy().reshape(size, size)
            self.heatmap = None
            if heatmap:
                p = L.orc_heatmap(f, ctypes.byref(n))
                self.heatmap = np.ctypeslib.as_array(p, (n.value,)).copy().reshape(size, size)
            self.tiles = []
            for s in range(4):
                p = L.orc_tiles(f, s, ctypes.byref(n))
                if not p or n.value == 0:
                    self.tiles.append(np.zeros(0, dtype=TILE_DTYPE))
                else:
                    buf = (ctypes.c_char * (n.value * 12)).from_address(p)
                    self.tiles.append(np.frombuffer(buf, dtype=TILE_DTYPE).copy())
            ti = ctypes.c_int32()
            p = L.orc_tape_pool(f, ctypes.byref(ti))
            self.tape_index = ti.value
            self.pool = np.ctypeslib.as_array(p, (ti.value,)).copy() if keep_pool else None
            c = Counters()
            L.orc_get_counters(f, ctypes.byref(c))
            self.counters = c.as_dict()
        finally:
            L.orc_frame_free(f)

    @property
    def image(self):
        return self.filled[3]


def tiles_digest(pool, tiles):
    """(len, hash) of the shortened tape of every tile in `tiles` (TILE_DTYPE array)."""
    L = lib()
    pool = np.ascontiguousarray(pool, dtype=np.uint64)
    tiles = np.ascontiguousarray(tiles, dtype=TILE_DTYPE)
    ln = np.zeros(tiles.size, dtype=np.int32)
    hs = np.zeros(tiles.size, dtype=np.uint64)
    L.orc_tiles_digest(_ptr(pool), pool.size, _ptr(tiles), tiles.size, _ptr(ln), _ptr(hs))
    return ln, hs


def interval_op(op, a_lo, a_hi, b_lo=None, b_hi=None, imm=0.0):
    L = lib()
    a_lo = np.ascontiguousarray(a_lo, dtype=np.float32)
    a_hi = np.ascontiguousarray(a_hi, dtype=np.float32)
    b_lo = None if b_lo is None else np.ascontiguousarray(b_lo, dtype=np.float32)
    b_hi = None if b_hi is None else np.ascontiguousarray(b_hi, dtype=np.float32)
    lo = np.empty_like(a_lo)
    hi = np.empty_like(a_lo)
    ch = np.zeros(a_lo.size, dtype=np.int32)
    L.orc_interval_op_n(op, a_lo.size, _ptr(a_lo), _ptr(a_hi), _ptr(b_lo), _ptr(b_hi), imm, _ptr(lo),
                        _ptr(hi), _ptr(ch))
    return lo, hi, ch


def float_op(op, a, b=None, imm=0.0):
    L = lib()
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = None if b is None else np.ascontiguousarray(b, dtype=np.float32)
    out = np.empty_like(a)
    L.orc_float_op_n(op, a.size, _ptr(a), _ptr(b), imm, _ptr(out))
    return out


def deriv_op(op, a4, b4=None, imm=0.0):
    L = lib()
    a4 = np.ascontiguousarray(a4, dtype=np.float32).reshape(-1, 4)
    b4 = None if b4 is None else np.ascontiguousarray(b4, dtype=np.float32).reshape(-1, 4)
    out = np.empty_like(a4)
    L.orc_deriv_op_n(op, a4.shape[0], _ptr(a4), _ptr(b4), imm, _ptr(out))
    return out


FMATH = {"sin": 0, "cos": 1, "asin": 2, "acos": 3, "atan": 4, "exp": 5, "log": 6}


def fmath(name, x):
    L = lib()
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    L.orc_fmath_n(FMATH[name], x.size, _ptr(x), _ptr(out))
    return out


def selftest_rounding(n=200000, seed=1):
    return lib().orc_selftest_rounding(n, seed)


def effects(which, depth, normals):
    """mpr::Effects over a frame: which = "ssao" or "shaded" -> (image, tmp), (S, S) int32."""
    depth = np.ascontiguousarray(depth, dtype=np.int32)
    normals = np.ascontiguousarray(normals, dtype=np.uint32)
    S = depth.shape[0]
    image = np.zeros((S, S), dtype=np.int32)
    tmp = np.zeros((S, S), dtype=np.int32)
    lib().orc_effects({"ssao": 0, "shaded": 1}[which], S, _ptr(depth), _ptr(normals), _ptr(image), _ptr(tmp))
    return image, tmp


def effects_tables():
    kernel = np.zeros((64, 3), dtype=np.float32)
    rvecs = np.zeros((256, 3), dtype=np.float32)
    lib().orc_effects_tables(_ptr(kernel), _ptr(rvecs))
    return kernel, rvecs


def glibc_rand(n, seed=1):
    out = np.zeros(n, dtype=np.int32)
    lib().orc_glibc_rand(seed, n, _ptr(out))
    return out
