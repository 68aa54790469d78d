"""mpr_amd — Python face of the MI355X-native mpr hot path.

A thin ctypes layer over the C ABI (include/mpr_amd.h, built as mpr_amd/libmpr_amd.so) that
keeps the reference's names: ``Tape(tree)`` (inc/tape.hpp:24-30), ``Context(image_size_px)``
with ``render2D`` / ``render3D`` / ``render2D_brute`` (inc/context.hpp:38-49) and the public
members callers read after a render: ``stages[i].filled``, ``stages[i].tiles``, ``normals``,
``tape_data``, ``tape_index`` (benchmark/render_3d_table.cpp:59-69, circle.cpp:42-103).

There is no CPU fallback: every render call goes to the HIP kernels and raises ``MprError``
when the library or a device is missing.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("MPR_AMD_LIB", os.path.join(_HERE, "libmpr_amd.so"))   # override: A/B builds
MODELS_DIR = os.path.join(_ROOT, "fixtures", "models")

TILE_DTYPE = np.dtype([("position", "<i4"), ("tape", "<i4"), ("next", "<i4")])

CTX_TIMING = 1
CTX_COUNTERS = 2
CTX_PARANOID = 16
CTX_SERIAL_STAGES = 4
CTX_TIMING_FLOAT = 8

# libfive packed opcode numbers understood by the front end (include/mpr_amd.h mpr_tree_op)
T_SQUARE, T_SQRT, T_NEG, T_SIN, T_COS, T_ASIN, T_ACOS, T_ATAN, T_EXP, T_ABS, T_LOG = 7, 8, 9, 10, 11, 13, 14, 15, 16, 17, 18
T_ADD, T_MUL, T_MIN, T_MAX, T_SUB, T_DIV = 20, 21, 22, 23, 24, 25

# clause opcodes (include/mpr_clause.h)
OP_NAMES = ["INVALID", "JUMP", "SQUARE_LHS", "SQRT_LHS", "NEG_LHS", "SIN_LHS", "COS_LHS", "ASIN_LHS",
            "ACOS_LHS", "ATAN_LHS", "EXP_LHS", "ABS_LHS", "LOG_LHS", "ADD_LHS_IMM", "ADD_LHS_RHS",
            "MUL_LHS_IMM", "MUL_LHS_RHS", "MIN_LHS_IMM", "MIN_LHS_RHS", "MAX_LHS_IMM", "MAX_LHS_RHS",
            "SUB_LHS_IMM", "SUB_IMM_RHS", "SUB_LHS_RHS", "DIV_LHS_IMM", "DIV_IMM_RHS", "DIV_LHS_RHS",
            "COPY_IMM", "COPY_LHS", "COPY_RHS"]
OP = {n: i for i, n in enumerate(OP_NAMES)}


class MprError(RuntimeError):
    pass


class Counters(ctypes.Structure):
    _fields_ = [
        ("tiles_in", ctypes.c_int64 * 3),
        ("tiles_active", ctypes.c_int64 * 3),
        ("voxel_tiles", ctypes.c_int64),
        ("clauses_fwd", ctypes.c_int64),
        ("clauses_bwd", ctypes.c_int64),
        ("clauses_written", ctypes.c_int64),
        ("clauses_fwd_voxels", ctypes.c_int64),
        ("clauses_fwd_normals", ctypes.c_int64),
        ("lane_clauses", ctypes.c_int64),
        ("normal_pixels", ctypes.c_int64),
        ("tape_index", ctypes.c_int32),
        ("pool_overflowed", ctypes.c_int32),
        ("slots_exceeded", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]

    def as_dict(self):
        out = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            out[name] = list(v) if hasattr(v, "__len__") else v
        return out


class CtxOptions(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int32), ("image_size_px", ctypes.c_int32),
                ("pool_clauses", ctypes.c_int64), ("flags", ctypes.c_int32)]


_LIB = None


def build(force=False):
    """Compile libmpr_amd.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force or not os.path.exists(LIB_PATH) or not os.path.exists(os.path.join(_HERE, "libmpr_amd_test.so")):
        subprocess.check_call(["make", "-s", "-j4", "-C", os.path.join(_HERE, "csrc")] + (["-B"] if force else []))
    return LIB_PATH


TEST_LIB_PATH = os.environ.get("MPR_AMD_TEST_LIB", os.path.join(_HERE, "libmpr_amd_test.so"))


class _Libs:
    """libmpr_amd.so — and, for the entry points of include/mpr_amd_test.h (`mpr_test_*`, `mpr_debug_*`: single primitives and code
    generators for the parity suite, development counters), libmpr_amd_test.so: the same sources compiled with those entry points
    in (round 6: the product library exports none of them).  Loaded on first use."""

    def __init__(self, product):
        object.__setattr__(self, "_product", product)
        object.__setattr__(self, "_test", None)

    def __getattr__(self, name):
        if name.startswith("mpr_test_") or name.startswith("mpr_debug_"):
            if self._test is None:
                if not os.path.exists(TEST_LIB_PATH):
                    raise MprError("libmpr_amd_test.so is missing: run mpr_amd.build()")
                object.__setattr__(self, "_test", ctypes.CDLL(TEST_LIB_PATH))
            return getattr(self._test, name)
        return getattr(self._product, name)


def lib():
    """Load libmpr_amd.so; raises MprError when it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise MprError("libmpr_amd.so is missing: run mpr_amd.build() / __graft_entry__.build() first "
                       "(there is no CPU fallback)")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64; loading it first lets
    # libmpr_amd.so bind to the same copy, so that torch tensors (multi-GPU gather buffers) and
    # this library can share the device.  The other order leaves torch unable to see the GPU.
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = _Libs(ctypes.CDLL(LIB_PATH))
    vp, i32, f32, P = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float, ctypes.POINTER
    L.mpr_last_error.restype = ctypes.c_char_p
    L.mpr_version.restype = ctypes.c_char_p
    L.mpr_op_str.restype = ctypes.c_char_p
    L.mpr_op_str.argtypes = [ctypes.c_uint8]
    for name in ("mpr_tree_x", "mpr_tree_y", "mpr_tree_z"):
        getattr(L, name).argtypes = [P(vp)]
    L.mpr_tree_const.argtypes = [f32, P(vp)]
    L.mpr_tree_unary.argtypes = [ctypes.c_int, vp, P(vp)]
    L.mpr_tree_binary.argtypes = [ctypes.c_int, vp, vp, P(vp)]
    L.mpr_tree_remap.argtypes = [vp, vp, vp, vp, P(vp)]
    L.mpr_tree_from_frep.argtypes = [ctypes.c_char_p, ctypes.c_size_t, P(vp)]
    L.mpr_tree_from_frep_file.argtypes = [ctypes.c_char_p, P(vp)]
    L.mpr_tree_to_frep.argtypes = [vp, ctypes.c_char_p, ctypes.c_size_t, P(ctypes.c_size_t)]
    L.mpr_tree_size.argtypes = [vp, P(ctypes.c_size_t)]
    L.mpr_tree_free.argtypes = [vp]
    L.mpr_tape_from_tree.argtypes = [vp, P(vp)]
    L.mpr_tape_from_clauses.argtypes = [vp, i32, P(vp)]
    L.mpr_tape_length.argtypes = [vp]
    L.mpr_tape_data.argtypes = [vp]
    L.mpr_tape_data.restype = P(ctypes.c_uint64)
    for name in ("mpr_tape_num_slots", "mpr_tape_num_choices", "mpr_tape_flags"):
        getattr(L, name).argtypes = [vp]
    L.mpr_tape_frame_is_tame.argtypes = [vp, i32, vp, f32, vp]
    L.mpr_tape_free.argtypes = [vp]
    L.mpr_ctx_create.argtypes = [i32, i32, P(vp)]
    L.mpr_ctx_create_ex.argtypes = [P(CtxOptions), P(vp)]
    L.mpr_ctx_destroy.argtypes = [vp]
    L.mpr_render2d.argtypes = [vp, vp, vp, f32]
    L.mpr_render3d.argtypes = [vp, vp, vp]
    L.mpr_render2d_brute.argtypes = [vp, vp, vp, f32]
    L.mpr_render2d_heatmap.argtypes = [vp, vp, vp, f32, vp]
    L.mpr_render3d_heatmap.argtypes = [vp, vp, vp, vp]
    L.mpr_render2d_async.argtypes = [vp, vp, vp, f32]
    L.mpr_render3d_async.argtypes = [vp, vp, vp]
    L.mpr_ctx_sync.argtypes = [vp]
    L.mpr_render3d_part.argtypes = [vp, vp, vp, vp, i32]
    L.mpr_render2d_part.argtypes = [vp, vp, vp, f32, vp, i32]
    L.mpr_partition_columns.argtypes = [i32, vp, i32, vp]
    L.mpr_render3d_part_async.argtypes = [vp, vp, vp, vp, i32]
    L.mpr_render2d_part_async.argtypes = [vp, vp, vp, f32, vp, i32]
    L.mpr_gather_plan.argtypes = [vp, vp, i32, i32, i32, i32]
    L.mpr_pack_planned_async.argtypes = [vp, vp]
    L.mpr_unpack_planned_async.argtypes = [vp, vp]
    L.mpr_pack_columns.argtypes = [vp, vp, i32, i32, i32, vp]
    L.mpr_unpack_columns.argtypes = [vp, vp, i32, i32, i32, vp]
    L.mpr_read_filled.argtypes = [vp, i32, vp]
    L.mpr_read_normals.argtypes = [vp, vp]
    L.mpr_read_tiles.argtypes = [vp, i32, vp, ctypes.c_size_t, P(ctypes.c_size_t)]
    L.mpr_read_tape_pool.argtypes = [vp, vp, ctypes.c_size_t, P(i32)]
    L.mpr_dev_filled.argtypes = [vp, i32]
    L.mpr_dev_filled.restype = vp
    L.mpr_dev_normals.argtypes = [vp]
    L.mpr_dev_normals.restype = vp
    L.mpr_ctx_stream.argtypes = [vp]
    L.mpr_ctx_stream.restype = vp
    L.mpr_get_counters.argtypes = [vp, P(Counters)]
    L.mpr_get_timings.argtypes = [vp, P(ctypes.c_char_p), P(f32), i32, P(i32)]
    L.mpr_ctx_float_kernel.argtypes = [vp]
    L.mpr_ctx_normals_kernel.argtypes = [vp]
    L.mpr_ctx_last_stage_pushed.argtypes = [vp]
    L.mpr_ctx_skip0_vetoes.argtypes = [vp]
    L.mpr_ctx_skip0_vetoes.restype = ctypes.c_int64
    L.mpr_ctx_resident_bytes.argtypes = [vp]
    L.mpr_column_weights.argtypes = [vp, vp, i32, vp, f32, vp]
    L.mpr_ctx_resident_bytes.restype = ctypes.c_int64
    L.mpr_ctx_last_stage_pushed.restype = i32
    L.mpr_ctx_float_kernel.restype = ctypes.c_char_p
    L.mpr_ctx_normals_kernel.restype = ctypes.c_char_p
    L.mpr_ctx_frame_tiles.argtypes = [vp, vp]
    L.mpr_ctx_tile_stage_forms.argtypes = [vp]
    L.mpr_ctx_tile_stage_forms.restype = ctypes.c_char_p
    L.mpr_tape_schedule_info.argtypes = [vp, P(i32), P(i32), vp]
    L.mpr_compiled_create.argtypes = [i32, vp, P(vp)]
    L.mpr_compiled_destroy.argtypes = [vp]
    L.mpr_compiled_destroy.restype = None
    L.mpr_compiled_source.argtypes = [vp]
    L.mpr_compiled_source.restype = ctypes.c_char_p
    L.mpr_compiled_render2d.argtypes = [vp, i32, vp, f32, vp]
    L.mpr_dev_filled.argtypes = [vp, i32]
    L.mpr_dev_filled.restype = vp
    L.mpr_effects_create.argtypes = [i32, P(vp)]
    L.mpr_effects_destroy.argtypes = [vp]
    L.mpr_effects_destroy.restype = None
    L.mpr_effects_draw_ssao.argtypes = [vp, vp]
    L.mpr_effects_draw_shaded.argtypes = [vp, vp]
    L.mpr_effects_read_image.argtypes = [vp, vp]
    L.mpr_effects_read_tmp.argtypes = [vp, vp]
    L.mpr_effects_tables_get.argtypes = [vp, vp, vp]
    L.mpr_test_interval_op.argtypes = [i32, i32, i32, vp, vp, vp, vp, f32, vp, vp, vp]
    L.mpr_test_interval_op_asm.argtypes = [i32, i32, i32, i32, vp, vp, vp, vp, f32, vp, vp, vp]
    L.mpr_test_float_op.argtypes = [i32, i32, i32, vp, vp, f32, vp]
    L.mpr_test_float_op_asm.argtypes = [i32, i32, i32, i32, vp, vp, f32, vp]
    L.mpr_test_deriv_op.argtypes = [i32, i32, i32, vp, vp, f32, vp]
    L.mpr_test_sqrt_all.argtypes = [i32, ctypes.c_uint64, ctypes.c_uint64, vp, vp]
    L.mpr_test_jit_row.argtypes = [i32, i32, ctypes.c_uint32, ctypes.c_uint32, i32, vp, i32]
    L.mpr_test_tile_gen.argtypes = [vp, i32, i32, vp, i32]
    L.mpr_test_tile_gen.restype = ctypes.c_int
    L.mpr_test_voxel_gen.argtypes = [vp, i32, i32, vp, i32, vp]
    L.mpr_test_voxel_gen.restype = ctypes.c_int
    L.mpr_test_float_op_gen.argtypes = [i32, i32, i32, ctypes.c_uint64, ctypes.c_uint64, i32, vp, vp, f32, vp]
    _LIB = L
    return L


def _check(rc):
    if rc != 0:
        raise MprError("mpr_amd error %d: %s" % (rc, lib().mpr_last_error().decode()))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def colmajor(mat, n):
    """numpy (n x n, indexed [row, col]) -> column-major float32[n*n] as Eigen stores it."""
    m = np.asarray(mat, dtype=np.float32)
    if m.shape != (n, n):
        raise ValueError("expected a %dx%d matrix" % (n, n))
    return np.ascontiguousarray(m.T).reshape(-1)


# ---------------------------------------------------------------------------------------------
class Tree:
    """Stand-in for libfive::Tree with the operators the reference's call sites use."""

    def __init__(self, value=None, _handle=None):
        L = lib()
        if _handle is not None:
            self._h = _handle
        else:
            h = ctypes.c_void_p()
            _check(L.mpr_tree_const(float(value), ctypes.byref(h)))
            self._h = h

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().mpr_tree_free(self._h)
        except Exception:
            pass

    @staticmethod
    def _axis(fn):
        h = ctypes.c_void_p()
        _check(getattr(lib(), fn)(ctypes.byref(h)))
        return Tree(_handle=h)

    @staticmethod
    def X():
        return Tree._axis("mpr_tree_x")

    @staticmethod
    def Y():
        return Tree._axis("mpr_tree_y")

    @staticmethod
    def Z():
        return Tree._axis("mpr_tree_z")

    @staticmethod
    def _lift(v):
        return v if isinstance(v, Tree) else Tree(float(v))

    @staticmethod
    def unary(op, a):
        h = ctypes.c_void_p()
        _check(lib().mpr_tree_unary(op, Tree._lift(a)._h, ctypes.byref(h)))
        return Tree(_handle=h)

    @staticmethod
    def binary(op, a, b):
        a, b = Tree._lift(a), Tree._lift(b)
        h = ctypes.c_void_p()
        _check(lib().mpr_tree_binary(op, a._h, b._h, ctypes.byref(h)))
        return Tree(_handle=h)

    def __add__(self, o): return Tree.binary(T_ADD, self, o)
    def __radd__(self, o): return Tree.binary(T_ADD, o, self)
    def __sub__(self, o): return Tree.binary(T_SUB, self, o)
    def __rsub__(self, o): return Tree.binary(T_SUB, o, self)
    def __mul__(self, o): return Tree.binary(T_MUL, self, o)
    def __rmul__(self, o): return Tree.binary(T_MUL, o, self)
    def __truediv__(self, o): return Tree.binary(T_DIV, self, o)
    def __rtruediv__(self, o): return Tree.binary(T_DIV, o, self)
    def __neg__(self): return Tree.unary(T_NEG, self)

    def remap(self, x, y, z):
        h = ctypes.c_void_p()
        _check(lib().mpr_tree_remap(self._h, Tree._lift(x)._h, Tree._lift(y)._h, Tree._lift(z)._h, ctypes.byref(h)))
        return Tree(_handle=h)

    def size(self):
        n = ctypes.c_size_t()
        _check(lib().mpr_tree_size(self._h, ctypes.byref(n)))
        return n.value

    @staticmethod
    def from_frep(path_or_bytes):
        h = ctypes.c_void_p()
        if isinstance(path_or_bytes, (bytes, bytearray)):
            _check(lib().mpr_tree_from_frep(bytes(path_or_bytes), len(path_or_bytes), ctypes.byref(h)))
        else:
            _check(lib().mpr_tree_from_frep_file(str(path_or_bytes).encode(), ctypes.byref(h)))
        return Tree(_handle=h)

    def to_frep(self):
        n = ctypes.c_size_t()
        _check(lib().mpr_tree_to_frep(self._h, None, 0, ctypes.byref(n)))
        buf = ctypes.create_string_buffer(n.value)
        _check(lib().mpr_tree_to_frep(self._h, buf, n.value, ctypes.byref(n)))
        return buf.raw[:n.value]


def tmin(a, b): return Tree.binary(T_MIN, a, b)
def tmax(a, b): return Tree.binary(T_MAX, a, b)
def sqrt(a): return Tree.unary(T_SQRT, a)
def square(a): return Tree.unary(T_SQUARE, a)
def tabs(a): return Tree.unary(T_ABS, a)
def sin(a): return Tree.unary(T_SIN, a)
def cos(a): return Tree.unary(T_COS, a)
def asin(a): return Tree.unary(T_ASIN, a)
def acos(a): return Tree.unary(T_ACOS, a)
def atan(a): return Tree.unary(T_ATAN, a)
def exp(a): return Tree.unary(T_EXP, a)
def log(a): return Tree.unary(T_LOG, a)


def model(name):
    """One of the reference's benchmark models (benchmark/files/*.frep, copied to fixtures/models)."""
    return Tree.from_frep(os.path.join(MODELS_DIR, name + ".frep"))


# ---------------------------------------------------------------------------------------------
class Tape:
    """mpr::Tape — a flat tape of 64-bit clauses (src/tape.cpp)."""

    def __init__(self, source):
        L = lib()
        h = ctypes.c_void_p()
        if isinstance(source, Tree):
            _check(L.mpr_tape_from_tree(source._h, ctypes.byref(h)))
        else:
            arr = np.ascontiguousarray(source, dtype=np.uint64)
            _check(L.mpr_tape_from_clauses(_ptr(arr), arr.size, ctypes.byref(h)))
        self._h = h

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().mpr_tape_free(self._h)
        except Exception:
            pass

    @property
    def length(self):
        return lib().mpr_tape_length(self._h)

    @property
    def data(self):
        n = self.length
        return np.ctypeslib.as_array(lib().mpr_tape_data(self._h), (n,)).copy()

    def schedule_levels(self):
        """Dependency level of every body clause (tape_schedule.hpp) -> (levels, nlevels, max_width)."""
        nl, mw = ctypes.c_int32(), ctypes.c_int32()
        lv = np.zeros(max(self.length - 2, 0), dtype=np.int32)
        _check(lib().mpr_tape_schedule_info(self._h, ctypes.byref(nl), ctypes.byref(mw), _ptr(lv)))
        return lv, nl.value, mw.value

    @property
    def num_slots(self):
        return lib().mpr_tape_num_slots(self._h)

    @property
    def num_choices(self):
        return lib().mpr_tape_num_choices(self._h)

    @property
    def flags(self):
        return lib().mpr_tape_flags(self._h)

    def frame_is_tame(self, mat, dim=3, z=0.0, trace=False):
        """Does every interval operation stay, over the whole view of a frame with this matrix, where the reference's interval
        routines are inclusion-isotone (csrc/frame_domain.hpp)?  trace=True: (verdict, enclosures[length, 2] of the clauses' results)."""
        m = np.ascontiguousarray(colmajor(mat, dim + 1), dtype=np.float32)
        tr = np.zeros((self.length, 2), dtype=np.float64) if trace else None
        rc = lib().mpr_tape_frame_is_tame(self._h, dim, _ptr(m), float(z), _ptr(tr) if trace else None)
        if rc < 0:
            _check(rc)
        return (bool(rc), tr) if trace else bool(rc)


def clause(op, out=0, lhs=0, rhs=0, imm=0.0):
    """Pack one clause (inc/clause.hpp:18-23)."""
    bits = int(np.float32(imm).view(np.uint32))
    return np.uint64((op & 0xFF) | ((out & 0xFF) << 8) | ((lhs & 0xFF) << 16) | ((rhs & 0xFF) << 24) | (bits << 32))


def decode(clauses):
    """-> list of (opname, out, lhs, rhs, imm) for a clause array."""
    out = []
    for c in np.asarray(clauses, dtype=np.uint64):
        c = int(c)
        imm = float(np.uint32(c >> 32).view(np.float32))
        out.append((OP_NAMES[c & 0xFF] if (c & 0xFF) < len(OP_NAMES) else "?", (c >> 8) & 0xFF, (c >> 16) & 0xFF,
                    (c >> 24) & 0xFF, imm))
    return out


class _Stage:
    def __init__(self, ctx, index):
        self._ctx, self._i = ctx, index

    @property
    def filled(self):
        """Tiles::filled — (side x side) int32, [y, x]."""
        c = self._ctx
        side = c.image_size_px // (64 >> (2 * self._i))
        a = np.empty(side * side, dtype=np.int32)
        _check(lib().mpr_read_filled(c._h, self._i, _ptr(a)))
        return a.reshape(side, side)

    @property
    def tiles(self):
        """Tiles::tiles — structured array (position, tape, next) of the last frame's list."""
        c = self._ctx
        n = ctypes.c_size_t()
        _check(lib().mpr_read_tiles(c._h, self._i, None, 0, ctypes.byref(n)))
        a = np.zeros(n.value, dtype=TILE_DTYPE)
        if n.value:
            _check(lib().mpr_read_tiles(c._h, self._i, _ptr(a), n.value, ctypes.byref(n)))
        return a

    @property
    def tile_array_size(self):
        n = ctypes.c_size_t()
        _check(lib().mpr_read_tiles(self._ctx._h, self._i, None, 0, ctypes.byref(n)))
        return n.value


class Context:
    """mpr::Context — owns the device buffers and renders frames (src/context.cu:1136-1508)."""

    def __init__(self, image_size_px, device=0, pool_clauses=0, flags=0):
        o = CtxOptions(device, image_size_px, pool_clauses, flags)
        h = ctypes.c_void_p()
        _check(lib().mpr_ctx_create_ex(ctypes.byref(o), ctypes.byref(h)))
        self._h = h
        self.image_size_px = image_size_px
        self.device = device
        self.stages = [_Stage(self, i) for i in range(4)]

    def close(self):
        if getattr(self, "_h", None):
            lib().mpr_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render2D(self, tape, mat=None, z=0.0, blocking=True):
        m = colmajor(np.eye(3) if mat is None else mat, 3)
        fn = lib().mpr_render2d if blocking else lib().mpr_render2d_async
        _check(fn(self._h, tape._h, _ptr(m), z))

    def render3D(self, tape, mat=None, blocking=True):
        m = colmajor(np.eye(4) if mat is None else mat, 4)
        fn = lib().mpr_render3d if blocking else lib().mpr_render3d_async
        _check(fn(self._h, tape._h, _ptr(m)))

    def render2D_brute(self, tape, mat=None, z=0.0):
        m = colmajor(np.eye(3) if mat is None else mat, 3)
        _check(lib().mpr_render2d_brute(self._h, tape._h, _ptr(m), z))

    def render2D_heatmap(self, tape, mat=None, z=0.0):
        """Context::render2D_heatmap (inc/context.hpp:51-56): renders, and returns work per pixel [y, x]."""
        m = colmajor(np.eye(3) if mat is None else mat, 3)
        heat = np.empty((self.image_size_px, self.image_size_px), dtype=np.float32)
        _check(lib().mpr_render2d_heatmap(self._h, tape._h, _ptr(m), z, _ptr(heat)))
        return heat

    def render3D_heatmap(self, tape, mat=None):
        """Context::render3D_heatmap (inc/context.hpp:57-58)."""
        m = colmajor(np.eye(4) if mat is None else mat, 4)
        heat = np.empty((self.image_size_px, self.image_size_px), dtype=np.float32)
        _check(lib().mpr_render3d_heatmap(self._h, tape._h, _ptr(m), _ptr(heat)))
        return heat

    def column_weights(self, tape, mat, z=0.0, dim=3):
        """Ambiguous first-stage tiles per 64 x 64 column (mpr_column_weights): the work proxy of the column deal."""
        m = colmajor(mat, dim + 1)
        w = np.zeros((self.image_size_px // 64) ** 2, dtype=np.float32)
        _check(lib().mpr_column_weights(self._h, tape._h, dim, _ptr(m), z, _ptr(w)))
        return w

    def render3D_part(self, tape, mat, owner, rank, blocking=True):
        m = colmajor(mat, 4)
        own = np.ascontiguousarray(owner, dtype=np.int32)
        fn = lib().mpr_render3d_part if blocking else lib().mpr_render3d_part_async
        _check(fn(self._h, tape._h, _ptr(m), _ptr(own), rank))

    def render2D_part(self, tape, mat, z, owner, rank, blocking=True):
        m = colmajor(mat, 3)
        own = np.ascontiguousarray(owner, dtype=np.int32)
        fn = lib().mpr_render2d_part if blocking else lib().mpr_render2d_part_async
        _check(fn(self._h, tape._h, _ptr(m), z, _ptr(own), rank))

    def gather_plan(self, owner, rank, world, capacity_cols, with_normals):
        """Make the ownership table resident: afterwards pack_planned / unpack_planned need no
        uploads and no host synchronisation (the steady state of the multi-GPU loop)."""
        own = np.ascontiguousarray(owner, dtype=np.int32)
        _check(lib().mpr_gather_plan(self._h, _ptr(own), rank, world, capacity_cols, int(with_normals)))

    def pack_planned(self, dev_ptr):
        _check(lib().mpr_pack_planned_async(self._h, ctypes.c_void_p(dev_ptr)))

    def unpack_planned(self, dev_ptr):
        _check(lib().mpr_unpack_planned_async(self._h, ctypes.c_void_p(dev_ptr)))

    @property
    def stream(self):
        """The context's hipStream_t as an integer (e.g. for torch.cuda.ExternalStream)."""
        return lib().mpr_ctx_stream(self._h)

    def pack_columns(self, owner, rank, capacity_cols, with_normals, dev_ptr):
        own = np.ascontiguousarray(owner, dtype=np.int32)
        _check(lib().mpr_pack_columns(self._h, _ptr(own), rank, capacity_cols, int(with_normals), ctypes.c_void_p(dev_ptr)))

    def unpack_columns(self, owner, rank, capacity_cols, with_normals, dev_ptr):
        own = np.ascontiguousarray(owner, dtype=np.int32)
        _check(lib().mpr_unpack_columns(self._h, _ptr(own), rank, capacity_cols, int(with_normals), ctypes.c_void_p(dev_ptr)))

    def sync(self):
        _check(lib().mpr_ctx_sync(self._h))

    @property
    def image(self):
        """stages[3].filled: the S x S occupancy image (2-D) or heightmap (3-D)."""
        return self.stages[3].filled

    @property
    def normals(self):
        a = np.empty(self.image_size_px * self.image_size_px, dtype=np.uint32)
        _check(lib().mpr_read_normals(self._h, _ptr(a)))
        return a.reshape(self.image_size_px, self.image_size_px)

    @property
    def tape_index(self):
        ti = ctypes.c_int32()
        _check(lib().mpr_read_tape_pool(self._h, None, 0, ctypes.byref(ti)))
        return ti.value

    @property
    def tape_data(self):
        ti = ctypes.c_int32()
        _check(lib().mpr_read_tape_pool(self._h, None, 0, ctypes.byref(ti)))
        a = np.zeros(max(ti.value, 0), dtype=np.uint64)
        if a.size:
            _check(lib().mpr_read_tape_pool(self._h, _ptr(a), a.size, ctypes.byref(ti)))
        return a

    def counters(self):
        c = Counters()
        _check(lib().mpr_get_counters(self._h, ctypes.byref(c)))
        return c.as_dict()

    def timings(self):
        """[(kernel name, ms)] of the last frame (needs CTX_TIMING)."""
        cap = 64
        names = (ctypes.c_char_p * cap)()
        ms = (ctypes.c_float * cap)()
        n = ctypes.c_int32()
        _check(lib().mpr_get_timings(self._h, names, ms, cap, ctypes.byref(n)))
        return [(names[i].decode(), float(ms[i])) for i in range(n.value)]

    def normals_kernel(self):
        """Name of the kernel the last frame's normals pass ran as (mpr_ctx_normals_kernel)."""
        return lib().mpr_ctx_normals_kernel(self._h).decode()

    def tile_stage_forms(self):
        """The form each tile stage of the last frame took (mpr_ctx_tile_stage_forms), e.g. "1:gen+bwd+records 2:gen/parent+guards"."""
        return lib().mpr_ctx_tile_stage_forms(self._h).decode()

    def frame_tiles(self):
        """Tiles of the last frame as it ran (mpr_ctx_frame_tiles): (evaluated per stage [3], left ambiguous [3], smallest tiles)."""
        a = np.zeros(7, dtype=np.int64)
        _check(lib().mpr_ctx_frame_tiles(self._h, _ptr(a)))
        return a[:3].tolist(), a[3:6].tolist(), int(a[6])

    def tiles_walked(self):
        """Development: tiles the last frame's float pass walked (context made with MPR_DEBUG_WALKED=1; k_eval_voxels_gen: the tiles it did not find hidden), or -1."""
        f = lib().mpr_debug_tiles_walked
        f.restype = ctypes.c_longlong
        f.argtypes = [ctypes.c_void_p]
        return int(f(self._h))

    def float_kernel(self):
        """Name of the kernel the last frame's float pass ran as (mpr_ctx_float_kernel)."""
        return lib().mpr_ctx_float_kernel(self._h).decode()

    def resident_bytes(self):
        """Device memory the context holds right now (mpr_ctx_resident_bytes)."""
        return int(lib().mpr_ctx_resident_bytes(self._h))

    def last_stage_pushed(self):
        """False when the last frame's last tile stage pushed no per-tile tapes (mpr_ctx_last_stage_pushed)."""
        return bool(lib().mpr_ctx_last_stage_pushed(self._h))

    def skip0_vetoes(self):
        """Frames whose shortcut past the 64^3 tiles failed its verification against those tiles and were rendered again from them."""
        return int(lib().mpr_ctx_skip0_vetoes(self._h))

    def paranoid_stats(self):
        """(frames rendered, frames rendered again the reference's way, cells in which the two renderings differed): CTX_PARANOID."""
        out = (ctypes.c_int64 * 3)()
        f = lib().mpr_ctx_paranoid_stats
        f.restype = ctypes.c_int
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _check(f(self._h, out))
        return int(out[0]), int(out[1]), int(out[2])

    def dev_filled(self, stage=3):
        return lib().mpr_dev_filled(self._h, stage)

    def dev_normals(self):
        return lib().mpr_dev_normals(self._h)


class CompiledTape:
    """The compiled-expression baseline (benchmark/dump_tape.cpp + brute.cu of the reference): the
    tape as straight-line HIP source compiled at run time, evaluated for every pixel."""

    def __init__(self, tape, device=0):
        h = ctypes.c_void_p()
        _check(lib().mpr_compiled_create(device, tape._h, ctypes.byref(h)))
        self._h = h

    def close(self):
        if self._h:
            lib().mpr_compiled_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def source(self):
        return lib().mpr_compiled_source(self._h).decode()

    def render2D(self, ctx, mat=None, z=0.0):
        """Evaluates into ctx.image (the context's stages[3].filled)."""
        m = colmajor(np.eye(3) if mat is None else mat, 3)
        ctx.sync()
        _check(lib().mpr_compiled_render2d(self._h, ctx.image_size_px, _ptr(m), z, lib().mpr_dev_filled(ctx._h, 3)))


class Effects:
    """mpr::Effects (inc/effects.hpp:21-37): SSAO and shading over a context's last render3D."""

    def __init__(self, device=0):
        h = ctypes.c_void_p()
        _check(lib().mpr_effects_create(device, ctypes.byref(h)))
        self._h = h
        self._size = 0

    def close(self):
        if self._h:
            lib().mpr_effects_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def drawSSAO(self, ctx):
        _check(lib().mpr_effects_draw_ssao(self._h, ctx._h))
        self._size = ctx.image_size_px

    def drawShaded(self, ctx):
        _check(lib().mpr_effects_draw_shaded(self._h, ctx._h))
        self._size = ctx.image_size_px

    def _read(self, fn):
        out = np.empty((self._size, self._size), dtype=np.int32)
        _check(fn(self._h, _ptr(out)))
        return out

    @property
    def image(self):
        return self._read(lib().mpr_effects_read_image)

    @property
    def tmp(self):
        return self._read(lib().mpr_effects_read_tmp)

    def tables(self):
        kernel = np.zeros((64, 3), dtype=np.float32)
        rvecs = np.zeros((256, 3), dtype=np.float32)
        _check(lib().mpr_effects_tables_get(self._h, _ptr(kernel), _ptr(rvecs)))
        return kernel, rvecs


def partition_columns(columns, nranks, weights=None):
    """Deterministic column -> rank deal (LPT when weights are given)."""
    owner = np.zeros(columns, dtype=np.int32)
    w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float32)
    _check(lib().mpr_partition_columns(columns, _ptr(w), nranks, _ptr(owner)))
    return owner


# ---- device primitive tests (parity fuzzing) ----
def dev_interval_op(op, a_lo, a_hi, b_lo=None, b_hi=None, imm=0.0, device=0, asm=False, variant=0):
    """One interval clause on the device: the compiled interval_clause, or (asm=True) the tile
    stages' assembly forward walk over a short tape (variant 1 / 2: lhs / rhs forwarded)."""
    a_lo = np.ascontiguousarray(a_lo, dtype=np.float32)
    a_hi = np.ascontiguousarray(a_hi, dtype=np.float32)
    b_lo = None if b_lo is None else np.ascontiguousarray(b_lo, dtype=np.float32)
    b_hi = None if b_hi is None else np.ascontiguousarray(b_hi, dtype=np.float32)
    lo, hi = np.empty_like(a_lo), np.empty_like(a_lo)
    ch = np.zeros(a_lo.size, dtype=np.int32)
    if asm:
        _check(lib().mpr_test_interval_op_asm(device, op, variant, a_lo.size, _ptr(a_lo), _ptr(a_hi), _ptr(b_lo),
                                              _ptr(b_hi), imm, _ptr(lo), _ptr(hi), _ptr(ch)))
    else:
        _check(lib().mpr_test_interval_op(device, op, a_lo.size, _ptr(a_lo), _ptr(a_hi), _ptr(b_lo), _ptr(b_hi), imm,
                                          _ptr(lo), _ptr(hi), _ptr(ch)))
    return lo, hi, ch


def dev_float_op(op, a, b=None, imm=0.0, device=0, asm=False, variant=0):
    """One float clause on the device: the compiled float_clause, or (asm=True) the float pass's
    assembly interpreter run over a short tape (variant 1 / 2: lhs / rhs is the previous clause's
    result, which the interpreter forwards in a register)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = None if b is None else np.ascontiguousarray(b, dtype=np.float32)
    out = np.empty_like(a)
    if asm:
        _check(lib().mpr_test_float_op_asm(device, op, variant, a.size, _ptr(a), _ptr(b), imm, _ptr(out)))
    else:
        _check(lib().mpr_test_float_op(device, op, a.size, _ptr(a), _ptr(b), imm, _ptr(out)))
    return out


def dev_float_op_gen(op, a, b=None, imm=0.0, device=0, variant=0, dl=0, dr=0):
    """One float clause through the host-generated float walk (csrc/voxel_gen.cpp) on the device; variant 1 / 2: the result
    overwrites its lhs / rhs operand; dl / dr: the tile's decisions (bit 0 = this clause)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = None if b is None else np.ascontiguousarray(b, dtype=np.float32)
    out = np.empty_like(a)
    _check(lib().mpr_test_float_op_gen(device, op, variant, dl, dr, a.size, _ptr(a), _ptr(b), imm, _ptr(out)))
    return out


def dev_interval_gen_op(op, a_lo, a_hi, b_lo=None, b_hi=None, imm=0.0, loose=False, device=0):
    """One interval clause through the tile stages' SCHEDULED code (csrc/interval_gen.cpp) on the device, exact or loose:
    (lo, hi, the lanes' choice 0 / 1 / 2, lanes whose loose walk asks for the exact one)."""
    a_lo = np.ascontiguousarray(a_lo, dtype=np.float32)
    a_hi = np.ascontiguousarray(a_hi, dtype=np.float32)
    b_lo = None if b_lo is None else np.ascontiguousarray(b_lo, dtype=np.float32)
    b_hi = None if b_hi is None else np.ascontiguousarray(b_hi, dtype=np.float32)
    lo, hi = np.empty_like(a_lo), np.empty_like(a_lo)
    ch = np.zeros(a_lo.size, dtype=np.int32)
    asks = np.zeros(a_lo.size, dtype=np.int32)
    f = lib().mpr_test_interval_gen_op
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                  ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    _check(f(device, op, int(loose), a_lo.size, _ptr(a_lo), _ptr(a_hi), _ptr(b_lo), _ptr(b_hi), imm, _ptr(lo), _ptr(hi), _ptr(ch), _ptr(asks)))
    return lo, hi, ch, asks


def dev_loose_gen(op, imm=0.0, other=(0.0, 0.0), x_is_rhs=False, first=0, count=1 << 32, device=0):
    """The LOOSE scheduled code of one clause on the bit patterns [first, first + count) — as [x, x] and as the interval between x
    and a scrambled copy of its bits; the other operand is the interval `other` — against the exact routine's enclosure:
    dict(bad, example, tested, asked_for_exact, widest in 2^-24 of the value)."""
    out = (ctypes.c_uint64 * 5)()
    f = lib().mpr_test_loose_gen
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_int32, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
    _check(f(device, op, imm, other[0], other[1], int(x_is_rhs), first, count, out))
    return dict(bad=int(out[0]), example=int(out[1]), tested=int(out[2]), asked_for_exact=int(out[3]), widest=int(out[4]))


def dev_tight_trig(is_sin, first=0, count=1 << 32, device=0):
    """The TIGHT code of one SIN_LHS / COS_LHS clause on the bit patterns [first, first + count) against the float pass's sinf / cosf:
    dict(bad, example, tested, asked_for_exact, hw_error (the instruction's largest error for |x| <= 1024), narrow, wide)."""
    out = (ctypes.c_uint64 * 7)()
    f = lib().mpr_test_tight_trig
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
    _check(f(device, int(is_sin), first, count, out))
    return dict(bad=int(out[0]), example=int(out[1]), tested=int(out[2]), asked_for_exact=int(out[3]), hw_error=int(out[4]) * 2.0 ** -40,
                narrow=int(out[5]), wide=int(out[6]))


def dev_float_gen_all(op, imm=0.0, first=0, count=1 << 32, device=0):
    """One clause through the host-generated float walk on the bit patterns [first, first + count) against float_clause:
    dict(tested, bad, example)."""
    out = (ctypes.c_uint64 * 3)()
    f = lib().mpr_test_float_gen_all
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_float, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
    _check(f(device, op, imm, first, count, out))
    return dict(tested=int(out[0]), bad=int(out[1]), example=int(out[2]))


def dev_float_in_enclosure(op, imm=0.0, first=0, count=1 << 32, device=0):
    """The float pass's f(x) against the exact interval routine's enclosure of [x, x] on the bit patterns [first, first + count):
    dict(tested, outside, nan_mismatch, example_outside, farthest (units of the end's last place), example_nan)."""
    out = (ctypes.c_uint64 * 6)()
    f = lib().mpr_test_float_in_enclosure
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_float, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
    _check(f(device, op, imm, first, count, out))
    return dict(tested=int(out[0]), outside=int(out[1]), nan_mismatch=int(out[2]), example_outside=int(out[3]), farthest=int(out[4]), example_nan=int(out[5]))


def dev_sqrt_all(first=0, count=1 << 32, device=0):
    """The float pass's square-root routine on the bit patterns [first, first + count): (mismatches against the correctly
    rounded root, one offending bit pattern)."""
    bad = ctypes.c_uint64(0)
    ex = ctypes.c_uint32(0)
    _check(lib().mpr_test_sqrt_all(device, first, count, ctypes.byref(bad), ctypes.byref(ex)))
    return int(bad.value), int(ex.value)


def dev_deriv_op(op, a4, b4=None, imm=0.0, device=0):
    a4 = np.ascontiguousarray(a4, dtype=np.float32).reshape(-1, 4)
    b4 = None if b4 is None else np.ascontiguousarray(b4, dtype=np.float32).reshape(-1, 4)
    out = np.empty_like(a4)
    _check(lib().mpr_test_deriv_op(device, op, a4.shape[0], _ptr(a4), _ptr(b4), imm, _ptr(out)))
    return out
