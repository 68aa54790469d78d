"""The tile stages' BACKWARD walks and the normals pass's Deriv walk as machine code generated on the host (csrc/tile_gen.cpp; the
forward walk: csrc/interval_gen.cpp, tests/test_interval_gen.py) —
one short instruction sequence per clause, slots and immediates in the instruction words).  Here that code is
disassembled with the ROCm assembler's llvm-mc and compared with the instructions it is meant to be — no GPU needed;
tests/test_gpu_render.py runs it against the interpreter and the oracle."""
import ctypes
import os
import struct
import subprocess

import numpy as np
import pytest

LLVM_MC = "/opt/rocm/lib/llvm/bin/llvm-mc"
PI, MINUS_PI = 0x40490FDB, 0xC0490FDB
pytestmark = pytest.mark.skipif(not os.path.exists(LLVM_MC), reason="llvm-mc not found")


def clause(op, out=0, lhs=0, rhs=0, imm=0):
    return op | out << 8 | lhs << 16 | rhs << 24 | imm << 32


def generated(mpr, words, which):
    arr = np.array(words, dtype=np.uint64)
    buf = (ctypes.c_uint32 * 65536)()
    n = mpr.lib().mpr_test_tile_gen(arr.ctypes.data, len(arr), which, buf, 65536)
    if n < 0:
        return None
    text = ",".join("0x%02x" % b for d in buf[:n] for b in struct.pack("<I", d))
    r = subprocess.run([LLVM_MC, "-arch=amdgcn", "-mcpu=gfx950", "-disassemble"], input=text.encode(), capture_output=True, check=True)
    assert not r.stderr.strip(), r.stderr.decode()
    return [" ".join(l.split()) for l in r.stdout.decode().splitlines() if l.strip() and not l.strip().startswith(".")]


def one(mpr, c, which=0):
    """the code of a tape with the single clause c (x, y, z in slots 1, 2, 3), without the epilogue"""
    code = generated(mpr, [clause(0, 1, 2, 3), c, clause(0, (c >> 8) & 0xFF)], which)
    assert code[-1] == "s_setpc_b64 s[38:39]"
    return code[:-1]


def test_backward_rows(mpr):
    OP = mpr.OP
    take = ["v_cmp_eq_u32_e32 vcc, v61, v62", "s_cbranch_vccz 1", "s_swappc_b64 s[36:37], s[62:63]"]
    store = ["v_lshlrev_b32_e32 v44, 3, v61", "s_mov_b64 exec, vcc", "global_store_dwordx2 v44, v[46:47], s[76:77]", "s_mov_b64 exec, -1"]
    # v60: active slots, v61: pool index of the last word written, v62: first index of the chunk
    assert one(mpr, clause(OP["ADD_LHS_RHS"], 4, 1, 2), 1) == [
        "v_bfe_u32 v32, v60, 4, 1", "v_sub_u32_e32 v61, v61, v32"] + take + [
        "v_and_b32_e32 v60, 0xffffffef, v60", "v_lshl_or_b32 v60, v32, 1, v60", "v_lshl_or_b32 v60, v32, 2, v60",
        "v_mov_b32_e32 v46, 0x201040e", "v_cmp_ne_u32_e32 vcc, 0, v32"] + store
    # an operand that is the out slot leaves its bit as it is; an immediate goes into the upper half of the word
    assert one(mpr, clause(OP["SUB_LHS_IMM"], 4, 4, 0, PI), 1) == [
        "v_bfe_u32 v32, v60, 4, 1", "v_sub_u32_e32 v61, v61, v32"] + take + [
        "v_mov_b32_e32 v46, 0x40415", "v_mov_b32_e32 v47, 0x40490fdb", "v_cmp_ne_u32_e32 vcc, 0, v32"] + store
    # min: a lane that chose a side keeps that operand and stores a COPY
    code = one(mpr, clause(OP["MIN_LHS_RHS"], 4, 1, 2), 1)
    assert code[:3] == ["v_bfe_u32 v32, v60, 4, 1", "v_bfe_u32 v33, v56, 0, 1", "v_bfe_u32 v34, v58, 0, 1"]
    assert "v_mov_b32_e32 v39, 0x201041c" in code and "v_mov_b32_e32 v40, 0x201041d" in code
    assert code[-4:] == store
    # ... or nothing, when the copy would be of a slot onto itself
    code = one(mpr, clause(OP["MAX_LHS_IMM"], 4, 4, 0, PI), 1)
    assert code[7] == "v_lshl_or_b32 v41, v36, 0, v41"          # this lane's tape keeps min / max clause 0
    assert code[8:11] == ["v_xor_b32_e32 v35, 1, v33", "v_and_b32_e32 v35, v35, v32", "v_sub_u32_e32 v61, v61, v35"]
    assert "v_mov_b32_e32 v40, 0x4041b" in code                     # COPY_IMM


def test_backward_rows_for_tapes_of_up_to_96_slots(mpr, tapes):
    """tile_gen_build_big_backward (a first stage's tape pushing for tapes the other generators do not take): backward_rows with the
    active slots in v60 / v64 / v65 and the lanes' choices read from the forward walk's 16-byte records in LDS."""
    OP = mpr.OP
    take = ["v_cmp_eq_u32_e32 vcc, v61, v62", "s_cbranch_vccz 1", "s_swappc_b64 s[36:37], s[62:63]"]
    store = ["v_lshlrev_b32_e32 v44, 3, v61", "s_mov_b64 exec, vcc", "global_store_dwordx2 v44, v[46:47], s[76:77]", "s_mov_b64 exec, -1"]
    # out slot 70 (bit 6 of v65), operands 33 (bit 1 of v64) and 2 (v60)
    assert one(mpr, clause(OP["ADD_LHS_RHS"], 70, 33, 2), 6) == [
        "v_bfe_u32 v32, v65, 6, 1", "v_sub_u32_e32 v61, v61, v32"] + take + [
        "v_and_b32_e32 v65, 0xffffffbf, v65", "v_lshl_or_b32 v64, v32, 1, v64", "v_lshl_or_b32 v60, v32, 2, v60",
        "v_mov_b32_e32 v46, 0x221460e", "v_cmp_ne_u32_e32 vcc, 0, v32"] + store
    # min: the clause's record was asked for when the walk began (the only one: nothing to ask for behind it), every lane takes the word of
    # its half and its own bit
    code = generated(mpr, [clause(0, 1, 2, 3), clause(OP["MIN_LHS_RHS"], 40, 1, 2), clause(0, 40)], 6)
    assert code[0] == "ds_read_b128 v[66:69], v75" and code[-1] == "s_setpc_b64 s[38:39]"
    assert code[1:7] == ["v_bfe_u32 v32, v64, 8, 1", "s_waitcnt lgkmcnt(0)", "v_cndmask_b32_e64 v36, v66, v67, s[64:65]", "v_bfe_u32 v33, v36, v74, 1",
                         "v_cndmask_b32_e64 v37, v68, v69, s[64:65]", "v_bfe_u32 v34, v37, v74, 1"]
    assert "v_add_u32_e32 v54, v54, v36" in code                      # min / max clauses the lane's tape keeps
    assert code[-5:-1] == store
    # two of them: the earlier one's record is asked for as soon as the later one's bits are out
    code = generated(mpr, [clause(0, 1, 2, 3), clause(OP["MAX_LHS_IMM"], 4, 1, 0, PI), clause(OP["MIN_LHS_RHS"], 5, 4, 2), clause(0, 5)], 6)
    assert code[0] == "ds_read_b128 v[66:69], v75 offset:16"
    k = code.index("v_bfe_u32 v34, v37, v74, 1")
    assert code[k + 1] == "ds_read_b128 v[66:69], v75" and code.count("s_waitcnt lgkmcnt(0)") == 2
    # whole tapes: architecture (93 slots, 488 min / max clauses), prospero; a slot beyond 95 is not taken
    for name, nch in (("architecture", 488), ("prospero", 2354)):
        words = [int(w) for w in tapes(name).data]
        arr = np.array(words, dtype=np.uint64)
        buf = (ctypes.c_uint32 * 400000)()
        n = mpr.lib().mpr_test_tile_gen(arr.ctypes.data, len(arr), 6, buf, 400000)
        assert 10000 < n < 400000
        assert sum(1 for d in buf[:n] if (d & 0xFFFF0000) == 0xD9FE0000) == nch      # one ds_read_b128 per min / max clause
    assert generated(mpr, [clause(0, 1, 2, 3), clause(OP["NEG_LHS"], 96, 1), clause(0, 96)], 6) is None


def test_backward_rows_for_tapes_that_are_shortened_again(mpr):
    """which = 3: the walk a stage below the first pushes by, and the one that writes tapes such a stage will shorten.  The tape
    being shortened is the PARENT's: s[0..23] hold one bit per clause of the root tape (is it on that tape?), a clause decided
    above (s[64:65] lhs / s[66:67] rhs) is walked as the COPY it is there, and v64.. collect the clauses of the tape written."""
    OP = mpr.OP
    take = ["v_cmp_eq_u32_e32 vcc, v61, v62", "s_cbranch_vccz 1", "s_swappc_b64 s[36:37], s[62:63]"]
    store = ["v_lshlrev_b32_e32 v44, 3, v61", "s_mov_b64 exec, vcc", "global_store_dwordx2 v44, v[46:47], s[76:77]", "s_mov_b64 exec, -1"]
    code = one(mpr, clause(OP["ADD_LHS_RHS"], 4, 1, 2), 3)
    assert code[0] == "s_bitcmp1_b32 s0, 0" and code[1] == "s_cbranch_scc0 24"      # not on the parent's tape: over the row (24 dwords)
    assert code[2:] == ["v_bfe_u32 v32, v60, 4, 1", "v_sub_u32_e32 v61, v61, v32"] + take + [
        "v_and_b32_e32 v60, 0xffffffef, v60", "v_lshl_or_b32 v60, v32, 1, v60", "v_lshl_or_b32 v60, v32, 2, v60",
        "v_mov_b32_e32 v46, 0x201040e", "v_mov_b32_e32 v47, 0", "v_lshl_or_b32 v64, v32, 0, v64", "v_cmp_ne_u32_e32 vcc, 0, v32"] + store
    # a min: its own body (decided by the lanes or not at all), or the COPY it became above
    code = one(mpr, clause(OP["MIN_LHS_RHS"], 4, 1, 2), 3)
    assert code[2] == "s_bitcmp1_b64 s[64:65], 0" and code[4] == "s_bitcmp1_b64 s[66:67], 0"
    branches = [k for k, l in enumerate(code) if l.startswith("s_branch")]
    assert len(branches) == 2
    lhs = code[branches[0] + 1:branches[1]]
    assert lhs[0] == "v_bfe_u32 v32, v60, 4, 1" and "v_mov_b32_e32 v46, 0x201041c" in lhs and "v_lshl_or_b32 v60, v32, 2, v60" in lhs   # a COPY's unused operand counts
    assert "v_mov_b32_e32 v46, 0x201041d" in code[branches[1] + 1:]
    own = code[6:branches[0]]
    assert "v_lshl_or_b32 v64, v35, 0, v64" in own and "v_lshl_or_b32 v41, v36, 0, v41" in own


def test_deriv_rows(mpr):
    """The normals pass's walk: lane = pixel * 4 + component (dx, dy, dz, value), slot s = v[50 + s], s[98:99] = the value lanes,
    the value of an operand reaches its quad through DPP quad_perm:[3,3,3,3]; routines return through s[70:71]."""
    OP = mpr.OP
    q3 = " quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf"

    def d(c):
        code = generated(mpr, [clause(0, 1, 2, 3), c, clause(0, (c >> 8) & 0xFF)], 2)
        assert code[-2:] == ["v_mov_b32_e32 v37, v%d" % (50 + ((c >> 8) & 0xFF)), "s_setpc_b64 s[38:39]"]
        return code[:-2]

    assert d(clause(OP["ADD_LHS_RHS"], 4, 1, 2)) == ["v_add_f32_e32 v54, v51, v52"]
    assert d(clause(OP["MUL_LHS_IMM"], 4, 1, 0, PI)) == ["v_mul_f32_e32 v54, 0x40490fdb, v51"]
    # a + imm, a - imm: only the value lane
    assert d(clause(OP["SUB_LHS_IMM"], 4, 1, 0, PI)) == ["v_subrev_f32_e32 v38, 0x40490fdb, v51", "v_cndmask_b32_e64 v54, v51, v38, s[98:99]"]
    # product rule: isv ? a b : a bv + b av
    assert d(clause(OP["MUL_LHS_RHS"], 4, 1, 2)) == [
        "v_mul_f32_dpp v38, v52, v51" + q3, "v_mul_f32_dpp v39, v51, v52" + q3, "v_mul_f32_e32 v40, v51, v52",
        "v_add_f32_e32 v38, v38, v39", "v_cndmask_b32_e64 v54, v38, v40, s[98:99]"]
    assert d(clause(OP["SQUARE_LHS"], 4, 1)) == [
        "v_mul_f32_dpp v38, v51, v51" + q3, "v_mul_f32_e32 v39, v51, v51", "v_add_f32_e32 v38, v38, v38",
        "v_cndmask_b32_e64 v54, v38, v39, s[98:99]"]
    # max: the comparison of the values, overridden by the decisions of the pixel's tiles (bit 0 of v74: lhs, of v76: rhs)
    assert d(clause(OP["MAX_LHS_RHS"], 4, 1, 2)) == [
        "v_mov_b32_dpp v39, v52" + q3, "v_mov_b32_dpp v46, v51" + q3, "v_cmp_ge_f32_e32 vcc, v46, v39",
        "v_bfe_u32 v40, v74, 0, 1", "v_bfe_u32 v41, v76, 0, 1", "v_cmp_ne_u32_e64 s[42:43], 0, v40", "v_cmp_ne_u32_e64 s[44:45], 0, v41",
        "s_nop 0", "s_or_b64 vcc, vcc, s[42:43]", "s_andn2_b64 vcc, vcc, s[44:45]", "v_cndmask_b32_e32 v54, v52, v51, vcc"]
    # exp: e = exp(av); isv ? e : e a — the float pass's routine, called
    assert d(clause(OP["EXP_LHS"], 4, 1)) == [
        "v_mov_b32_e32 v43, v51", "v_mov_b32_dpp v35, v51" + q3, "s_swappc_b64 s[70:71], s[76:77]", "v_mul_f32_e32 v38, v37, v43",
        "v_cndmask_b32_e64 v54, v38, v37, s[98:99]"]
    # a register written by one of the two instructions before a DPP read gets its wait states
    code = generated(mpr, [clause(0, 1, 2, 3), clause(OP["ADD_LHS_RHS"], 4, 1, 2), clause(OP["SQUARE_LHS"], 5, 4), clause(0, 5)], 2)
    assert code[:3] == ["v_add_f32_e32 v54, v51, v52", "s_nop 1", "v_mul_f32_dpp v38, v54, v54" + q3]


def test_whole_tapes(mpr):
    allowed = {"v_mov_b32_e32", "v_add_f32_e64", "v_add_f32_e32", "v_sub_f32_e32", "v_subrev_f32_e32", "v_xor_b32_e32", "v_mul_f32_e64",
               "v_mul_f32_e32", "s_swappc_b64", "s_setpc_b64", "v_cndmask_b32_e64", "v_cndmask_b32_e32", "v_or_b32_e32", "v_and_b32_e32",
               "v_bfe_u32", "v_lshl_or_b32", "v_sub_u32_e32", "v_add_u32_e32", "v_cmp_eq_u32_e32", "v_cmp_ne_u32_e32", "s_cbranch_vccz",
               "v_lshlrev_b32_e32", "s_mov_b64", "global_store_dwordx2", "s_nop", "s_movk_i32"}
    tape = mpr.Tape(mpr.model("bear"))
    words = [int(w) for w in np.asarray(tape.data)]
    allowed |= {"v_mul_f32_dpp", "v_mov_b32_dpp", "v_cmp_lt_f32_e32", "v_cmp_ge_f32_e32", "v_cmp_ne_u32_e64", "s_or_b64", "s_andn2_b64"}
    allowed |= {"s_bitcmp1_b32", "s_bitcmp1_b64", "s_cbranch_scc0", "s_cbranch_scc1", "s_branch"}
    for which in (1, 2, 3):
        code = generated(mpr, words, which)
        assert code is not None and code[-1] == "s_setpc_b64 s[38:39]"
        assert {l.split()[0] for l in code} <= allowed
        # one store per clause in the backward code, one decision record per min / max clause in the forward code
        if which == 1:
            assert sum(l.startswith("global_store") for l in code) == len(words) - 2
        elif which == 3:
            assert sum(l.startswith("s_bitcmp1_b32") for l in code) == len(words) - 2
        elif which == 2:
            assert sum(l.startswith("s_andn2_b64 vcc") for l in code) == tape.num_choices
    # tapes the conventions do not fit are left to the interpreter
    for name in ("architecture", "prospero"):
        t = mpr.Tape(mpr.model(name))
        assert generated(mpr, [int(w) for w in np.asarray(t.data)], 1) is None


def test_deriv_walk_with_guarded_dead_runs(mpr, tapes):
    """which = 5: the normals pass's walk jumping over the runs that are dead for EVERY pixel of the wavefront (s[64:65] / s[66:67]:
    the min / max clauses all 64 lanes decided for the lhs / rhs).  Nothing decided: exactly the plain Deriv walk; decisions skip
    whole clauses only."""
    from test_voxel_gen import disassemble, walk
    words = [int(w) for w in tapes("bear").data]
    arr = np.array(words, dtype=np.uint64)
    buf = (ctypes.c_uint32 * 65536)()
    n = mpr.lib().mpr_test_tile_gen(arr.ctypes.data, len(arr), 2, buf, 65536)
    plain = list(buf[:n])
    n = mpr.lib().mpr_test_tile_gen(arr.ctypes.data, len(arr), 5, buf, 65536)
    guarded = list(buf[:n])
    assert n > len(plain)
    # (a guard resets the emitter's DPP hazard bookkeeping: an s_nop the plain code needs in front of a quad_perm read may be
    # missing or extra right behind a guard, never wrong: two scalar instructions stand in for the wait states)
    strip = lambda lines: [l for l in lines if not l.startswith("s_nop") and not l.startswith("s_setpc_b64 s[38:39]")]
    assert strip(walk(guarded, 0, 0)) == strip(disassemble(plain))
    rng = np.random.default_rng(4)
    for trial in range(6):
        dl = int(rng.integers(0, 1 << 27)) & ~int(rng.integers(0, 1 << 27))
        dr = int(rng.integers(0, 1 << 27)) & ~dl & ~int(rng.integers(0, 1 << 27))
        got = strip(walk(guarded, dl, dr))
        it = iter(strip(disassemble(plain)))
        assert all(any(x == y for y in it) for x in got)
        assert len(got) < len(strip(disassemble(plain)))
