"""The generated-code float pass (kernels_voxel_jit.hip) builds every clause's instructions from a template row:
`dword = T + (T & mask) + base`, T = the clause's registers and literal picked by a byte selector.  The library
restates that arithmetic on the host (mpr_test_jit_row); here the rows are disassembled with the ROCm assembler's
llvm-mc and compared with the instructions they are meant to be — no GPU needed."""
import ctypes
import os
import struct
import subprocess

import pytest

LLVM_MC = "/opt/rocm/lib/llvm/bin/llvm-mc"
PI = 0x40490FDB


def disassemble(dwords):
    text = ",".join("0x%02x" % b for d in dwords for b in struct.pack("<I", d))
    r = subprocess.run([LLVM_MC, "-arch=amdgcn", "-mcpu=gfx950", "-disassemble"], input=text.encode(), capture_output=True, check=True)
    assert not r.stderr.strip(), r.stderr.decode()
    return [" ".join(l.split()) for l in r.stdout.decode().splitlines() if l.strip() and not l.strip().startswith(".")]


def row(mpr, table, op, out=5, lhs=2, rhs=3, imm=PI, choice=7, opcode=None):
    buf = (ctypes.c_uint32 * 16)()
    clause = (op if opcode is None else opcode) | out << 8 | lhs << 16 | rhs << 24
    n = mpr.lib().mpr_test_jit_row(table, op, clause, imm, choice, buf, 16)
    assert n >= 0
    return disassemble(list(buf[:n]))


@pytest.mark.skipif(not os.path.exists(LLVM_MC), reason="llvm-mc not found")
def test_rows_disassemble_to_the_intended_instructions(mpr):
    OP = mpr.OP
    # slot s is v[48 + s]: out v53, lhs v50, rhs v51
    assert row(mpr, 1, OP["ADD_LHS_RHS"]) == ["v_add_f32_e32 v53, v50, v51"]
    assert row(mpr, 1, OP["SUB_LHS_IMM"]) == ["v_subrev_f32_e32 v53, 0x40490fdb, v50"]
    assert row(mpr, 1, OP["SUB_IMM_RHS"]) == ["v_sub_f32_e32 v53, 0x40490fdb, v51"]
    assert row(mpr, 1, OP["NEG_LHS"]) == ["v_xor_b32_e32 v53, 0x80000000, v50"]
    assert row(mpr, 1, OP["SQRT_LHS"]) == ["v_mov_b32_e32 v35, v50", "s_swappc_b64 s[30:31], s[54:55]", "v_mov_b32_e32 v53, v37"]
    # group form: min with the decisions of choice 7 in s[76:79] (no canonicalising v_max pair: the code runs with MODE.IEEE off);
    # the selects take their mask from an SGPR pair (VOP3): the VOP2 form on a vcc the scalar unit wrote runs at a tenth of the rate
    assert row(mpr, 1, OP["MIN_LHS_RHS"]) == [
        "v_min_f32_e32 v37, v50, v51",
        "s_bitcmp1_b64 s[76:77], 7", "s_cselect_b64 s[94:95], -1, 0", "v_cndmask_b32_e64 v37, v37, v50, s[94:95]",
        "s_bitcmp1_b64 s[78:79], 7", "s_cselect_b64 s[94:95], -1, 0", "v_cndmask_b32_e64 v53, v37, v51, s[94:95]"]
    assert row(mpr, 1, OP["MAX_LHS_IMM"]) == [
        "v_mov_b32_e32 v38, 0x40490fdb", "v_max_f32_e32 v37, v50, v38",
        "s_bitcmp1_b64 s[76:77], 7", "s_cselect_b64 s[94:95], -1, 0", "v_cndmask_b32_e64 v37, v37, v50, s[94:95]",
        "s_bitcmp1_b64 s[78:79], 7", "s_cselect_b64 s[94:95], -1, 0", "v_cndmask_b32_e64 v53, v37, v38, s[94:95]"]
    # ... and of choice 64 + 7 in s[48:51] (row 33 = MIN_LHS_RHS for decisions 64..127)
    assert row(mpr, 1, 33, opcode=OP["MIN_LHS_RHS"])[1] == "s_bitcmp1_b64 s[48:49], 7"
    # tile form: no decisions
    assert row(mpr, 0, OP["MAX_LHS_RHS"]) == ["v_max_f32_e32 v53, v50, v51"]
    assert row(mpr, 0, OP["MIN_LHS_IMM"]) == ["v_min_f32_e32 v53, 0x40490fdb, v50"]
    # division by a constant, inline (the translator puts RN(1 / c) where the zeros are)
    assert row(mpr, 1, 30, opcode=OP["DIV_LHS_IMM"]) == [
        "v_mul_f32_e32 v39, v50, v50", "v_cmp_class_f32_e32 vcc, v39, v7", "s_cbranch_vccz 2", "v_mov_b32_e32 v35, v50",
        "s_swappc_b64 s[30:31], s[70:71]", "v_mul_f32_e32 v37, 0, v50", "v_fmamk_f32 v39, v37, 0xc0490fdb, v50",
        "v_fmamk_f32 v37, v39, 0x0, v37", "v_fmamk_f32 v39, v37, 0xc0490fdb, v50", "v_fmamk_f32 v53, v39, 0x0, v37"]


def test_every_opcode_has_a_row_of_at_most_fifteen_dwords(mpr):
    buf = (ctypes.c_uint32 * 16)()
    for table in (0, 1):
        for op in range(2, 30):
            n = mpr.lib().mpr_test_jit_row(table, op, op | 5 << 8 | 2 << 16 | 3 << 24, PI, 0, buf, 16)
            assert 1 <= n <= 15, (table, op, n)
    assert mpr.lib().mpr_test_jit_row(2, 2, 0, 0, 0, buf, 16) == -1
