"""The float pass of tapes the host generates code for runs the ROOT tape's float walk as machine code written by
csrc/voxel_gen.cpp: clauses on their own slot registers, square root / exp / log / division by a constant in line, decided
min / max clauses as scalar branches, runs of dead clauses jumped over.  Here that code is checked without a GPU: the words are
compared with what the ROCm assembler makes of the instructions they are meant to be — for the inline bodies that is the text of
csrc/asm_float_bodies.hpp (the interpreters' routines, which the oracle is pinned against) with the argument and result
registers renamed — and the guards of bear's tape are checked against a liveness analysis of its own.
tests/test_gpu_primitives.py and the frame tests run it on the device against the oracle."""
import ctypes
import os
import re
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM_MC = "/opt/rocm/lib/llvm/bin/llvm-mc"
pytestmark = pytest.mark.skipif(not os.path.exists(LLVM_MC), reason="llvm-mc not found")
PI = 0x40490FDB


def clause(op, out=0, lhs=0, rhs=0, imm=0):
    return op | out << 8 | lhs << 16 | rhs << 24 | imm << 32


def generated(mpr, words, min_run=5):
    arr = np.array(words, dtype=np.uint64)
    buf = (ctypes.c_uint32 * 131072)()
    info = (ctypes.c_int32 * 3)()
    n = mpr.lib().mpr_test_voxel_gen(arr.ctypes.data, len(arr), min_run, buf, 131072, info)
    assert n >= 0
    return list(buf[:n]), list(info)


def assemble(lines):
    """-> dwords of the instructions in `lines`"""
    r = subprocess.run([LLVM_MC, "-arch=amdgcn", "-mcpu=gfx950", "-show-encoding"], input=("\n".join(lines) + "\n").encode(),
                       capture_output=True, check=True)
    assert not r.stderr.strip(), r.stderr.decode()
    out = []
    for l in r.stdout.decode().splitlines():
        m = re.search(r"encoding: \[([^\]]*)\]", l)
        if m:
            b = bytes(int(x, 16) for x in m.group(1).split(","))
            out += list(struct.unpack("<%dI" % (len(b) // 4), b))
    return out


def disassemble(dwords, sizes=False):
    text = ",".join("0x%02x" % b for d in dwords for b in struct.pack("<I", d))
    r = subprocess.run([LLVM_MC, "-arch=amdgcn", "-mcpu=gfx950", "-disassemble", "-show-encoding"], input=text.encode(), capture_output=True, check=True)
    assert not r.stderr.strip(), r.stderr.decode()
    out, sz = [], []
    for l in r.stdout.decode().splitlines():
        m = re.match(r"\s*(\S.*?)\s*; encoding: \[([^\]]*)\]", l)
        if m and not m.group(1).startswith("."):
            out.append(" ".join(m.group(1).split()))
            sz.append(len(m.group(2).split(",")) // 4)
    return (out, sz) if sizes else out


def insn_lengths(code):
    """dwords of every instruction of `code`, decoded from the words themselves (llvm-mc -show-encoding re-encodes what it
    disassembled, and a literal that has an inline-constant twin comes back shorter than it went in)."""
    out, pc = [], 0
    while pc < len(code):
        w = code[pc]
        top = w >> 23
        n = 1
        if top == 0b101111101:                                   # SOP1
            n += (w & 0xFF) == 255
        elif top == 0b101111110:                                 # SOPC
            n += (w & 0xFF) == 255 or ((w >> 8) & 0xFF) == 255
        elif top == 0b101111111:                                 # SOPP
            pass
        elif (w >> 28) == 0b1011:                                # SOPK (s_setreg_imm32_b32 carries a literal)
            n += ((w >> 23) & 0x1F) == 20
        elif (w >> 30) == 0b10:                                  # SOP2
            n += (w & 0xFF) == 255 or ((w >> 8) & 0xFF) == 255
        elif (w >> 26) in (0b110100, 0b110110, 0b110111, 0b111000, 0b110000):      # VOP3, DS, FLAT / GLOBAL, MUBUF, SMEM
            n = 2
        elif (w >> 25) in (0b0111110, 0b0111111):                # VOPC, VOP1
            n += (w & 0x1FF) in (249, 250, 255)
        else:                                                    # VOP2 (v_fmamk_f32 / v_fmaak_f32 always carry their constant)
            n += (w & 0x1FF) in (249, 250, 255) or (w >> 25) in (23, 24)
        out.append(n)
        pc += n
    assert pc == len(code), "the code does not end on an instruction boundary"
    return out


def body_of(macro):
    """the instructions of one MPR_ASM_* macro of asm_float_bodies.hpp (labels kept, as 'L_x:')"""
    src = open(os.path.join(ROOT, "mpr_amd", "csrc", "asm_float_bodies.hpp")).read()
    m = re.search(r"#define %s\b(.*?)(?=\n#define|\n/\*|\Z)" % macro, src, re.S)
    assert m, macro
    text = "".join(re.findall(r'"((?:[^"\\]|\\.)*)"', m.group(1)))
    lines = [l.strip() for l in text.replace("\\n", "\n").split("\n")]
    return [l for l in lines if l]


PRO = 3          # dwords of the prologue: x, y, z into their slots
# one clause, operands in slots 1 (v49) and 2 (v50), result in slot 4 (v52); the end clause names slot 4
def one(mpr, c, min_run=5):
    code, info = generated(mpr, [clause(0, 1, 2, 3), c, clause(0, (c >> 8) & 0xFF)], min_run)
    assert disassemble(code[:PRO]) == ["v_mov_b32_e32 v49, v32", "v_mov_b32_e32 v50, v33", "v_mov_b32_e32 v51, v34"]
    return code[PRO:], info


def split_main(code, out_reg):
    """main code up to and including the epilogue (v_mov v37, out; s_setpc), and the stubs behind it"""
    end = assemble(["v_mov_b32 v37, v%d" % out_reg, "s_setpc_b64 s[72:73]"])
    for i in range(len(code) - 1):
        if code[i:i + 2] == end:
            return code[:i], code[i + 2:]
    raise AssertionError("no epilogue")


def test_one_instruction_clauses(mpr):
    OP = mpr.OP
    cases = [
        (clause(OP["ADD_LHS_RHS"], 4, 1, 2), "v_add_f32 v52, v49, v50"),
        (clause(OP["ADD_LHS_IMM"], 4, 1, 0, PI), "v_add_f32 v52, 0x40490fdb, v49"),
        (clause(OP["ADD_LHS_IMM"], 4, 1, 0, 0x3F800000), "v_add_f32 v52, 1.0, v49"),       # an inline constant where there is one
        (clause(OP["SUB_LHS_IMM"], 4, 1, 0, PI), "v_subrev_f32 v52, 0x40490fdb, v49"),      # lhs - imm
        (clause(OP["SUB_IMM_RHS"], 4, 0, 2, PI), "v_sub_f32 v52, 0x40490fdb, v50"),         # imm - rhs
        (clause(OP["SUB_LHS_RHS"], 4, 1, 2), "v_sub_f32 v52, v49, v50"),
        (clause(OP["MUL_LHS_RHS"], 4, 1, 2), "v_mul_f32 v52, v49, v50"),
        (clause(OP["MUL_LHS_IMM"], 4, 1, 0, PI), "v_mul_f32 v52, 0x40490fdb, v49"),
        (clause(OP["SQUARE_LHS"], 4, 1), "v_mul_f32 v52, v49, v49"),
        (clause(OP["NEG_LHS"], 4, 1), "v_xor_b32 v52, 0x80000000, v49"),
        (clause(OP["ABS_LHS"], 4, 1), "v_and_b32 v52, 0x7fffffff, v49"),
        (clause(OP["COPY_LHS"], 4, 1), "v_mov_b32 v52, v49"),
        (clause(OP["COPY_RHS"], 4, 0, 2), "v_mov_b32 v52, v50"),
        (clause(OP["COPY_IMM"], 4, 0, 0, PI), "v_mov_b32 v52, 0x40490fdb"),
        # a division by a power of two is a multiplication by its reciprocal: the same real number, rounded once either way
        (clause(OP["DIV_LHS_IMM"], 4, 1, 0, 0x40000000), "v_mul_f32 v52, 0.5, v49"),
        (clause(OP["DIV_LHS_IMM"], 4, 1, 0, 0xC1000000), "v_mul_f32 v52, 0xbe000000, v49"),
        (clause(OP["DIV_LHS_IMM"], 4, 1, 0, 0x7E800000), "v_mul_f32 v52, 0x00800000, v49"),
    ]
    for c, text in cases:
        code, info = one(mpr, c)
        main, stubs = split_main(code, 52)
        assert main == assemble([text]), (text, disassemble(main))
        assert stubs == [] and info[2] == 0


def renamed(lines, a, o):
    return [re.sub(r"\bv37\b", "v%d" % o, re.sub(r"\bv35\b", "v%d" % a, l)) for l in lines]


@pytest.mark.parametrize("o", [52, 49])            # a fresh result register / the result over its operand
def test_square_root_exp_log_in_line(mpr, o):
    """The inline bodies ARE asm_float_bodies.hpp's fast paths on the clause's own registers; anything else leaves through a stub
    that calls the interpreters' full routine on (v35) -> v37."""
    OP = mpr.OP
    out_slot = o - 48
    stub = lambda routine: assemble(["v_mov_b32 v35, v49", "s_swappc_b64 s[30:31], s[%d:%d]" % (routine, routine + 1), "v_mov_b32 v%d, v37" % o])

    # square root
    code, info = one(mpr, clause(OP["SQRT_LHS"], out_slot, 1))
    main, stubs = split_main(code, o)
    body = [l for l in body_of("MPR_ASM_SQRT_BODY") if not l.endswith(":")]
    assert body[3].startswith("s_cbranch_scc0")
    want = assemble(renamed(body[:3], 49, o) + ["s_cbranch_scc0 0"] + renamed(body[4:], 49, o))
    assert [w for k, w in enumerate(main) if k != 5] == [w for k, w in enumerate(want) if k != 5], disassemble(main)
    # ... the branch reaches the stub, the stub comes back behind the body
    assert (main[5] >> 16) == 0xBF84 and (main[5] & 0xFFFF) == len(main) + 2 - (5 + 1)
    assert stubs[:-1] == stub(54) and (stubs[-1] >> 16) == 0xBF82
    back = stubs[-1] & 0xFFFF
    assert PRO + len(main) + 2 + len(stubs) + (back - 0x10000 if back >= 0x8000 else back) == PRO + len(main)

    # exp: the range test comes first (the result may overwrite the operand), the body follows unchanged
    code, info = one(mpr, clause(OP["EXP_LHS"], out_slot, 1))
    main, stubs = split_main(code, o)
    body = [l for l in body_of("MPR_ASM_EXP_BODY") if not l.endswith(":")]
    assert body[-3].startswith("s_mov_b32 s91, 0x42ae0000") and body[-2].startswith("v_cmp_nle_f32 vcc, |v35|, s91") and body[-1].startswith("s_cbranch_vccnz")
    # (the polynomial's first step, c r + c', is one v_fmaak with c waiting in v5 where the routine spends a v_mov and a v_fmac)
    k = body.index("v_mov_b32 v40, 0x3ab743ce")
    assert body[k + 1] == "v_fmac_f32 v40, 0x39506967, v39"
    body[k:k + 2] = ["v_fmaak_f32 v40, v5, v39, 0x3ab743ce"]
    want = assemble(["v_cmp_nle_f32 vcc, |v49|, s80", "s_cbranch_vccnz 0"] + renamed(body[:-3], 49, o))
    assert main[:2] == want[:2] and main[3:] == want[3:], disassemble(main)
    assert (main[2] >> 16) == 0xBF87 and (main[2] & 0xFFFF) == len(main) + 2 - 3
    assert stubs[:-1] == stub(56)

    # log: the class test, then the body from L_logmain on with the operand read in place
    code, info = one(mpr, clause(OP["LOG_LHS"], out_slot, 1))
    main, stubs = split_main(code, o)
    body = body_of("MPR_ASM_LOG_BODY")
    body = body[body.index("L_logmain_%=:") + 1:]
    assert body[0] == "v_lshrrev_b32 v39, 23, v38" and body[1] == "v_add_u32 v39, v39, v40" and body[2] == "v_and_b32 v38, 0x7fffff, v38"
    assert body[-3].startswith("s_cmp_eq_u32 s91, 1") and body[-1] == "L_logdone_%=:"
    k = body.index("v_mov_b32 v40, 0xbdebd1b8")
    assert body[k + 1] == "v_fmac_f32 v40, 0x3d9021bb, v38"
    body[k:k + 2] = ["v_fmaak_f32 v40, v6, v38, 0xbdebd1b8"]
    fast = ["v_lshrrev_b32 v39, 23, v49", "v_add_u32 v39, 0xffffff82, v39", "v_and_b32 v38, 0x7fffff, v49"] + renamed(body[3:-3], 49, o)
    want = assemble(["v_cmp_class_f32 vcc, v49, s81", "s_cmp_eq_u64 vcc, exec", "s_cbranch_scc0 0"] + fast)
    assert main[:3] == want[:3] and main[4:] == want[4:], disassemble(main)
    assert (main[3] >> 16) == 0xBF84 and (main[3] & 0xFFFF) == len(main) + 2 - 4
    assert stubs[:-1] == stub(58)


def test_division_by_a_constant_in_line(mpr):
    OP = mpr.OP
    code, info = one(mpr, clause(OP["DIV_LHS_IMM"], 4, 1, 0, 0x40400000))            # x / 3
    main, stubs = split_main(code, 52)
    y = struct.unpack("<I", struct.pack("<f", np.float32(1.0) / np.float32(3.0)))[0]
    want = assemble(["v_mul_f32 v39, v49, v49", "v_cmp_class_f32 vcc, v39, v7", "s_cbranch_vccnz 0",
                     "v_mul_f32 v37, 0x%08x, v49" % y, "v_fmamk_f32 v39, v37, 0xc0400000, v49", "v_fmamk_f32 v37, v39, 0x%08x, v37" % y,
                     "v_fmamk_f32 v39, v37, 0xc0400000, v49", "v_fmamk_f32 v52, v39, 0x%08x, v37" % y])
    assert main[:2] == want[:2] and main[3:] == want[3:], disassemble(main)
    assert stubs[:-1] == assemble(["v_mov_b32 v35, v49", "v_mov_b32 v36, 0x40400000", "s_swappc_b64 s[30:31], s[52:53]", "v_mov_b32 v52, v37"])
    # outside 2^-30 .. 2^30 (and not a power of two): the general division
    code, info = one(mpr, clause(OP["DIV_LHS_IMM"], 4, 1, 0, 0x4F400000))
    main, stubs = split_main(code, 52)
    assert main == assemble(["v_mov_b32 v35, v49", "v_mov_b32 v36, 0x4f400000", "s_swappc_b64 s[30:31], s[52:53]", "v_mov_b32 v52, v37"])
    code, info = one(mpr, clause(OP["DIV_LHS_RHS"], 4, 1, 2))
    assert split_main(code, 52)[0] == assemble(["v_mov_b32 v35, v49", "v_mov_b32 v36, v50", "s_swappc_b64 s[30:31], s[52:53]", "v_mov_b32 v52, v37"])
    code, info = one(mpr, clause(OP["DIV_IMM_RHS"], 4, 0, 2, PI))
    assert split_main(code, 52)[0] == assemble(["v_mov_b32 v35, 0x40490fdb", "v_mov_b32 v36, v50", "s_swappc_b64 s[30:31], s[52:53]", "v_mov_b32 v52, v37"])


def test_called_routines(mpr):
    OP = mpr.OP
    for name, pair in (("SIN_LHS", 60), ("COS_LHS", 62)):
        code, info = one(mpr, clause(OP[name], 4, 1))
        assert split_main(code, 52)[0] == assemble(["v_mov_b32 v35, v49", "s_swappc_b64 s[30:31], s[%d:%d]" % (pair, pair + 1), "v_mov_b32 v52, v37"])
    for name, pair in (("ASIN_LHS", 64), ("ACOS_LHS", 66), ("ATAN_LHS", 68)):
        code, info = one(mpr, clause(OP[name], 4, 1))
        assert split_main(code, 52)[0] == assemble(["v_mov_b32 v0, v49", "s_swappc_b64 s[30:31], s[%d:%d]" % (pair, pair + 1), "v_mov_b32 v52, v0",
                                                    "v_mov_b32 v7, 0x2ff", "v_mov_b32 v5, 0x39506967", "v_mov_b32 v6, 0x3d9021bb"])


def test_min_max_with_the_tiles_decisions(mpr):
    """s[76:77] / s[78:79]: min / max clauses the tile decided for the lhs / rhs.  Decided: a branch to a stub that copies the
    chosen operand — or straight past the clause when that operand already sits in the out register."""
    OP = mpr.OP
    code, info = one(mpr, clause(OP["MAX_LHS_RHS"], 4, 1, 2))
    main, stubs = split_main(code, 52)
    assert disassemble(main) == ["s_bitcmp1_b64 s[76:77], 0", "s_cbranch_scc1 5", "s_bitcmp1_b64 s[78:79], 0", "s_cbranch_scc1 5", "v_max_f32_e32 v52, v49, v50"]
    # (behind the clause: the epilogue's two dwords, then the stubs; each returns to the instruction behind the v_max)
    assert disassemble(stubs) == ["v_mov_b32_e32 v52, v49", "s_branch 65532", "v_mov_b32_e32 v52, v50", "s_branch 65530"]
    code, info = one(mpr, clause(OP["MIN_LHS_IMM"], 4, 1, 0, PI))
    main, stubs = split_main(code, 52)
    assert disassemble(main)[-1] == "v_min_f32_e32 v52, 0x40490fdb, v49"
    assert disassemble(stubs)[2] == "v_mov_b32_e32 v52, 0x40490fdb"
    # the result over the lhs: "decided for the lhs" has nothing to do
    code, info = one(mpr, clause(OP["MIN_LHS_RHS"], 1, 1, 2))
    main, stubs = split_main(code, 49)
    assert disassemble(main) == ["s_bitcmp1_b64 s[76:77], 0", "s_cbranch_scc1 3", "s_bitcmp1_b64 s[78:79], 0", "s_cbranch_scc1 3", "v_min_f32_e32 v49, v49, v50"]
    assert disassemble(stubs) == ["v_mov_b32_e32 v49, v50", "s_branch 65532"]
    # the second min / max of a tape tests bit 1
    OPm = OP["MAX_LHS_RHS"]
    code, info = generated(mpr, [clause(0, 1, 2, 3), clause(OPm, 4, 1, 2), clause(OPm, 4, 4, 3), clause(0, 4)])
    text = disassemble(code)
    assert "s_bitcmp1_b64 s[76:77], 1" in text and "s_bitcmp1_b64 s[78:79], 1" in text and info[0] == 2


def walk(code, dl, dr):
    """Follow the code's scalar branches for a tile with decisions (dl, dr); routines return at once, v_cmp-driven branches (the
    rare operands) are not taken.  -> the disassembled instructions executed, in order."""
    text, sizes = disassemble(code), insn_lengths(code)
    assert len(text) == len(sizes)
    at, pc = [], 0
    for n in sizes:                       # dword index of every instruction
        at.append(pc)
        pc += n
    index = {a: k for k, a in enumerate(at)}
    out, k, scc = [], 0, 0
    for _ in range(100000):
        l = text[k]
        if l.startswith("s_setpc_b64 s[72:73]") or l.startswith("s_setpc_b64 s[38:39]"):
            return out
        m = re.match(r"s_bitcmp1_b64 s\[(\d+):\d+\], (\d+)", l)
        if m:      # (the float walk keeps the tile's decisions in s[76:79], the interval walk those from above in s[72:75])
            scc = ((dl if m.group(1) in ("76", "72", "64") else dr) >> int(m.group(2))) & 1      # (the Deriv walk: s[64:65] / s[66:67])
        m = re.match(r"(s_branch|s_cbranch_scc1|s_cbranch_scc0|s_cbranch_vccnz) (\d+)", l)
        if m:
            off = int(m.group(2))
            off = off - 0x10000 if off >= 0x8000 else off
            taken = m.group(1) == "s_branch" or (m.group(1) == "s_cbranch_scc1" and scc)
            if taken:
                k = index[at[k] + 1 + off]
                continue
        elif not l.startswith("s_bitcmp1"):
            out.append(l)
        k += 1
    raise AssertionError("the walk does not end")


def test_dead_runs_are_jumped_over(mpr):
    """max(f(x), g(y)) where f and g are chains of four clauses: with the clause decided for the lhs the rhs chain is dead and
    the code jumps over it, and the other way round; undecided, everything runs."""
    OP = mpr.OP
    t = [clause(0, 1, 2, 3)]
    t += [clause(OP["ADD_LHS_IMM"], 4, 1, 0, PI), clause(OP["SQUARE_LHS"], 4, 4), clause(OP["NEG_LHS"], 4, 4), clause(OP["ABS_LHS"], 4, 4)]
    t += [clause(OP["MUL_LHS_IMM"], 5, 2, 0, PI), clause(OP["SQUARE_LHS"], 5, 5), clause(OP["NEG_LHS"], 5, 5), clause(OP["ABS_LHS"], 5, 5)]
    t += [clause(OP["MAX_LHS_RHS"], 4, 4, 5), clause(0, 4)]
    code, info = generated(mpr, t, min_run=4)
    assert info[0] == 1 and info[1] == 2
    f = ["v_add_f32_e32 v52, 0x40490fdb, v49", "v_mul_f32_e32 v52, v52, v52", "v_xor_b32_e32 v52, 0x80000000, v52", "v_and_b32_e32 v52, 0x7fffffff, v52"]
    g = ["v_mul_f32_e32 v53, 0x40490fdb, v50", "v_mul_f32_e32 v53, v53, v53", "v_xor_b32_e32 v53, 0x80000000, v53", "v_and_b32_e32 v53, 0x7fffffff, v53"]
    pro = ["v_mov_b32_e32 v49, v32", "v_mov_b32_e32 v50, v33", "v_mov_b32_e32 v51, v34"]
    ex = lambda dl, dr: walk(code, dl, dr)
    assert ex(0, 0) == pro + f + g + ["v_max_f32_e32 v52, v52, v53", "v_mov_b32_e32 v37, v52"]
    assert ex(1, 0) == pro + f + ["v_mov_b32_e32 v37, v52"]                                   # decided lhs: in place, nothing to copy
    assert ex(0, 1) == pro + g + ["v_mov_b32_e32 v52, v53", "v_mov_b32_e32 v37, v52"]
    # shorter than min_run: no guards
    assert generated(mpr, t, min_run=5)[1][1] == 0
    assert generated(mpr, t, min_run=0)[1][1] == 0


def symbolic(lines):
    """The value of v37 after the straight-line instruction sequence `lines` (disassembly), as a hash of the dataflow that
    produced it: every instruction's result = H(mnemonic, its source operands' values).  Two sequences that leave the same hash
    computed the same expression of (x, y, z)."""
    reg = {"v32": ("x",), "v33": ("y",), "v34": ("z",)}

    def val(tok):
        if re.fullmatch(r"-?\|?v\d+\|?|vcc|s\[\d+:\d+\]", tok) is None:
            return ("const", tok)
        name = tok.strip("-|")
        return (tok[0] == "-", "|" in tok, reg.get(name, ("stale", name)))

    for l in lines:
        m = re.match(r"(\S+) (.*)", l)
        op, args = m.group(1), [a.strip() for a in m.group(2).split(",")]
        if op == "s_swappc_b64":
            routine = args[1]
            if routine in ("s[64:65]", "s[66:67]", "s[68:69]"):
                reg["v0"] = hash(("leaf", routine, reg.get("v0")))
            else:
                reg["v37"] = hash(("routine", routine, reg.get("v35"), reg.get("v36") if routine == "s[52:53]" else None))
            continue
        if op.startswith("s_"):
            continue
        dst, srcs = args[0], args[1:]
        ins = [val(a) for a in srcs]
        if op.startswith("v_fmac"):
            ins.append(val(dst))
        if op.startswith("v_cndmask_b32_e32"):
            ins.append(val("vcc"))
        if op.startswith("v_subbrev_co"):
            reg[args[1]] = hash((op, "carry", tuple(ins)))
            ins = [val(a) for a in args[2:]]
        reg[dst] = hash((op, tuple(ins)))
    return reg.get("v37")


@pytest.mark.parametrize("min_run", [5, 3, 1])
def test_guards_of_bear_keep_every_live_clause(mpr, tapes, min_run):
    """For the tape the benchmark is quoted on: whatever the tile's decisions, the code with its guards computes the same
    expression as the code without any (symbolic execution of both instruction streams: a jumped-over clause that the result
    still depends on would leave a stale register in the dataflow), while running fewer instructions."""
    tape = tapes("bear").data
    words = [int(w) for w in tape]
    code, info = generated(mpr, words, min_run=min_run)
    plain, pinfo = generated(mpr, words, min_run=0)
    assert info[0] == 27 and info[1] >= 20 and pinfo[1] == 0
    assert walk(code, 0, 0) == walk(plain, 0, 0)                    # nothing decided: every clause runs
    rng = np.random.default_rng(5)
    fewer = 0
    for trial in range(24):
        dl = dr = 0
        for k in range(27):
            r = rng.integers(0, 3 if trial < 12 else 6)
            if r == 1:
                dl |= 1 << k
            elif r == 2:
                dr |= 1 << k
        got, ref = walk(code, dl, dr), walk(plain, dl, dr)
        assert symbolic(got) == symbolic(ref) is not None, (trial, hex(dl), hex(dr))
        assert len(got) <= len(ref)
        fewer += len(got) < len(ref)
    assert fewer >= 20
