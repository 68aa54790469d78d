"""The tile stages' interval forward walk as scheduled gfx950 code (csrc/interval_gen.cpp), checked WITHOUT a GPU three ways:
 1. the words the library emits == what the ROCm assembler makes of the generator's own assembler text (encoder);
 2. that text, run by tests/gfx_emu.py on random tiles, against the ORACLE's interval arithmetic clause by clause: the exact code
    bit for bit (lo, hi, every lane's min / max decisions; also on infinities, NaNs, zeros and denormals), the loose code as an
    enclosure that decides no more than the exact walk — with the hardware's approximate instructions moved by a random ulp;
 3. the wait states the chip does not interlock, along the executed path.
tests/test_gpu_*.py run the same code on the chip."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from gfx_emu import Emu, f2u, u2f

LLVM_MC = "/opt/rocm/lib/llvm/bin/llvm-mc"
pytestmark = pytest.mark.skipif(not os.path.exists(LLVM_MC), reason="llvm-mc not found")
FIRST, BELOW, GUARDED, MASKS = 0, 1, 2, 3
ROUTINE_OP = {68: "SQRT_LHS", 82: "DIV_LHS_RHS", 84: "DIV_LHS_RHS", 86: "ASIN_LHS", 88: "ACOS_LHS", 90: "ATAN_LHS", 98: "EXP_LHS", 96: "LOG_LHS"}


def generate(mpr, words, kind, loose, window=0, min_run=3, tight=False):
    f = mpr.lib().mpr_test_interval_gen
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                  ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
    arr = np.asarray(words, dtype=np.uint64)
    buf = (ctypes.c_uint32 * 400000)()
    txt = ctypes.create_string_buffer(8_000_000)
    info = (ctypes.c_int32 * 8)()
    # (loose = 3: loose code for the harness with 64 vector registers, as the tile stages run it)
    # (MASKS: the first-stage walk of tapes of up to 93 slots, all registers)
    # (+ 8: tight code — a second enclosure of the result in v[38:39], for the harness with 80 vector registers)
    n = f(arr.ctypes.data, len(arr), kind, ((1 if kind == MASKS else 3) if loose else 0) | (8 if tight else 0), window, min_run, buf, 400000, txt, 8_000_000, info)
    if n < 0:
        return None
    names = ["instructions", "nops", "window", "max_vgprs", "max_sgpr_pairs", "nchoices", "est_cycles"]
    return list(buf[:n]), txt.value.decode().splitlines(), dict(zip(names, info))


def assemble(lines):
    """per instruction, the bytes the ROCm assembler encodes it to"""
    r = subprocess.run([LLVM_MC, "-arch=amdgcn", "-mcpu=gfx950", "-show-encoding"], input="\n".join(lines).encode(), capture_output=True)
    assert r.returncode == 0 and not r.stderr.strip(), r.stderr.decode()[:2000]
    out = []
    for l in r.stdout.decode().splitlines():
        m = re.search(r"encoding: \[([^\]]*)\]", l)
        if m:
            out.append([int(b, 16) for b in m.group(1).split(",")])
    assert len(out) == len(lines)
    return out


def checked_code(mpr, words, kind, loose, window=0, min_run=3, tight=False):
    g = generate(mpr, words, kind, loose, window, min_run, tight)
    assert g is not None
    code, lines, info = g
    enc = assemble(lines)
    flat = [b for e in enc for b in e]
    mine = [b for w in code for b in int(w).to_bytes(4, "little")]
    if flat != mine:
        at = 0
        for l, e in zip(lines, enc):
            assert mine[at:at + len(e)] == e, "%s: assembler %s, generator %s" % (l, bytes(e).hex(), bytes(mine[at:at + len(e)]).hex())
            at += len(e)
    assert flat == mine
    return lines, [len(e) // 4 for e in enc], info


# ---- the oracle's walk of a tape, clause by clause ----
def tape_fields(words):
    w = np.asarray(words, dtype=np.uint64)
    op = (w & 0xFF).astype(int)
    return op, ((w >> 8) & 0xFF).astype(int), ((w >> 16) & 0xFF).astype(int), ((w >> 24) & 0xFF).astype(int), ((w >> 32) & 0xFFFFFFFF).astype(np.uint32)


def oracle_walk(mpr, orc, words, x, y, z, dec_l=0, dec_r=0):
    """-> (lo, hi) of the end clause and per min / max clause the lanes' choices (0 / 1 / 2); a clause decided above is the copy of
    the chosen operand it is on the parent's tape (reference src/context.cu:351-458)"""
    OP = mpr.OP
    minmax = {OP["MIN_LHS_IMM"], OP["MIN_LHS_RHS"], OP["MAX_LHS_IMM"], OP["MAX_LHS_RHS"]}
    op, o, l, r, imm = tape_fields(words)
    n = len(x[0])
    slots = {int(o[0]): x, int(l[0]): y, int(r[0]): z}
    zero = (np.zeros(n, np.float32), np.zeros(n, np.float32))
    choices = []
    k = 0
    for i in range(1, len(words)):
        if op[i] == 0:
            return slots[int(o[i])], choices
        A = slots.get(int(l[i]), zero)
        B = slots.get(int(r[i]), zero)
        K = u2f(np.array([imm[i]], dtype=np.uint32))[0]
        if op[i] in minmax:
            if dec_l >> k & 1:
                out, ch = A, np.zeros(n, np.int32)
            elif dec_r >> k & 1:
                out = B if r[i] != 0 else (np.full(n, K, np.float32), np.full(n, K, np.float32))
                ch = np.zeros(n, np.int32)
            else:
                lo, hi, ch = orc.interval_op(int(op[i]), A[0], A[1], B[0], B[1], float(K))
                out = (lo, hi)
            choices.append(ch)
            k += 1
        else:
            lo, hi, _ = orc.interval_op(int(op[i]), A[0], A[1], B[0], B[1], float(K))
            out = (lo, hi)
        slots[int(o[i])] = out
    raise AssertionError("no end clause")


def live_choices(mpr, words, dec_l, dec_r):
    """the min / max clauses still on the tape a tile below walks: not decided above, and reached from the result with the decided
    clauses' other operands cut off"""
    OP = mpr.OP
    minmax = {OP["MIN_LHS_IMM"], OP["MIN_LHS_RHS"], OP["MAX_LHS_IMM"], OP["MAX_LHS_RHS"]}
    imm_forms = {OP[n] for n in OP if n.endswith("_IMM") or n == "COPY_IMM"}
    op, o, l, r, _ = tape_fields(words)
    cur, ldef, rdef, which = {}, {}, {}, {}
    k = 0
    end = None
    for i in range(1, len(words)):
        if op[i] == 0:
            end = i
            break
        ldef[i] = cur.get(int(l[i])) if l[i] else None
        rdef[i] = cur.get(int(r[i])) if r[i] else None
        if op[i] in minmax:
            which[i] = k
            k += 1
        cur[int(o[i])] = i
    seen, stack = set(), [cur.get(int(o[end]))]
    while stack:
        c = stack.pop()
        if c is None or c in seen:
            continue
        seen.add(c)
        kk = which.get(c)
        cut_l = kk is not None and dec_r >> kk & 1
        cut_r = kk is not None and dec_l >> kk & 1
        if not cut_l:
            stack.append(ldef[c])
        if not cut_r:
            stack.append(rdef[c])
    return [which[c] for c in sorted(which) if c in seen and not ((dec_l | dec_r) >> which[c] & 1)]


def tiles(rng, n, weird=False):
    """n random tiles of a 3-D frame (centres in the view cube, sides 2 / 16 .. 2 / 256), as (lo, hi) float32 pairs per axis"""
    out = []
    for _ in range(3):
        c = rng.uniform(-1, 1, n)
        h = rng.choice([1 / 16, 1 / 64, 1 / 256], n)
        lo, hi = (c - h).astype(np.float32), (c + h).astype(np.float32)
        if weird:
            k = n // 4
            special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 3e38, -3e38, 1e-38, 1.0, -1.0], dtype=np.float32)
            lo[:k] = rng.choice(special, k)
            hi[:k] = rng.choice(special, k)
        out.append((lo, hi))
    return out


def oracle_call(mpr, orc):
    def call(routine, a_lo, a_hi, b_lo, b_hi):
        lo, hi, _ = orc.interval_op(mpr.OP[ROUTINE_OP[routine]], a_lo, a_hi, b_lo, b_hi, 0.0)
        return lo, hi
    return call


def run_code(mpr, orc, lines, sizes, x, y, z, dec_l=0, dec_r=0, rng=None, perturb=False):
    n = len(x[0])
    emu = Emu(lines, sizes, n, rng=rng, perturb=perturb, call=oracle_call(mpr, orc))
    for a, (lo, hi) in enumerate((x, y, z)):
        emu.v[2 * a] = f2u(lo)
        emu.v[2 * a + 1] = f2u(hi)
    for r in (56, 57, 58, 59):
        emu.v[r] = 0
    emu.uni[72] = dec_l
    emu.uni[74] = dec_r
    how = emu.run()
    assert how == "done", how
    dl = emu.v[56].astype(np.uint64) | (emu.v[57].astype(np.uint64) << np.uint64(32))
    dr = emu.v[58].astype(np.uint64) | (emu.v[59].astype(np.uint64) << np.uint64(32))
    return u2f(emu.v[36]), u2f(emu.v[37]), dl, dr, emu


def same_bits(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return (f2u(a) == f2u(b)) | (np.isnan(a) & np.isnan(b))


def parent_decisions(mpr, orc, words, rng, tries=40):
    """decisions a real parent tile makes on this tape (so that imposing them is what a frame does), and children inside it"""
    for _ in range(tries):
        c = rng.uniform(-0.9, 0.9, 3)
        h = rng.choice([1 / 4, 1 / 8, 1 / 16])
        box = [(np.array([ci - h], np.float32), np.array([ci + h], np.float32)) for ci in c]
        _, ch = oracle_walk(mpr, orc, words, *box)
        dl = sum(1 << k for k, a in enumerate(ch) if a[0] == 1)
        dr = sum(1 << k for k, a in enumerate(ch) if a[0] == 2)
        if dl | dr:
            return dl, dr, c, h
    return 0, 0, np.zeros(3), 0.25


def children(rng, c, h, n):
    out = []
    for a in range(3):
        cc = rng.uniform(c[a] - h * 0.75, c[a] + h * 0.75, n)
        out.append(((cc - h / 4).astype(np.float32), (cc + h / 4).astype(np.float32)))
    return out


MODELS = ["bear", "smooth", "two_spheres", "ring", "sphere"]


@pytest.mark.parametrize("name", MODELS + ["trig"])
@pytest.mark.parametrize("kind", [FIRST, BELOW, GUARDED])
def test_exact_code_equals_the_oracle_bit_for_bit(mpr, orc, tapes, name, kind):
    words = [int(w) for w in tapes(name).data]
    lines, sizes, info = checked_code(mpr, words, kind, False)
    rng = np.random.default_rng(11 + kind)
    for trial in range(3):
        if kind == FIRST:
            dl = dr = 0
            x, y, z = tiles(rng, 96, weird=trial == 2)
        else:
            dl, dr, c, h = parent_decisions(mpr, orc, words, rng)
            x, y, z = children(rng, c, h, 96)
        (elo, ehi), choices = oracle_walk(mpr, orc, words, x, y, z, dl, dr)
        lo, hi, cl, cr, emu = run_code(mpr, orc, lines, sizes, x, y, z, dl, dr, rng)
        assert not emu.hazards, emu.hazards[:3]
        assert same_bits(lo, elo).all() and same_bits(hi, ehi).all()
        for k in live_choices(mpr, words, dl, dr):
            assert ((cl >> np.uint64(k) & np.uint64(1)) == (choices[k] == 1)).all(), (name, k)
            assert ((cr >> np.uint64(k) & np.uint64(1)) == (choices[k] == 2)).all(), (name, k)


@pytest.mark.parametrize("name", MODELS)
@pytest.mark.parametrize("kind", [FIRST, BELOW, GUARDED])
def test_loose_code_encloses_the_oracle_and_decides_no_more(mpr, orc, tapes, name, kind):
    words = [int(w) for w in tapes(name).data]
    lines, sizes, info = checked_code(mpr, words, kind, True)
    rng = np.random.default_rng(23 + kind)
    clean = 0
    for trial in range(4):
        if kind == FIRST:
            dl = dr = 0
            x, y, z = tiles(rng, 128)
        else:
            dl, dr, c, h = parent_decisions(mpr, orc, words, rng)
            x, y, z = children(rng, c, h, 128)
        (elo, ehi), choices = oracle_walk(mpr, orc, words, x, y, z, dl, dr)
        lo, hi, cl, cr, emu = run_code(mpr, orc, lines, sizes, x, y, z, dl, dr, rng, perturb=True)
        assert not emu.hazards, emu.hazards[:3]
        ok = ~emu.bad                                             # the lanes that would not ask for the exact walk
        clean += int(ok.sum())
        assert (lo[ok] <= elo[ok]).all() and (hi[ok] >= ehi[ok]).all(), name
        assert not np.isnan(lo[ok]).any() and not np.isnan(hi[ok]).any()
        for k in live_choices(mpr, words, dl, dr):
            assert not ((cl >> np.uint64(k) & np.uint64(1)).astype(bool) & ok & (choices[k] != 1)).any(), (name, k)
            assert not ((cr >> np.uint64(k) & np.uint64(1)).astype(bool) & ok & (choices[k] != 2)).any(), (name, k)
        # ... and not much wider: the classification must barely move
        width = np.maximum(ehi[ok] - elo[ok], np.float32(1e-6))
        assert np.median((hi[ok] - lo[ok]) / width) < 1.01
    assert clean > 300, "the loose walk asks for the exact one nearly everywhere"


def test_every_opcode_through_both_kinds_of_code(mpr, orc):
    """one-clause tapes: every opcode x operand classes; exact: equal bits, loose: an enclosure wherever it does not ask for the redo"""
    OP = mpr.OP
    rng = np.random.default_rng(5)
    n = 256

    def clause(op, out=0, lhs=0, rhs=0, imm=0):
        return op | out << 8 | lhs << 16 | rhs << 24 | imm << 32

    unary = ["SQUARE_LHS", "SQRT_LHS", "NEG_LHS", "SIN_LHS", "COS_LHS", "ASIN_LHS", "ACOS_LHS", "ATAN_LHS", "EXP_LHS", "ABS_LHS", "LOG_LHS", "COPY_LHS"]
    binary = ["ADD_LHS_RHS", "MUL_LHS_RHS", "MIN_LHS_RHS", "MAX_LHS_RHS", "SUB_LHS_RHS", "DIV_LHS_RHS", "COPY_RHS"]
    immf = ["ADD_LHS_IMM", "MUL_LHS_IMM", "MIN_LHS_IMM", "MAX_LHS_IMM", "SUB_LHS_IMM", "SUB_IMM_RHS", "DIV_LHS_IMM", "DIV_IMM_RHS", "COPY_IMM"]
    imms = [0x40490FDB, 0xC0490FDB, 0x3F000000, 0xBE4CCCCD, 0x41200000, 0x40000000]
    cases = [(o, 0) for o in unary + binary] + [(o, k) for o in immf for k in imms]
    for oname, K in cases:
        rhs_only = oname in ("SUB_IMM_RHS", "DIV_IMM_RHS", "COPY_RHS")
        c = clause(OP[oname], 4, 0 if oname in ("SUB_IMM_RHS", "DIV_IMM_RHS", "COPY_IMM") else 1, 2 if (oname in binary or rhs_only) else 0, K)
        words = [clause(0, 1, 2, 3), c, clause(0, 4)]
        for loose in (False, True):
            g = generate(mpr, words, FIRST, loose)
            if loose and oname in ("ASIN_LHS", "ACOS_LHS", "ATAN_LHS"):
                assert g is None
                continue
            lines, sizes, _ = checked_code(mpr, words, FIRST, loose)
            for cls in range(3):
                # ordinary intervals of either sign; narrow ones near zero; wide ones that straddle it
                if cls == 0:
                    c0 = rng.uniform(-3, 3, (3, n)); h = rng.uniform(0, 0.5, (3, n))
                elif cls == 1:
                    c0 = rng.uniform(-1e-3, 1e-3, (3, n)); h = rng.uniform(0, 1e-3, (3, n))
                else:
                    c0 = rng.uniform(-1, 1, (3, n)); h = rng.uniform(1, 40, (3, n))
                box = [((c0[a] - h[a]).astype(np.float32), (c0[a] + h[a]).astype(np.float32)) for a in range(3)]
                if not loose and cls == 0:
                    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, 3e38, -3e38], dtype=np.float32)
                    for a in range(3):
                        box[a][0][:64] = rng.choice(special, 64)
                        box[a][1][:64] = rng.choice(special, 64)
                (elo, ehi), choices = oracle_walk(mpr, orc, words, *box)
                lo, hi, cl, cr, emu = run_code(mpr, orc, lines, sizes, *box, rng=rng, perturb=loose)
                assert not emu.hazards, (oname, emu.hazards[:2])
                if not loose:
                    assert same_bits(lo, elo).all() and same_bits(hi, ehi).all(), oname
                    if choices:
                        assert ((cl & np.uint64(1)) == (choices[0] == 1)).all() and ((cr & np.uint64(1)) == (choices[0] == 2)).all(), oname
                else:
                    ok = ~emu.bad
                    assert (lo[ok] <= elo[ok]).all() and (hi[ok] >= ehi[ok]).all(), (oname, hex(K), cls)
                    if oname not in ("SQRT_LHS", "LOG_LHS", "DIV_LHS_RHS", "DIV_IMM_RHS", "EXP_LHS"):
                        assert ok.all(), (oname, cls)            # these have no domain to leave at these magnitudes
                    if choices:
                        assert not ((cl & np.uint64(1)).astype(bool) & ok & (choices[0] != 1)).any()
                        assert not ((cr & np.uint64(1)).astype(bool) & ok & (choices[0] != 2)).any()


def test_the_schedule_is_shorter_than_the_tape_order(mpr, tapes):
    words = [int(w) for w in tapes("bear").data]
    for loose in (False, True):
        in_order = generate(mpr, words, FIRST, loose, window=1)[2]
        scheduled = generate(mpr, words, FIRST, loose)[2]
        assert scheduled["window"] > 1
        assert scheduled["est_cycles"] < 0.8 * in_order["est_cycles"]
        assert scheduled["max_vgprs"] <= (54 if loose else 96)
    # the loose walk: no calls, no branches but the redo at the end, a third of the instructions of round 4's walk
    lines = generate(mpr, words, FIRST, True)[1]
    assert not any(l.startswith("s_swappc") for l in lines)
    assert sum(l.startswith("s_cbranch") for l in lines) == 1
    assert len(lines) < 2700
    # ... and no vector register beyond v63: the kernel around it runs six wavefronts per SIMD (interval_gen.hpp: IGEN_LEAN_VGPRS)
    assert max(int(r) for l in lines for r in re.findall(r"\bv(\d+)", l)) <= 63
    # tapes the loose arithmetic does not take (asin / acos / atan: the exact routines only)
    assert generate(mpr, [int(w) for w in tapes("trig").data], FIRST, True) is None


@pytest.mark.parametrize("name", ["architecture", "hello_world", "bear", "smooth", "two_spheres", "many_slots"])
def test_masks_code_of_a_first_stage_encloses_the_oracle_and_records_its_choices(mpr, orc, tapes, name):
    """IW_FIRST_MASKS (csrc/interval_gen.hpp): the loose first-stage walk of tapes beyond 24 slots / 64 min / max clauses — architecture:
    93 and 488 — with the choices recorded as the interpreter records them (16 bytes per min / max clause: the lanes that chose the
    lhs, the lanes that chose the rhs), for the interpreter's backward walk.  The assembler's words; no missing wait state; the
    enclosures hold the oracle's; a lane the code records as decided the oracle decides the same way; s[78:79] = the lanes that
    decided anything."""
    words = [int(w) for w in tapes(name).data]
    g = generate(mpr, words, MASKS, True)
    if name == "many_slots":
        assert g is None            # 150 values alive at once: 300 registers
        return
    lines, sizes, info = checked_code(mpr, words, MASKS, True)
    assert any(l.startswith("ds_write_b128 v60, v[56:59]") for l in lines)
    nch = info["nchoices"]
    assert sum(l.startswith("ds_write_b128") for l in lines) == (nch + 63) // 64
    if name == "architecture":
        assert nch == 488 and 150 < info["max_vgprs"] <= 243 and info["instructions"] < 7500, info
    rng = np.random.default_rng(5)
    clean = 0
    for trial in range(2):
        x, y, z = tiles(rng, 96)
        (elo, ehi), choices = oracle_walk(mpr, orc, words, x, y, z)
        n = len(x[0])
        emu = Emu(lines, sizes, n, rng=rng, perturb=True, call=oracle_call(mpr, orc))
        for a, (lo_, hi_) in enumerate((x, y, z)):
            emu.v[2 * a] = f2u(lo_)
            emu.v[2 * a + 1] = f2u(hi_)
        assert emu.run() == "done"
        assert not emu.hazards, emu.hazards[:3]
        lo, hi = u2f(emu.v[36]), u2f(emu.v[37])
        ok = ~emu.mask[40]                                        # s[40:41]: the lanes that ask for the exact walk
        clean += int(ok.sum())
        assert (lo[ok] <= elo[ok]).all() and (hi[ok] >= ehi[ok]).all(), name
        assert sorted(emu.choice_masks)[:nch] == list(range(nch))
        anyone = np.zeros(n, dtype=bool)
        for k in range(nch):
            ml, mr = emu.choice_masks[k]
            assert not (ml & mr).any()
            assert not (ml & ok & (choices[k] != 1)).any() and not (mr & ok & (choices[k] != 2)).any(), (name, k)
            anyone |= ml | mr
        assert (emu.mask[78] == anyone).all()
        # ... and it decides nearly everything the oracle decides
        decided_o = sum(int(((c != 0) & ok).sum()) for c in choices)
        decided_g = sum(int(((emu.choice_masks[k][0] | emu.choice_masks[k][1]) & ok).sum()) for k in range(nch))
        assert decided_g >= 0.98 * decided_o, (decided_g, decided_o)
    assert clean > 100


# ---- tight code (round 6): a second enclosure of the result, with sin / cos enclosed by their monotone pieces ----
def float_walk(mpr, orc, words, px, py, pz, dec_l=0, dec_r=0):
    """the float pass's value of the tape at points (float32 arrays), with what was decided above applied (a decided min / max is the
    chosen operand: the tape the float pass walks)"""
    OP = mpr.OP
    minmax = {OP["MIN_LHS_IMM"], OP["MIN_LHS_RHS"], OP["MAX_LHS_IMM"], OP["MAX_LHS_RHS"]}
    op, o, l, r, imm = tape_fields(words)
    slots = {int(o[0]): px, int(l[0]): py, int(r[0]): pz}
    zero = np.zeros(len(px), np.float32)
    k = 0
    for i in range(1, len(words)):
        if op[i] == 0:
            return slots[int(o[i])]
        A = slots.get(int(l[i]), zero)
        B = slots.get(int(r[i]), zero)
        K = u2f(np.array([imm[i]], dtype=np.uint32))[0]
        if op[i] in minmax:
            if dec_l >> k & 1:
                out = A
            elif dec_r >> k & 1:
                out = B if r[i] != 0 else np.full(len(px), K, np.float32)
            else:
                out = orc.float_op(int(op[i]), A, B, float(K))
            k += 1
        else:
            out = orc.float_op(int(op[i]), A, B, float(K))
        slots[int(o[i])] = out
    raise AssertionError("no end clause")


def small_tiles(rng, n, c=None, h0=None, sizes=(1 / 64, 1 / 256, 1 / 512)):
    out = []
    for a in range(3):
        cc = rng.uniform(-1, 1, n) if c is None else rng.uniform(c[a] - h0 * 0.9, c[a] + h0 * 0.9, n)
        h = rng.choice(sizes, n)
        out.append(((cc - h).astype(np.float32), (cc + h).astype(np.float32)))
    return out


@pytest.mark.parametrize("name", ["bear", "trig_blend"])
@pytest.mark.parametrize("kind", [FIRST, BELOW, GUARDED])
def test_tight_code_leaves_the_walk_as_it_is_and_its_second_enclosure_holds_the_float_values(mpr, orc, tapes, name, kind):
    """Tight code (csrc/interval_gen.hpp): the assembler's words; no missing wait state; everything the loose code leaves — the
    result, every lane's decisions, the lanes that ask for the exact walk — bit for bit what the code without the second result
    leaves; the second enclosure lies inside the first and holds the float pass's value (the oracle's float routines, clause by
    clause, on the tape with what was decided above applied) at the corners, the centre and random points of every tile; and on
    tiles the size of a frame's smallest it decides a good part of what the wide one leaves open."""
    words = [int(w) for w in tapes(name).data]
    lines_t, sizes_t, info_t = checked_code(mpr, words, kind, True, tight=True)
    lines_w, sizes_w, info_w = checked_code(mpr, words, kind, True)
    assert any(l.startswith("v_sin_f32") or l.startswith("v_cos_f32") for l in lines_t) and not any("v_cos_f32" in l or "v_sin_f32" in l for l in lines_w)
    assert max(int(r) for l in lines_t for r in re.findall(r"\bv(\d+)", l)) <= 79
    rng = np.random.default_rng(31 + kind)
    decided_w = decided_t = total = 0
    for trial in range(4):
        if kind == FIRST:
            dl = dr = 0
            x, y, z = small_tiles(rng, 128)
        else:
            dl, dr, c, h = parent_decisions(mpr, orc, words, rng)
            x, y, z = small_tiles(rng, 128, c, h)
        # the same walk: no perturbation, so that the two codes see the same instruction results
        lo_w, hi_w, cl_w, cr_w, emu_w = run_code(mpr, orc, lines_w, sizes_w, x, y, z, dl, dr, np.random.default_rng(1))
        lo, hi, cl, cr, emu = run_code(mpr, orc, lines_t, sizes_t, x, y, z, dl, dr, np.random.default_rng(1))
        assert not emu.hazards, emu.hazards[:3]
        assert same_bits(lo, lo_w).all() and same_bits(hi, hi_w).all() and (emu.bad == emu_w.bad).all()
        live = np.uint64(sum(1 << k for k in live_choices(mpr, words, dl, dr)))      # (a clause a guard jumps over records whatever its registers hold)
        assert ((cl & live) == (cl_w & live)).all() and ((cr & live) == (cr_w & live)).all()
        # the second enclosure, the instructions moved by what the chip may move them
        lo, hi, cl, cr, emu = run_code(mpr, orc, lines_t, sizes_t, x, y, z, dl, dr, rng, perturb=True)
        tlo, thi = u2f(emu.v[38]), u2f(emu.v[39])
        ok = ~emu.bad
        assert not np.isnan(tlo[ok]).any() and not np.isnan(thi[ok]).any()
        assert (tlo[ok] >= lo[ok] - 1e-5 * np.abs(lo[ok])).all() and (thi[ok] <= hi[ok] + 1e-5 * np.abs(hi[ok])).all()
        for s in range(12):
            if s < 8:
                f = [np.float32((s >> a) & 1) for a in range(3)]
            elif s == 8:
                f = [np.float32(0.5)] * 3
            else:
                f = [rng.uniform(0, 1, len(x[0])).astype(np.float32) for _ in range(3)]
            pts = [np.minimum(np.maximum((b[0] + (b[1] - b[0]) * f[a]).astype(np.float32), b[0]), b[1]) for a, b in enumerate((x, y, z))]
            v = float_walk(mpr, orc, words, *pts, dl, dr)
            good = ok & ~np.isnan(v)
            assert (v[good] >= tlo[good]).all() and (v[good] <= thi[good]).all(), (name, kind, s, int((~((v >= tlo) & (v <= thi)) & good).sum()))
        total += int(ok.sum())
        decided_w += int((((lo > 0) | (hi < 0)) & ok).sum())
        decided_t += int((((tlo > 0) | (thi < 0)) & ok).sum())
    assert decided_t >= decided_w
    if name == "bear" and kind == FIRST:
        assert decided_t - decided_w >= (total - decided_w) // 3, (decided_w, decided_t, total)      # (of what the wide one leaves open)


def test_tight_code_only_where_there_is_something_to_tighten(mpr, tapes):
    assert generate(mpr, [int(w) for w in tapes("smooth").data], FIRST, True, tight=True) is None      # no sin / cos
    assert generate(mpr, [int(w) for w in tapes("bear").data], FIRST, False, tight=True) is None        # exact code has one result


def test_tapes_made_side_by_side_carry_the_same_code(mpr, tapes):
    """A tape's nine scheduled walks are built side by side (tile_gen.cpp: build_tape_code, std::async) on per-thread scratch
    (interval_gen.cpp: schedule_region); callers may make tapes from several threads at once.  Eight threads, three models: every tape
    renders through the same code — the generator's output for its clauses, hashed, is what one thread alone produces."""
    import hashlib
    import threading

    def digest(words):
        h = hashlib.sha1()
        for kind in (FIRST, BELOW, GUARDED):
            for loose, tight in ((False, False), (True, False), (True, True)):
                g = generate(mpr, words, kind, loose, tight=tight)
                h.update(b"none" if g is None else np.asarray(g[0], dtype=np.uint32).tobytes())
        return h.hexdigest()

    names = ["bear", "hello_world", "trig_blend"]
    alone = {n: digest(tapes(n).data) for n in names}
    trees = {n: (mpr.model(n) if n != "trig_blend" else None) for n in names}
    results, errors = [], []

    def work(k):
        try:
            for rep in range(3):
                n = names[(k + rep) % len(names)]
                t = mpr.Tape(trees[n]) if trees[n] is not None else mpr.Tape(tapes(n).data)
                results.append((n, digest(t.data)))
        except Exception as e:          # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert len(results) == 24 and all(d == alone[n] for n, d in results)
