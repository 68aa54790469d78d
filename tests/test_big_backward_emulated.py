"""The generated backward walk of many-slot tapes (csrc/tile_gen.hpp: tile_gen_build_big_backward) run on the CPU: its disassembly
interpreted lane by lane for 64 tiles with random choices, against the reference's Algorithm 2 (src/context.cu:351-458) restated in a few
lines of Python — the clause sequences the lanes' tapes end up with, chunk links included.  (On the chip the same code is held against the
interpreter's backward walk and the oracle: tests/test_gpu_big_first_stage.py.)"""
import ctypes
import os
import re
import struct
import subprocess

import numpy as np
import pytest

LLVM_MC = "/opt/rocm/lib/llvm/bin/llvm-mc"
pytestmark = pytest.mark.skipif(not os.path.exists(LLVM_MC), reason="llvm-mc not found")
CHUNK = 64


def disassembled(mpr, words):
    arr = np.array(words, dtype=np.uint64)
    buf = (ctypes.c_uint32 * 400000)()
    n = mpr.lib().mpr_test_tile_gen(arr.ctypes.data, len(arr), 6, buf, 400000)
    assert n > 0
    text = ",".join("0x%02x" % b for d in buf[:n] for b in struct.pack("<I", d))
    r = subprocess.run([LLVM_MC, "-arch=amdgcn", "-mcpu=gfx950", "-disassemble"], input=text.encode(), capture_output=True, check=True)
    assert not r.stderr.strip(), r.stderr.decode()[:500]
    return [" ".join(l.split()) for l in r.stdout.decode().splitlines() if l.strip() and not l.strip().startswith(".")]


class Lanes:
    """the few instructions the walk is made of, 64 lanes wide"""

    def __init__(self, lines, choices, pool_words, run_chunks):
        self.lines = lines
        self.v = np.zeros((256, 64), dtype=np.uint32)
        self.vcc = np.zeros(64, dtype=bool)
        self.exec = np.ones(64, dtype=bool)
        self.choices = choices                      # [nch][2] 64-bit lane masks
        self.pool = np.zeros(pool_words, dtype=np.uint64)
        self.pending = None                         # the ds_read in flight
        self.run_chunks = run_chunks

    def src(self, o):
        o = o.strip()
        if o.startswith("v"):
            return self.v[int(o[1:])].copy()
        if o.startswith("0x"):
            return np.full(64, int(o, 16), dtype=np.uint32)
        if "." in o:                                # a constant the disassembler prints as the float it encodes
            return np.full(64, struct.unpack("<I", struct.pack("<f", float(o)))[0], dtype=np.uint32)
        return np.full(64, int(o) & 0xFFFFFFFF, dtype=np.uint32)

    def put(self, d, val):
        r = int(d[1:])
        self.v[r] = np.where(self.exec, val.astype(np.uint32), self.v[r])

    def chunk_full(self):
        """tile_gen_asm.hpp: L_chunk — lanes whose chunk is full move to the next one of their run and write the two links"""
        full = self.v[61] == self.v[62]
        old_first = self.v[62].copy()
        new_first = self.v[62] + np.uint32(CHUNK)
        out = full & (new_first >= self.v[63])
        assert not out.any(), "the test sizes the runs so that nobody runs out"
        for lane in np.nonzero(full)[0]:
            nf = int(new_first[lane])
            self.pool[nf + 63] = np.uint64(1) | (np.uint64(0xFFFFFF81) << np.uint64(32))
            self.pool[int(old_first[lane])] = np.uint64(1) | (np.uint64(127) << np.uint64(32))
        self.v[62] = np.where(full, new_first, self.v[62])
        self.v[61] = np.where(full, new_first + np.uint32(62), self.v[61])

    def run(self):
        i = 0
        n = len(self.lines)
        while i < n:
            name, _, rest = self.lines[i].partition(" ")
            ops = [o.strip() for o in rest.split(",")] if rest else []
            i += 1
            if name == "s_setpc_b64":
                assert ops == ["s[38:39]"] and i == n
                return
            if name == "ds_read_b128":
                m = re.fullmatch(r"v75(?: offset:(\d+))?", ops[1])
                assert ops[0] == "v[66:69]" and m
                self.pending = int(m.group(1) or 0) // 16
            elif name == "s_waitcnt":
                m1, m2 = self.choices[self.pending]
                self.v[66][:] = m1 & 0xFFFFFFFF; self.v[67][:] = m1 >> 32; self.v[68][:] = m2 & 0xFFFFFFFF; self.v[69][:] = m2 >> 32
            elif name == "v_bfe_u32":
                a, off, w = self.src(ops[1]), self.src(ops[2]), self.src(ops[3])
                assert (w == 1).all()
                self.put(ops[0], (a >> (off & 31)) & 1)
            elif name == "v_cndmask_b32_e64":
                assert ops[3] == "s[64:65]"
                hi = np.arange(64) >= 32
                self.put(ops[0], np.where(hi, self.src(ops[2]), self.src(ops[1])))
            elif name == "v_cndmask_b32_e32":
                self.put(ops[0], np.where(self.vcc, self.src(ops[2]), self.src(ops[1])))
            elif name in ("v_sub_u32_e32", "v_add_u32_e32", "v_and_b32_e32", "v_or_b32_e32", "v_xor_b32_e32", "v_lshlrev_b32_e32"):
                a, b = self.src(ops[1]), self.src(ops[2])
                self.put(ops[0], {"v_sub_u32_e32": a - b, "v_add_u32_e32": a + b, "v_and_b32_e32": a & b, "v_or_b32_e32": a | b,
                                  "v_xor_b32_e32": a ^ b, "v_lshlrev_b32_e32": b << (a & 31)}[name])
            elif name == "v_lshl_or_b32":
                self.put(ops[0], (self.src(ops[1]) << (self.src(ops[2]) & 31)) | self.src(ops[3]))
            elif name == "v_mov_b32_e32":
                self.put(ops[0], self.src(ops[1]))
            elif name == "v_cmp_eq_u32_e32":
                self.vcc = self.src(ops[1]) == self.src(ops[2])
            elif name == "v_cmp_ne_u32_e32":
                self.vcc = self.src(ops[1]) != self.src(ops[2])
            elif name == "s_cbranch_vccz":
                if not self.vcc.any():
                    i += int(ops[0])
            elif name == "s_swappc_b64":
                assert ops == ["s[36:37]", "s[62:63]"]
                self.chunk_full()
            elif name == "s_mov_b64":
                assert ops[0] == "exec"
                self.exec = self.vcc.copy() if ops[1] == "vcc" else np.ones(64, dtype=bool)
            elif name == "global_store_dwordx2":
                assert ops == ["v44", "v[46:47]", "s[76:77]"]
                for lane in np.nonzero(self.exec)[0]:
                    self.pool[int(self.v[44][lane]) // 8] = np.uint64(self.v[46][lane]) | (np.uint64(self.v[47][lane]) << np.uint64(32))
            elif name == "s_nop":
                pass
            else:
                raise AssertionError("instruction " + self.lines[i - 1])


def reference_push(mpr, words, choice_of, out_slot):
    """Algorithm 2 for one tile: the clause words of its shortened tape, from the end backwards (without head and end)"""
    OP = mpr.OP
    minmax = {OP["MIN_LHS_IMM"], OP["MIN_LHS_RHS"], OP["MAX_LHS_IMM"], OP["MAX_LHS_RHS"]}
    end = next(i for i in range(1, len(words)) if words[i] & 0xFF == 0)
    active = {out_slot}
    out = []
    k = sum(1 for i in range(1, end) if words[i] & 0xFF in minmax)
    for i in range(end - 1, 0, -1):
        w = words[i]
        op, o, l, r = w & 0xFF, (w >> 8) & 0xFF, (w >> 16) & 0xFF, (w >> 24) & 0xFF
        ch = 0
        if op in minmax:
            k -= 1
            ch = choice_of(k)
        if o not in active:
            continue
        active.discard(o)
        if op in minmax and ch == 1:
            active.add(l)
            if l != o:
                out.append((w & ~0xFF) | OP["COPY_LHS"])
        elif op in minmax and ch == 2:
            if r:
                active.add(r)
                if r != o:
                    out.append((w & ~0xFF) | OP["COPY_RHS"])
            else:
                out.append((w & ~0xFF) | OP["COPY_IMM"])
        else:
            if l:
                active.add(l)
            if r:
                active.add(r)
            out.append(w)
    return out


@pytest.mark.parametrize("name,undecided", [("architecture", 0.15), ("involute_gear_3d", 0.15), ("hello_world", 0.15), ("architecture", 0.92), ("prospero", 0.5)])
def test_generated_backward_walk_is_algorithm_2(mpr, tapes, name, undecided):
    """(undecided = share of the min / max clauses a lane leaves undecided: 0.92 keeps nearly the whole tape — twenty chunks per lane)"""
    words = [int(w) for w in tapes(name).data]
    lines = disassembled(mpr, words)
    OP = mpr.OP
    minmax = {OP["MIN_LHS_IMM"], OP["MIN_LHS_RHS"], OP["MAX_LHS_IMM"], OP["MAX_LHS_RHS"]}
    end = next(i for i in range(1, len(words)) if words[i] & 0xFF == 0)
    nch = sum(1 for i in range(1, end) if words[i] & 0xFF in minmax)
    rng = np.random.default_rng(3)
    # per choice: the lanes that chose the lhs / the rhs (most decide: tapes shorten a lot), disjoint
    pick = rng.choice([0, 1, 2], size=(nch, 64), p=[undecided, (1 - undecided) * 0.55, (1 - undecided) * 0.45])
    masks = [(int(sum(1 << j for j in range(64) if pick[k, j] == 1)), int(sum(1 << j for j in range(64) if pick[k, j] == 2))) for k in range(nch)]
    run_chunks = (end + 1 + 61) // 62 + 1
    pushing = rng.random(64) < 0.8
    emu = Lanes(lines, masks, 64 * run_chunks * CHUNK + 1024, run_chunks)
    out_slot = (words[end] >> 8) & 0xFF
    for lane in range(64):
        first = 512 + lane * run_chunks * CHUNK
        emu.v[62][lane] = first
        emu.v[63][lane] = first + run_chunks * CHUNK
        emu.v[61][lane] = first + CHUNK - 1          # the end clause sits in word 63 of the first chunk
        if pushing[lane]:
            emu.v[60 if out_slot < 32 else 64 if out_slot < 64 else 65][lane] = 1 << (out_slot & 31)
    emu.v[74] = np.arange(64, dtype=np.uint32) & 31
    emu.run()
    for lane in range(64):
        want = reference_push(mpr, words, lambda k: int(pick[k, lane]), out_slot) if pushing[lane] else []
        # read the lane's words back in the order they were written: down from word 62 of its first chunk, on through the links
        got = []
        first = 512 + lane * run_chunks * CHUNK
        pos, chunk = first + CHUNK - 2, first
        last = int(emu.v[61][lane])
        while True:
            if chunk == int(emu.v[62][lane]) and pos < last:
                break
            if pos == chunk:                          # word 0: the link forward (a JUMP of +127 from the previous chunk's view)
                assert int(emu.pool[pos]) == 1 | (127 << 32)
                chunk += CHUNK
                assert int(emu.pool[chunk + 63]) == 1 | (0xFFFFFF81 << 32)
                pos = chunk + 62
                continue
            got.append(int(emu.pool[pos]))
            pos -= 1
        assert got == want, (name, lane, len(got), len(want))
        kept = sum(1 for w in want if (w & 0xFF) in minmax)
        assert int(emu.v[54][lane]) == kept          # min / max clauses the lane's tape keeps (the next stage's bound)
        assert int(emu.v[55][lane]) == 0
