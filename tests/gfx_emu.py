"""A lane-vectorised emulator of the gfx950 instructions csrc/interval_gen.cpp emits (TEST INFRASTRUCTURE).

It reads the generator's own assembler text (the text tests/test_interval_gen.py also hands to the ROCm assembler and compares,
byte for byte, with the words the library emits), so the emulator and the encoder are two independent readings of each
instruction.  Arithmetic is float32 in ROUND-UP mode, reproduced exactly from float64 products / two-sums; the hardware's
approximate instructions (v_exp_f32, v_log_f32, v_rcp_f32, v_sqrt_f32: within an ulp) are the correctly rounded value moved by a
random -1 / 0 / +1 ulp per lane when `perturb` is set, and flush denormals the way the chip does.  Calls (s_swappc_b64 through a
routine's SGPR pair) run the ORACLE's interval routine on v[36:39] and then trash every register the real routines may clobber.
"""
import re

import numpy as np

F32_MAX = np.float32(3.4028234663852886e38)
SIGN = np.uint32(0x80000000)
INLINE_F = {"0.5": 0x3F000000, "-0.5": 0xBF000000, "1.0": 0x3F800000, "-1.0": 0xBF800000, "2.0": 0x40000000, "-2.0": 0xC0000000,
            "4.0": 0x40800000, "-4.0": 0xC0800000, "0.15915494": 0x3E22F983}
FLOAT_OPS = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_min_f32", "v_max_f32", "v_max3_f32", "v_med3_f32", "v_fma_f32",
             "v_exp_f32", "v_log_f32", "v_rcp_f32", "v_sqrt_f32", "v_cndmask_b32", "v_sin_f32", "v_cos_f32", "v_floor_f32"}
TRIG_ABS_ERROR = 2.0 ** -19        # what v_sin_f32 / v_cos_f32 are moved by at most when `perturb` is set (measured on the chip: see
                                   # tests/test_gpu_primitives.py::test_tight_sin_cos_code_on_every_float)


def f2u(x):
    return np.asarray(x, dtype=np.float32).view(np.uint32)


def u2f(x):
    return np.asarray(x, dtype=np.uint32).view(np.float32)


def next_up(r):
    return np.nextafter(r, np.float32(np.inf), dtype=np.float32)


def ru_from_sum(s, e):
    """float32 round-up of the exact real s + e (float64 arrays; |e| at most half an ulp of s)"""
    with np.errstate(all="ignore"):
        r = s.astype(np.float32)
        diff = (s - r.astype(np.float64)) + e
        r = np.where(diff > 0, next_up(r), r)
        # round-up never overflows downwards: a finite negative sum beyond the range is the largest finite negative number
        r = np.where(np.isneginf(r) & np.isfinite(s), -F32_MAX, r)
        # an exact zero sum of opposite-signed zeros / cancelling terms is +0 (numpy's round-to-nearest agrees)
    return r.astype(np.float32)


def two_sum(a, b):
    with np.errstate(all="ignore"):
        s = a + b
        bb = s - a
        e = (a - (s - bb)) + (b - bb)
        e = np.where(np.isfinite(s), e, 0.0)
    return s, e


def ru_add(a, b):
    s, e = two_sum(a.astype(np.float64), b.astype(np.float64))
    return ru_from_sum(s, e)


def ru_mul(a, b):
    with np.errstate(all="ignore"):
        p = a.astype(np.float64) * b.astype(np.float64)
    return ru_from_sum(p, np.zeros_like(p))


def ru_fma(a, b, c):
    with np.errstate(all="ignore"):
        p = a.astype(np.float64) * b.astype(np.float64)
    s, e = two_sum(p, c.astype(np.float64))
    return ru_from_sum(s, e)


def key(x):
    """a total order on non-NaN floats as integers (-0 below +0)"""
    u = f2u(x).astype(np.int64)
    return np.where(u & 0x80000000, 0x80000000 - (u & 0x7FFFFFFF) - 1, 0x80000000 + u)


def fmax(a, b):
    an, bn = np.isnan(a), np.isnan(b)
    r = np.where(key(a) >= key(b), a, b)
    return np.where(an, b, np.where(bn, a, r)).astype(np.float32)


def fmin(a, b):
    an, bn = np.isnan(a), np.isnan(b)
    r = np.where(key(a) <= key(b), a, b)
    return np.where(an, b, np.where(bn, a, r)).astype(np.float32)


def flush(x):
    u = f2u(x)
    den = (u & 0x7F800000) == 0
    return u2f(np.where(den, u & 0x80000000, u))


class Emu:
    """registers: v[256][n] uint32, lane masks per scalar pair, the two decided words as integers"""

    def __init__(self, lines, sizes, n, rng=None, perturb=False, call=None):
        self.lines = [self.parse(l) for l in lines]
        self.pos = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
        self.at = {int(p): i for i, p in enumerate(self.pos[:-1])}
        self.n = n
        self.v = np.zeros((256, n), dtype=np.uint32)
        self.v[:] = 0x7FC0BAD0                     # whatever is read before it is written is a NaN
        self.mask = {}
        self.uni = {}                              # uniform 64-bit scalars (decided above)
        self.scc = False
        self.rng = rng or np.random.default_rng(0)
        self.perturb = perturb
        self.call = call
        self.bad = None
        self.executed = 0
        self.trans_written = {}                    # hazard bookkeeping: vgpr -> executed index of a transcendental write
        self.sgpr_valu_written = {}                # sgpr base -> executed index of a VALU write
        self.hazards = []
        self.lane_writes = {}                      # v_writelane_b32: vgpr -> lane -> (scalar pair, half, the mask written)
        self.choice_masks = {}                     # ds_write_b128 of v[56:59]: choice -> (lanes that chose the lhs, the rhs)

    @staticmethod
    def parse(line):
        name, _, rest = line.partition(" ")
        ops = [o.strip() for o in rest.split(",")] if rest.strip() else []
        return name, ops

    # ---- operands ----
    def src_bits(self, o, is_float):
        neg = o.startswith("-") and (o[1:2] in "v|" or o[1:].startswith("s"))
        if neg:
            o = o[1:]
        ab = o.startswith("|")
        if ab:
            o = o[1:-1]
        if o.startswith("v"):
            x = self.v[int(o[1:])].copy()
        elif o in INLINE_F:
            x = np.full(self.n, INLINE_F[o], dtype=np.uint32)
        elif o.startswith("0x"):
            x = np.full(self.n, int(o, 16), dtype=np.uint32)
        elif re.fullmatch(r"-?\d+", o):
            x = np.full(self.n, int(o) & 0xFFFFFFFF, dtype=np.uint32)
        else:
            raise ValueError("operand " + o)
        if ab:
            x = x & np.uint32(0x7FFFFFFF)
        if neg:
            x = x ^ SIGN
        return x

    def sreg(self, o):
        m = re.fullmatch(r"s\[(\d+):(\d+)\]", o)
        if m:
            return int(m.group(1))
        if o == "vcc":
            return 106
        raise ValueError("scalar pair " + o)

    def src_mask(self, o):
        if o == "0":
            return np.zeros(self.n, dtype=bool)
        if o == "-1" or o == "exec":
            return np.ones(self.n, dtype=bool)
        r = self.sreg(o)
        if r not in self.mask:
            raise ValueError("lane mask s[%d:%d] read before it is written" % (r, r + 1))
        return self.mask[r]

    def vreads(self, name, ops):
        if name.startswith("v_cmp"):
            return [o for o in ops[1:]]
        if name == "v_cndmask_b32":
            return ops[1:3]
        return ops[1:]

    def hazard_check(self, idx, name, ops):
        """the wait states the chip does not interlock (csrc/interval_gen.cpp: insert_wait_states)"""
        is_valu = name.startswith("v_")
        if not is_valu:
            return
        trans = name in ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32")
        for o in self.vreads(name, ops):
            m = re.fullmatch(r"-?\|?v(\d+)\|?", o)
            if m and not trans:
                w = self.trans_written.get(int(m.group(1)))
                if w is not None and self.executed - w < 1:
                    self.hazards.append((idx, "transcendental result read with no wait state: " + name + " " + ", ".join(ops)))
        if name == "v_writelane_b32":
            r = int(ops[1][1:]) & ~1
            w = self.sgpr_valu_written.get(r)
            if w is not None and self.executed - w < 2:
                self.hazards.append((idx, "scalar register read %d instruction(s) after a VALU wrote it: %s" % (self.executed - w, ", ".join(ops))))
        if name == "v_cndmask_b32" and len(ops) == 4:
            r = self.sreg(ops[3])
            w = self.sgpr_valu_written.get(r)
            if w is not None and self.executed - w < 2:
                self.hazards.append((idx, "lane mask read %d instruction(s) after a VALU wrote it: %s" % (self.executed - w, ", ".join(ops))))

    # ---- one walk; returns "done" / "redo" ----
    def run(self, max_steps=200000):
        i = 0
        steps = 0
        while True:
            steps += 1
            if steps > max_steps:
                raise RuntimeError("runaway code")
            name, ops = self.lines[i]
            nxt = i + 1
            self.hazard_check(i, name, ops)
            if name == "s_nop":
                self.executed += int(ops[0]) + 1
                i = nxt
                continue
            self.executed += 1
            if name == "s_setpc_b64":
                r = self.sreg(ops[0])
                return "done" if r == 38 else "redo" if r == 60 else "pc s%d" % r
            if name in ("s_branch", "s_cbranch_scc1", "s_cbranch_scc0"):
                off = int(ops[0])
                if off >= 32768:
                    off -= 65536
                take = name == "s_branch" or (name == "s_cbranch_scc1") == self.scc
                if take:
                    nxt = self.at[int(self.pos[i + 1]) + off]
                i = nxt
                continue
            if name == "s_swappc_b64":
                self.do_call(self.sreg(ops[1]))
                i = nxt
                continue
            if name.startswith("s_"):
                self.salu(name, ops)
            elif name == "v_writelane_b32":
                # one lane of a register takes half of a lane mask (the masks here are n lanes wide: kept whole, with the half's name)
                s_ = int(ops[1][1:])
                if (s_ & ~1) not in self.mask:
                    raise ValueError("lane mask s[%d:%d] read before it is written" % (s_ & ~1, (s_ & ~1) + 1))
                self.lane_writes.setdefault(int(ops[0][1:]), {})[int(ops[2])] = (s_ & ~1, s_ & 1, self.mask[s_ & ~1].copy())
            elif name == "ds_write_b128":
                m = re.fullmatch(r"v\[(\d+):(\d+)\](?: offset:(\d+))?", ops[1].strip())
                assert ops[0] == "v60" and m and (int(m.group(1)), int(m.group(2))) == (56, 59), ops
                group = int(m.group(3) or 0) // 1024
                for lane in range(64):
                    parts = [self.lane_writes.get(r, {}).get(lane) for r in (56, 57, 58, 59)]
                    if any(p is None for p in parts):
                        continue
                    assert parts[0][:2] == (parts[1][0], 0) and parts[1][1] == 1 and parts[2][:2] == (parts[3][0], 0) and parts[3][1] == 1, (lane, [p[:2] for p in parts])
                    assert (parts[0][2] == parts[1][2]).all() and (parts[2][2] == parts[3][2]).all()
                    self.choice_masks[group * 64 + lane] = (parts[0][2], parts[2][2])
            else:
                self.valu(name, ops)
            i = nxt

    def do_call(self, routine):
        a_lo, a_hi, b_lo, b_hi = (u2f(self.v[r].copy()) for r in (36, 37, 38, 39))
        lo, hi = self.call(routine, a_lo, a_hi, b_lo, b_hi)
        # what the routines may clobber: v0..v55 but the decisions' neighbours v56.., v64..v69; s0..s31, s40..s59, s92..s95, vcc
        for r in list(range(0, 56)) + list(range(64, 70)):
            self.v[r] = 0x7FC0DEAD
        self.v[40] = f2u(lo)
        self.v[41] = f2u(hi)
        for r in list(self.mask):
            if r < 32 or 40 <= r < 60 or 92 <= r < 96 or r == 106:
                del self.mask[r]
        self.scc = bool(self.rng.integers(0, 2))

    def salu(self, name, ops):
        if ops and name not in ("s_bitcmp1_b64", "s_cmp_lg_u64"):
            self.sgpr_valu_written.pop(self.sreg(ops[0]), None)         # a scalar instruction's result needs no wait
        if name == "s_mov_b64":
            self.mask[self.sreg(ops[0])] = self.src_mask(ops[1]).copy()
        elif name in ("s_and_b64", "s_or_b64", "s_andn2_b64", "s_orn2_b64", "s_xor_b64"):
            a, b = self.src_mask(ops[1]), self.src_mask(ops[2])
            r = {"s_and_b64": a & b, "s_or_b64": a | b, "s_andn2_b64": a & ~b, "s_orn2_b64": a | ~b, "s_xor_b64": a ^ b}[name]
            self.mask[self.sreg(ops[0])] = r
            self.scc = bool(r.any())
        elif name == "s_bitcmp1_b64":
            self.scc = bool(self.uni[self.sreg(ops[0])] >> int(ops[1]) & 1)
        elif name == "s_cselect_b64":
            a, b = self.src_mask(ops[1]), self.src_mask(ops[2])
            self.mask[self.sreg(ops[0])] = (a if self.scc else b).copy()
        elif name == "s_cmp_lg_u64":
            m = self.src_mask(ops[0])
            self.bad = m.copy()                     # (the loose walk's verdict; the caller looks at it lane by lane)
            self.scc = False if self.ignore_redo else bool(m.any())
        else:
            raise ValueError("scalar instruction " + name)

    ignore_redo = True

    def approx(self, exact64, x):
        with np.errstate(all="ignore"):
            r = exact64.astype(np.float32)
        if self.perturb:
            step = self.rng.integers(-1, 2, size=self.n)
            fin = np.isfinite(r) & (r != 0)
            up = next_up(r)
            dn = np.nextafter(r, np.float32(-np.inf), dtype=np.float32)
            r = np.where(fin & (step > 0), up, np.where(fin & (step < 0), dn, r))
        return flush(r)

    def valu(self, name, ops):
        base = name[:-4] if name.endswith("_e32") or name.endswith("_e64") else name
        fl = base in FLOAT_OPS or base.startswith("v_cmp")
        if base.startswith("v_cmp"):
            a = u2f(self.src_bits(ops[1], True))
            b = u2f(self.src_bits(ops[2], True))
            kind = base[len("v_cmp_"):-4]
            with np.errstate(all="ignore"):
                un = np.isnan(a) | np.isnan(b)
                table = {"lt": a < b, "le": a <= b, "gt": a > b, "ge": a >= b, "eq": a == b, "lg": (a < b) | (a > b), "o": ~un, "u": un,
                         "nlt": ~(a < b), "nle": ~(a <= b), "ngt": ~(a > b), "nge": ~(a >= b), "neq": ~(a == b), "nlg": ~((a < b) | (a > b))}
            if not base.endswith("_f32"):
                raise ValueError(base)
            r = self.sreg(ops[0])
            self.mask[r] = table[kind]
            self.sgpr_valu_written[r] = self.executed
            return
        d = int(ops[0][1:])
        trans = False
        if base == "v_mov_b32":
            r = self.src_bits(ops[1], False)
        elif base == "v_cndmask_b32":
            m = self.src_mask(ops[3])
            r = np.where(m, self.src_bits(ops[2], True), self.src_bits(ops[1], True))
        elif base in ("v_xor_b32", "v_and_b32", "v_or_b32", "v_add_u32", "v_sub_u32", "v_max_u32", "v_min_u32"):
            a, b = self.src_bits(ops[1], False), self.src_bits(ops[2], False)
            r = {"v_xor_b32": a ^ b, "v_and_b32": a & b, "v_or_b32": a | b, "v_add_u32": a + b, "v_sub_u32": a - b,
                 "v_max_u32": np.maximum(a, b), "v_min_u32": np.minimum(a, b)}[base]
        elif base == "v_lshl_or_b32":
            a, s, c = (self.src_bits(o, False) for o in ops[1:4])
            r = (a << (s & np.uint32(31))) | c
        elif base in ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_min_f32", "v_max_f32"):
            a, b = u2f(self.src_bits(ops[1], True)), u2f(self.src_bits(ops[2], True))
            if base == "v_add_f32":
                r = ru_add(a, b)
            elif base == "v_sub_f32":
                r = ru_add(a, -b)
            elif base == "v_subrev_f32":
                r = ru_add(b, -a)
            elif base == "v_mul_f32":
                r = ru_mul(a, b)
            elif base == "v_min_f32":
                r = fmin(a, b)
            else:
                r = fmax(a, b)
            r = f2u(r)
        elif base == "v_max3_f32":
            a, b, c = (u2f(self.src_bits(o, True)) for o in ops[1:4])
            r = f2u(fmax(fmax(a, b), c))
        elif base == "v_med3_f32":
            a, b, c = (u2f(self.src_bits(o, True)) for o in ops[1:4])
            anynan = np.isnan(a) | np.isnan(b) | np.isnan(c)
            med = fmax(fmin(a, b), fmin(fmax(a, b), c))
            r = f2u(np.where(anynan, fmin(fmin(a, b), c), med))
        elif base == "v_fma_f32":
            a, b, c = (u2f(self.src_bits(o, True)) for o in ops[1:4])
            r = f2u(ru_fma(a, b, c))
        elif base in ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_sqrt_f32"):
            trans = True
            x = flush(u2f(self.src_bits(ops[1], True))).astype(np.float64)
            with np.errstate(all="ignore"):
                if base == "v_exp_f32":
                    y = np.exp2(x)
                elif base == "v_log_f32":
                    y = np.log2(x)
                elif base == "v_rcp_f32":
                    y = 1.0 / x
                else:
                    y = np.sqrt(x)
            r = f2u(self.approx(y, x))
        elif base in ("v_sin_f32", "v_cos_f32"):
            # the argument in revolutions; beyond 256 of them the instructions return 0 / 1
            trans = True
            x = u2f(self.src_bits(ops[1], True)).astype(np.float64)
            with np.errstate(all="ignore"):
                y = np.sin(2 * np.pi * x) if base == "v_sin_f32" else np.cos(2 * np.pi * x)
                y = np.where(np.abs(x) > 256.0, 0.0 if base == "v_sin_f32" else 1.0, y)
                if self.perturb:
                    y = np.clip(y + self.rng.uniform(-TRIG_ABS_ERROR, TRIG_ABS_ERROR, size=self.n), -1.0, 1.0)
                y = np.where(np.isfinite(x), y, np.nan)
                r = f2u(y.astype(np.float32))
        elif base == "v_floor_f32":
            x = u2f(self.src_bits(ops[1], True))
            with np.errstate(all="ignore"):
                r = f2u(np.floor(x).astype(np.float32))
        else:
            raise ValueError("vector instruction " + name)
        self.v[d] = r
        if trans:
            self.trans_written[d] = self.executed
        else:
            self.trans_written.pop(d, None)
