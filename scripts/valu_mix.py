"""Instruction classes of the float pass's generated code (k_eval_voxels_jit_groups) for one model: how many of its VALU
instructions issue at the full rate, at half rate (v_cmp*, v_cndmask, v_cvt*, v_ldexp, v_div_scale / fmas / fixup, f64, anything
reading an SGPR pair as a mask) and at quarter rate (v_exp / v_log / v_rcp / v_rsq / v_sqrt) — profiles/r02a_issue_rates.txt and
profiles/r03a_issue_rates3.txt hold the measured rates.  The counts come from the sources themselves: the routines' assembly text
(mpr_amd/csrc/asm_float_bodies.hpp, fast paths: the branch every lane of a benchmark frame takes) and the translator's rows
(kernels_voxel_jit.hip: jt::row_of), weighted with the opcode histogram of the model's tape.  bench.py scales the mix to the
measured SQ_INSTS_VALU total for its rate-weighted roofline fraction.

    python scripts/valu_mix.py            ->  profiles/valu_mix.json
"""
import collections
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import mpr_amd as m

QUARTER = re.compile(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_f32")
HALF = re.compile(r"v_(cmp|cmpx|cndmask|cvt|ldexp|div_scale|div_fmas|div_fixup|readlane|perm|subbrev|addc|subb)|_f64|v_pk_")


def classify(text):
    """(full, half, quarter) VALU instructions of an assembly text; scalar instructions and labels are not counted"""
    full = half = quarter = 0
    for ins in re.split(r"\\n|\n", text):
        ins = ins.strip().strip('"').strip()
        if not ins.startswith("v_"):
            continue
        if QUARTER.match(ins):
            quarter += 1
        elif HALF.search(ins.split()[0]):
            half += 1
        else:
            full += 1
    return np.array([full, half, quarter], dtype=np.float64)


def routine_bodies():
    src = open(os.path.join(ROOT, "mpr_amd", "csrc", "asm_float_bodies.hpp")).read()
    out = {}
    for name in ("DIV", "SQRT", "EXP", "LOG", "SINCOS"):
        mm = re.search(r"#define MPR_ASM_%s_BODY(.*?)(?=\n\n|\n/\*|\n#define|\Z)" % name, src, re.S)
        body = mm.group(1)
        # (square root, exp, log: the BODY macro is the fast path; the rare cases sit in the TAIL macro)
        out[name] = classify(body)
    return out


def host_generated(R):
    """The float walk the HOST generates (csrc/voxel_gen.cpp) for the tapes it takes: the instructions a walk of the whole root tape
    executes (nothing decided: no run jumped over), disassembled with llvm-mc; the routines it still calls (sin / cos, general
    division) from their assembly text."""
    import ctypes, struct, subprocess
    mc = "/opt/rocm/lib/llvm/bin/llvm-mc"
    out = {}
    if not os.path.exists(mc):
        return out
    for model in ("bear",):
        d = np.ascontiguousarray(m.Tape(m.model(model)).data, dtype=np.uint64)
        buf = (ctypes.c_uint32 * 131072)()
        info = (ctypes.c_int32 * 3)()
        n = m.lib().mpr_test_voxel_gen(d.ctypes.data, len(d), 0, buf, 131072, info)      # no guards: the straight walk
        if n < 0:
            continue
        text = ",".join("0x%02x" % b for w in buf[:n] for b in struct.pack("<I", w))
        dis = subprocess.run([mc, "-arch=amdgcn", "-mcpu=gfx950", "-disassemble"], input=text.encode(), capture_output=True, check=True).stdout.decode()
        lines = [" ".join(l.split()) for l in dis.splitlines() if l.strip() and not l.strip().startswith(".")]
        main_code = lines[:next(k for k, l in enumerate(lines) if l.startswith("s_setpc_b64 s[72:73]"))]       # the stubs behind it never run here
        total = classify("\n".join(main_code))
        calls = collections.Counter(l.split(",")[-1].strip() for l in main_code if l.startswith("s_swappc_b64"))
        total = total + calls.get("s[52:53]", 0) * R["DIV"] + calls.get("s[60:61]", 0) * (R["SINCOS"] + np.array([2.0, 0, 0])) + \
            calls.get("s[62:63]", 0) * np.array([1.0, 1, 0])                      # (bear: every cosine is of the argument just seen)
        frac = total / total.sum()
        out[model] = {"full": round(float(frac[0]), 4), "half": round(float(frac[1]), 4), "quarter": round(float(frac[2]), 4),
                      "valu_per_walk_of_the_root_tape": float(total.sum()), "issue_units_per_walk": float(total @ np.array([1.0, 2.0, 4.0])),
                      "guarded_runs": int(info[1]), "note": "a tile's walk runs 451 of the 544 clauses on average (scripts/skip_study.py)"}
    return out


def main():
    R = routine_bodies()
    one = np.array([1.0, 0, 0])
    mov2 = np.array([2.0, 0, 0])                       # v_mov in, v_mov out around a call
    # group form rows (kernels_voxel_jit.hip: jt::row_of / jt::minmax); division by a constant: powers of two become a
    # multiplication, other constants the inline sequence (v_mul, v_cmp_class, v_mul, 4 x v_fmamk)
    minmax = np.array([1.0, 2, 0])                     # v_min / v_max + two v_cndmask
    divc = np.array([6.0, 1, 0])
    names = {}
    for line in open(os.path.join(ROOT, "include", "mpr_clause.h")):
        mm = re.match(r"\s*(MPR_OP_\w+)\s*=\s*(\d+)", line)
        if mm:
            names[int(mm.group(2))] = mm.group(1)[7:]
    result = {"_note": "fractions of the VALU instructions of the group form's generated code per class, fast paths of the routines, "
                       "root tape's opcode histogram (the per-tile tapes are the root tape minus a few per cent)",
              "_routines_full_half_quarter": {k: v.tolist() for k, v in R.items()}}
    per_kernel = {}
    for model in ("bear", "architecture", "involute_gear_3d", "hello_world", "prospero", "involute_gear_2d"):
        tape = m.Tape(m.model(model))
        d = np.asarray(tape.data)
        imm = (d >> np.uint64(32)).astype(np.uint32).view(np.float32)
        total = np.zeros(3)
        by = collections.OrderedDict()
        last_sin = None
        for k in range(1, len(d) - 1):
            op = names.get(int(d[k]) & 0xFF, "?")
            if op in ("SQRT_LHS",):
                c = R["SQRT"] + mov2
            elif op == "EXP_LHS":
                c = R["EXP"] + mov2
            elif op == "LOG_LHS":
                c = R["LOG"] + mov2
            elif op in ("SIN_LHS", "COS_LHS"):
                c = R["SINCOS"] + mov2 + (np.array([2.0, 0, 0]) if op == "SIN_LHS" else np.array([3.0, 1, 0]))
                if op == "SIN_LHS":
                    last_sin = ((int(d[k]) >> 16) & 0xFF, k)            # the slot whose cosine the routine keeps
                elif last_sin is not None and last_sin[0] == ((int(d[k]) >> 16) & 0xFF) and \
                        all(((int(d[j]) >> 8) & 0xFF) != last_sin[0] for j in range(last_sin[1], k)):
                    c = mov2 + np.array([1.0, 1, 0])                   # the cosine of the argument just seen: a compare and a move
            elif op in ("ASIN_LHS", "ACOS_LHS", "ATAN_LHS"):
                c = np.array([40.0, 8, 1])            # compiled leaves: an estimate, none of the benchmark's 3-D models but the gears use them
            elif op.startswith("MIN") or op.startswith("MAX"):
                c = minmax + (np.array([1.0, 0, 0]) if op.endswith("IMM") else 0)
            elif op == "DIV_LHS_IMM":
                v = abs(float(imm[k]))
                mant, _ = np.frexp(v)
                c = one if mant == 0.5 else divc
            elif op.startswith("DIV"):
                c = R["DIV"] + np.array([3.0, 0, 0])
            else:
                c = one
            total += c
            by[op] = by.get(op, 0) + float(c @ np.array([1.0, 2.0, 4.0]))
        frac = total / total.sum()
        units = float(total @ np.array([1.0, 2.0, 4.0]))
        per_kernel[model] = {"full": round(float(frac[0]), 4), "half": round(float(frac[1]), 4), "quarter": round(float(frac[2]), 4),
                             "valu_per_walk_of_the_root_tape": float(total.sum()), "issue_units_per_walk": units,
                             "issue_units_by_opcode": {k: round(v / units, 4) for k, v in sorted(by.items(), key=lambda kv: -kv[1])}}
    result["k_eval_voxels_jit_groups"] = per_kernel
    result["k_eval_voxels_gen"] = host_generated(R)
    with open(os.path.join(ROOT, "profiles", "valu_mix.json"), "w") as f:
        json.dump(result, f, indent=1)
        f.write("\n")
    print(json.dumps(result, indent=1))


if __name__ == "__main__":
    main()
