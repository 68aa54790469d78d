"""Condense gpurun_out/<tag>/ (scripts/profile_round.sh) into the files kept under profiles/.

  profiles/<tag>_bench.json          the bench.py line
  profiles/<tag>_kernel_stats.csv    per-kernel calls / total / average (rocprofv3 --kernel-trace --stats)
  profiles/pmc_summary.json          HBM bytes and SQ instruction counts per launch of the frame's kernels from the PMC passes

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  On gfx950 FETCH_SIZE tallies 128-byte
requests at 64 bytes for wide coalesced reads (MI355X_MICROARCH.md, "HBM"), so both the raw figure
and the doubled one are stored; `hbm_bytes_per_launch` uses the doubled read figure + raw writes.
"""
import csv, glob, json, os, sys, collections

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

b = os.path.join(src, "bench.json")
if os.path.exists(b):
    line = [l for l in open(b).read().splitlines() if l.startswith("{")]
    if line:
        with open(os.path.join(dst, tag + "_bench.json"), "w") as f:
            f.write(json.dumps(json.loads(line[-1]), indent=1) + "\n")


def short(name):
    return name.replace("void mprk::", "").split("(")[0]


def find(sub, pat):
    g = glob.glob(os.path.join(src, sub, "**", pat), recursive=True)
    return g[0] if g else None


ks = find("stats", "*kernel_stats.csv")
if ks:
    rows = list(csv.DictReader(open(ks)))
    with open(os.path.join(dst, tag + "_kernel_stats.csv"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu --no-also      (the default 20 warm-up + 100 timed frames, + 35 full_frames, + 12 first_frames)\n")
        f.write("name,calls,total_ns,average_ns,percentage\n")
        for r in rows:
            f.write('"%s",%s,%s,%s,%s\n' % (r.get("Name"), r.get("Calls"), r.get("TotalDurationNs"),
                                           r.get("AverageNs"), r.get("Percentage")))


def per_kernel(sub, counter):
    p = find(sub, "*counter_collection.csv")
    if not p:
        return {}
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(p)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        acc[k][0] += float(r["Counter_Value"])
        acc[k][1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items() if v[1]}


fetch = per_kernel("pmc_fetch", "FETCH_SIZE")
write = per_kernel("pmc_write", "WRITE_SIZE")
def first(prefixes, table):
    """the kernel of this run among the forms a pass can take (prefix match on the short name)"""
    for pre in prefixes:
        for k in table:
            if k.startswith(pre):
                return k
    return None


SQ = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_BUSY_CU_CYCLES", "SQ_WAVES")
sq = {c: per_kernel("pmc_sq", c) for c in SQ}
seen = set(fetch) | set(write) | set(sq[SQ[0]])
names = {"eval_voxels_f": first(("k_eval_voxels_gen_fp<3", "k_eval_voxels_gen<3", "k_eval_voxels_jit_groups<3", "k_eval_voxels_jit<3", "k_eval_voxels_asm<3", "k_eval_voxels<3"), seen),
         "eval_tiles_i": first(("k_eval_tiles<3, true",), seen), "eval_tiles_wide": first(("k_eval_tiles_wide<3>",), seen),
         "eval_pixels_d": first(("mprk::k_eval_normals_gen", "k_eval_normals_gen", "mprk::k_eval_normals_asm", "k_eval_normals_asm"), seen)}
out = {"_note": "KiB counters x1024; read bytes doubled per the gfx950 FETCH_SIZE correction; "
                "per launch = mean over all launches of that kernel in the run (tile stages: mean over the 3 stages)",
       "_source": "scripts/profile_round.sh %s: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE | SQ_INSTS_* of bench.py --steps 20 --warmup 5" % tag}
for key, kn in names.items():
    if kn is None:
        continue
    f = fetch.get(kn)
    w = write.get(kn)
    if not f and not w:
        continue
    fb = f[0] * 1024 if f else None
    wb = w[0] * 1024 if w else None
    out[key] = {"kernel": kn, "launches": (f or w)[1],
                "fetch_bytes_raw": fb, "fetch_bytes_corrected": 2 * fb if fb is not None else None,
                "write_bytes": wb,
                "hbm_bytes_per_launch": (2 * fb if fb is not None else 0) + (wb or 0)}
    s = {c: sq[c][kn][0] for c in SQ if kn in sq[c]}
    if s:
        out[key]["sq"] = s          # per launch, summed over the chip
if len(out) > 2:
    with open(os.path.join(dst, "pmc_summary.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
print(json.dumps(out, indent=1))
