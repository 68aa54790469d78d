"""Condense gpurun_out/<tag>/ (scripts/profile_round.sh) into the files kept under profiles/.

  profiles/<tag>_bench.json          the bench.py line
  profiles/<tag>_kernel_stats.csv    per-kernel calls / total / average (rocprofv3 --kernel-trace --stats)
  profiles/pmc_summary.json          HBM bytes per launch of the dominant kernel from the PMC passes

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
        f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu --no-also\n")
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
names = {"eval_voxels_f": "k_eval_voxels_asm<3>", "eval_tiles_i": "k_eval_tiles<3, true>", "eval_tiles_wide": "k_eval_tiles_wide<3>",
         "eval_pixels_d": "mprk::k_eval_normals_asm"}
out = {"_note": "KiB counters x1024; read bytes doubled per the gfx950 FETCH_SIZE correction; "
                "per launch = mean over all launches of that kernel in the run (tile stages: mean over the 3 stages)",
       "_source": "gpurun_out/%s/pmc_fetch, pmc_write (scripts/profile_round.sh)" % tag}
for key, kn in names.items():
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
if len(out) > 2:
    with open(os.path.join(dst, "pmc_summary.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
print(json.dumps(out, indent=1))
