#!/bin/bash
# Development probe (GPU box): SQ counters of the tile stages with exact / loose enclosures.  usage: scripts/r04_probe4.sh <tag>
TAG=${1:-r04m}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for l in 0 1; do
  MPR_TILE_GEN_LOOSE=$l timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CU_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM SQ_INSTS_LDS --output-format csv -d $OUT/l$l -o a -- python $ROOT/scripts/one_frame.py bear 3 1024 4 > $OUT/l$l.log 2>&1
done
cd $ROOT
python - <<PY
import csv, glob, collections
for l in (0, 1):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in glob.glob("$OUT/l%d/*counter_collection.csv" % l):
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"].replace("void mprk::", "").split("(")[0]
            if "eval_tiles" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("MPR_TILE_GEN_LOOSE=%d" % l)
    for k, d in acc.items():
        n = len(next(iter(d.values())))
        # launches alternate 16^3 stage / 4^3 stage: print per position in the frame
        per = 2
        for j in range(per):
            print("  %s launch %d of a frame: " % (k, j) + "  ".join("%s=%.3g" % (c, sum(v[j::per]) / max(len(v[j::per]), 1)) for c, v in sorted(d.items())))
PY
