#!/bin/bash
# instruction-cache counters per kernel of one model's frame (development aid).  usage: pmc_icache.sh <model> <dim> <size>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/icache_$1
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVES --output-format csv -d $OUT -o i -- python $ROOT/scripts/one_frame.py $1 $2 $3 3 > $OUT/log 2>&1
cd $ROOT
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for p in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].replace("void mprk::", "").split("(")[0]
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    if "rocclr" in k: continue
    g = lambda n: d[n][0] / max(d[n][1], 1) if n in d else 0
    req, miss = g("SQC_ICACHE_REQ"), g("SQC_ICACHE_MISSES")
    print("%-28s req %12.0f  hits %12.0f  misses %10.0f (%.2f%%)  dup %10.0f  waves %9.0f" % (k[:28], req, g("SQC_ICACHE_HITS"), miss, 100 * miss / max(req, 1), g("SQC_ICACHE_MISSES_DUPLICATE"), g("SQ_WAVES")))
PY
