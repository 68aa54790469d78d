#!/bin/bash
# SQ instruction-mix counters of one model's frame (development aid).  usage: pmc_sq.sh <tag> <model> <dim> <size>
TAG=$1; MODEL=$2; DIM=$3; SIZE=$4
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/scripts/one_frame.py $MODEL $DIM $SIZE 3"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/a -o a -- $CMD > $OUT/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d $OUT/b -o b -- $CMD > $OUT/b.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_LEVEL_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM --output-format csv -d $OUT/c -o c -- $CMD > $OUT/c.log 2>&1
cd $ROOT
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for p in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].replace("void mprk::", "").split("(")[0]
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    if "rocclr" in k: continue
    print(k)
    for c, (v, n) in sorted(d.items()):
        print("   %-24s %16.0f /launch (%d)" % (c, v / n, n))
PY
