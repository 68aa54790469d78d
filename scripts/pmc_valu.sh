#!/bin/bash
# How busy are the vector units?  SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_* / SQ_WAIT_* per kernel of one model's frame
# (development aid).  usage: pmc_valu.sh <model> <dim> <size>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/valu_$1
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
CMD="python $ROOT/scripts/one_frame.py $1 $2 $3 3"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/a -o a -- $CMD > $OUT/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_IFETCH_LEVEL SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU SQ_CYCLES --output-format csv -d $OUT/b -o b -- $CMD > $OUT/b.log 2>&1
cd $ROOT
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for p in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].replace("void mprk::", "").split("(")[0]
        a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    if "rocclr" in k or not any(s in k for s in ("eval_voxels", "eval_tiles", "eval_normals")): continue
    print(k)
    for c, (v, n) in sorted(d.items()):
        print("   %-26s %16.0f /launch (%d)" % (c, v / n, n))
PY
