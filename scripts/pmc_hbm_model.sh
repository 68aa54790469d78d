#!/bin/bash
# HBM traffic per kernel launch of one model's frame (development aid).  usage: pmc_hbm_model.sh <model> <dim> <size>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/hbm_$1
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -o f -- python $ROOT/scripts/one_frame.py $1 $2 $3 3 > $OUT/f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w -o w -- python $ROOT/scripts/one_frame.py $1 $2 $3 3 > $OUT/w.log 2>&1
cd $ROOT
python - <<PY
import csv, glob, collections
for sub, cn, mul in (("f", "FETCH_SIZE", 2048.0), ("w", "WRITE_SIZE", 1024.0)):
    rows = []
    for p in glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True):
        rows += [r for r in csv.DictReader(open(p)) if r["Counter_Name"] == cn]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    # last frame: after the last preload
    idx = [i for i, r in enumerate(rows) if "preload" in r["Kernel_Name"]][-1]
    print(cn, "(MB per launch, last frame; reads with the gfx950 x2 correction)")
    for r in rows[idx:]:
        k = r["Kernel_Name"].replace("void mprk::", "").split("(")[0]
        print("   %-30s %10.1f" % (k[:30], float(r["Counter_Value"]) * mul / 1e6))
PY
