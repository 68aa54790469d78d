#!/bin/bash
# per-launch kernel durations of one frame (development aid).  usage: stage_times.sh <model> <dim> <size>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/stage_$1
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $ROOT/scripts/one_frame.py $1 $2 $3 4 > $OUT/log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/t_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last frame only: find last k_preload_tiles
idx=[i for i,r in enumerate(rows) if "preload" in r["Kernel_Name"]][-1]
t0=int(rows[idx]["Start_Timestamp"]); prev=t0
for r in rows[idx:]:
    n=r["Kernel_Name"].replace("void mprk::","").split("(")[0]
    b,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print("%-34s grid=%8s %9.1f us   start +%8.1f us  gap %6.1f us" % (n[:34], r.get("Grid_Size", r.get("Grid_Size_X","?")), (e-b)/1e3, (b-t0)/1e3, (b-prev)/1e3))
    prev=e
PY
