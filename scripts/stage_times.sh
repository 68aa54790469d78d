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
for r in rows[idx:]:
    n=r["Kernel_Name"].replace("void mprk::","").split("(")[0]
    if "rocclr" in n: continue
    print("%-34s grid=%8s %9.1f us" % (n, r.get("Grid_Size", r.get("Grid_Size_X","?")), (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
PY
