#!/bin/bash
# Kernel trace of pipelined frames (MPR_PIPELINE=1): start / end of every launch of the last frame, so that the overlap of
# the last tile stage (stream 1) and the float pass (stream 2) can be read off.  usage: pipe_trace.sh <model> <size> <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$3
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for mode in 1 0; do
MPR_PIPELINE=$mode timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/p$mode -o t -- python $ROOT/scripts/one_frame.py $1 3 $2 6 > $OUT/log$mode 2>&1
python - <<PY
import csv, glob
p = glob.glob("$OUT/p$mode/**/t_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last frame that was rendered the fast way (one_frame.py ends with a reader's reference frame): the last-but-one preload
idx = [i for i, r in enumerate(rows) if "preload" in r["Kernel_Name"]][-2]
end = [i for i, r in enumerate(rows) if "preload" in r["Kernel_Name"]][-1]
t0 = int(rows[idx]["Start_Timestamp"])
print("MPR_PIPELINE=$mode  $1 $2^3  (microseconds from the frame's first launch)")
for r in rows[idx:end]:
    n = r["Kernel_Name"].replace("void mprk::", "").split("(")[0]
    b, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("  %-44s queue %-3s start %8.1f  end %8.1f  (%7.1f us)" % (n[:44], r.get("Queue_Id", "?"), (b - t0) / 1e3, (e - t0) / 1e3, (e - b) / 1e3))
print("  frame: %.1f us" % ((max(int(r["End_Timestamp"]) for r in rows[idx:end]) - t0) / 1e3))
PY
done
