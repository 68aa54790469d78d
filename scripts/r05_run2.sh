#!/bin/bash
# round 5: per-launch kernel durations of the headline frame with the scheduled walks on / off / unscheduled (window 1)
cd "$(dirname "$0")/.." || exit 1
ROOT=$(pwd)
mkdir -p gpurun_out/r05b
export TMPDIR=/tmp
cat > /tmp/frames.py <<'PY'
import os, sys
sys.path.insert(0, os.environ['MPR_ROOT'])
import numpy as np, mpr_amd as m
T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
S = int(sys.argv[2]); tape = m.Tape(m.model(sys.argv[1])); ctx = m.Context(S)
for _ in range(30): ctx.render3D(tape, T)
print(ctx.tile_stage_forms())
PY
for cfg in "1 0" "0 0" "1 1"; do
  set -- $cfg
  for ms in "bear 1024" "bear 2048"; do
    set -- $cfg $ms
    tag="s$1w$2_$3$4"
    (cd /tmp && MPR_ROOT=$ROOT MPR_TILE_GEN_SCHED=$1 MPR_IGEN_WINDOW=$2 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$tag -o x -- python /tmp/frames.py $3 $4 > /tmp/prof_$tag.log 2>&1)
    f=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
    echo "== $tag: $(grep -v rocprof /tmp/prof_$tag.log | tail -1)"
    python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 10 frames: group launches of k_eval_tiles by order within a frame (a frame starts with k_preload_tiles)
frames, cur = [], []
for r in rows:
    n = r["Kernel_Name"]
    if "k_preload_tiles" in n:
        if cur: frames.append(cur)
        cur = []
    cur.append((n, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
frames.append(cur)
acc = collections.OrderedDict()
use = frames[-12:-2]
for fr in use:
    cnt = collections.Counter()
    for n, us, s, e in fr:
        short = n.split("(")[0].replace("void ", "").replace("mprk::", "")
        cnt[short] += 1
        key = "%s#%d" % (short, cnt[short])
        acc.setdefault(key, []).append(us)
    acc.setdefault("frame_span", []).append((fr[-1][3] - fr[0][2]) / 1000.0)
for k, v in acc.items():
    print("   %-60s %8.1f us  (n=%d)" % (k[:60], sum(v) / len(v), len(v)))
PY
  done
done 2>&1 | tee gpurun_out/r05b/launches.txt
