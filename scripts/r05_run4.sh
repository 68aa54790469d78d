#!/bin/bash
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
for _ in range(12): ctx.render3D(tape, T)
print(ctx.tile_stage_forms())
PY
for dbg in 0 32; do
  for ms in "bear 1024" "bear 2048"; do
    set -- $ms
    tag="d${dbg}_$1$2"
    (cd /tmp && MPR_ROOT=$ROOT MPR_DEBUG_TILES=$dbg rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$tag -o x -- python /tmp/frames.py $1 $2 > /tmp/prof_$tag.log 2>&1)
    f=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
    echo "== $tag: $(grep -v rocprof /tmp/prof_$tag.log | tail -1)"
    python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
frames, cur = [], []
for r in rows:
    n = r["Kernel_Name"]
    if "k_preload_tiles" in n:
        if cur: frames.append(cur)
        cur = []
    cur.append((n, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0, int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Grid_Size", "")))
frames.append(cur)
acc = collections.OrderedDict()
for fr in frames[-8:-1]:
    cnt = collections.Counter()
    for n, us, s, e, g in fr:
        short = n.split("(")[0].replace("void ", "").replace("mprk::", "")
        if "k_eval_tiles" not in short and "skip0" not in short: continue
        cnt[short] += 1
        acc.setdefault("%s#%d grid %s" % (short, cnt[short], g), []).append(us)
for k, v in acc.items():
    print("   %-70s %8.1f us  (n=%d)" % (k[:70], sum(v) / len(v), len(v)))
PY
  done
done 2>&1 | tee gpurun_out/r05b/nowalk.txt
