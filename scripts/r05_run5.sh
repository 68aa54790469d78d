#!/bin/bash
# round 5: the lean (six wavefronts per SIMD) loose stages: parity of the suites that run them, per-launch times, frame times
cd "$(dirname "$0")/.." || exit 1
ROOT=$(pwd)
mkdir -p gpurun_out/r05c
export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 200 python scripts/r05_probe.py bear:256 bear:1024 bear:2048 2>/dev/null | tee gpurun_out/r05c/probe.jsonl | cut -c1-420
cat > /tmp/frames.py <<'PY'
import os, sys
sys.path.insert(0, os.environ['MPR_ROOT'])
import numpy as np, mpr_amd as m
T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
S = int(sys.argv[2]); tape = m.Tape(m.model(sys.argv[1])); ctx = m.Context(S)
for _ in range(30): ctx.render3D(tape, T)
print(ctx.tile_stage_forms())
PY
for lean in 1 0; do
  for ms in "bear 1024" "bear 2048"; do
    set -- $ms
    tag="l${lean}_$1$2"
    (cd /tmp && MPR_ROOT=$ROOT MPR_TILE_GEN_LEAN=$lean rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$tag -o x -- python /tmp/frames.py $1 $2 > /tmp/prof_$tag.log 2>&1)
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
    cur.append((n, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
frames.append(cur)
acc = collections.OrderedDict()
for fr in frames[-12:-2]:
    cnt = collections.Counter()
    for n, us, s, e in fr:
        short = n.split("(")[0].replace("void ", "").replace("mprk::", "")
        cnt[short] += 1
        acc.setdefault("%s#%d" % (short, cnt[short]), []).append(us)
    acc.setdefault("frame_span", []).append((fr[-1][3] - fr[0][2]) / 1000.0)
for k, v in acc.items():
    if "eval_tiles" in k or "frame_span" in k or "voxels" in k: print("   %-60s %8.1f us  (n=%d)" % (k[:60], sum(v) / len(v), len(v)))
PY
  done
done 2>&1 | tee gpurun_out/r05c/launches.txt
