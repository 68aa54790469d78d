#!/bin/bash
# Per-launch kernel durations of a model's frames, in launch order within a frame (rocprofv3 --kernel-trace; mean over ten frames).
# usage (on the GPU box, from the repository's root):  scripts/launch_times.sh MODEL SIZE [ENV=VALUE ...]
#   scripts/launch_times.sh bear 1024                          the default frames
#   scripts/launch_times.sh bear 1024 MPR_LAST_STAGE_PUSH=1    frames that leave the reference's tiles and tapes behind
#   scripts/launch_times.sh bear 1024 MPR_TILE_GEN_LEAN=0      loose stages in the 128-register kernel
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
MODEL=${1:-bear}; SIZE=${2:-1024}; shift 2
export TMPDIR=/tmp
cat > /tmp/launch_times_frames.py <<'PY'
import os, sys
sys.path.insert(0, os.environ['MPR_ROOT'])
import numpy as np, mpr_amd as m
T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
tape = m.Tape(m.model(sys.argv[1])); ctx = m.Context(int(sys.argv[2]))
dim = 2 if sys.argv[1] in ("prospero", "involute_gear_2d", "hello_world", "circle") else 3
for _ in range(30): (ctx.render3D(tape, T) if dim == 3 else ctx.render2D(tape))
print(ctx.tile_stage_forms(), "|", ctx.float_kernel(), "|", ctx.normals_kernel() if dim == 3 else "")
PY
D=/tmp/launch_times_$$
(cd /tmp && env MPR_ROOT=$ROOT "$@" rocprofv3 --kernel-trace --output-format csv -d $D -o x -- python /tmp/launch_times_frames.py $MODEL $SIZE > $D.log 2>&1)
echo "== $MODEL $SIZE $*: $(grep -v rocprof $D.log | tail -1)"
python - "$(find $D -name '*kernel_trace.csv' | head -1)" <<'PY'
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
    acc.setdefault("frame_span (first launch's start to last launch's end)", []).append((fr[-1][3] - fr[0][2]) / 1000.0)
for k, v in acc.items():
    print("   %-64s %8.1f us  (n=%d)" % (k[:64], sum(v) / len(v), len(v)))
PY
