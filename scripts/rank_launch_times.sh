#!/bin/bash
# Per-launch durations of ONE rank's frames in a WORLD-rank deal, played on one GPU without the collective (what sets the floor of the
# simulated scaling, scripts/sim_scaling.py).  usage: rank_launch_times.sh MODEL SIZE WORLD RANK
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cat > /tmp/rank_frames.py <<'PY'
import os, sys
sys.path.insert(0, os.environ['MPR_ROOT'])
import numpy as np, torch, mpr_amd as m
from mpr_amd.multigpu import TileParallelRenderer
name, S, world, rank = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
tape = m.Tape(m.model(name)); ctx = m.Context(S)
def mk(n):
    t = torch.zeros(n, dtype=torch.int32, device="cuda"); torch.cuda.synchronize(); return t, t.data_ptr()
tpr = TileParallelRenderer(ctx, m, rank, world, mk, lambda o, i: None, dim=3)
tpr.plan(tape, T)
for _ in range(30): tpr.render(tape, T)
print(ctx.tile_stage_forms(), "|", ctx.float_kernel())
PY
D=/tmp/rank_times_$$
(cd /tmp && MPR_ROOT=$ROOT rocprofv3 --kernel-trace --output-format csv -d $D -o x -- python /tmp/rank_frames.py "$@" > $D.log 2>&1)
echo "== rank $4 of $3, $1 $2: $(grep -v rocprof $D.log | tail -1)"
python - "$(find $D -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
frames, cur = [], []
for r in rows:
    n = r["Kernel_Name"]
    if "k_preload_tiles" in n or "k_zero_owned" in n:
        if cur and ("k_preload_tiles" in n and any("k_eval" in x[0] for x in cur)): frames.append(cur); cur = []
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
    acc.setdefault("sum of launches", []).append(sum(x[1] for x in fr))
for k, v in acc.items():
    print("   %-64s %8.1f us  (n=%d)" % (k[:64], sum(v) / len(v), len(v)))
PY
