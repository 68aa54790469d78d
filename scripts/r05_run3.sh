#!/bin/bash
# round 5: instruction counters per launch of the tile stages (headline frame), + the walk-cycles probe
cd "$(dirname "$0")/.." || exit 1
ROOT=$(pwd)
mkdir -p gpurun_out/r05b
export TMPDIR=/tmp
python scripts/walk_cycles.py bear 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05b/walk_cycles.txt
cat > /tmp/frames.py <<'PY'
import os, sys
sys.path.insert(0, os.environ['MPR_ROOT'])
import numpy as np, mpr_amd as m
T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
S = int(sys.argv[2]); tape = m.Tape(m.model(sys.argv[1])); ctx = m.Context(S)
for _ in range(12): ctx.render3D(tape, T)
print(ctx.tile_stage_forms())
PY
for sched in 1 0; do
  tag="pmc_s$sched"
  (cd /tmp && MPR_ROOT=$ROOT MPR_TILE_GEN_SCHED=$sched rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_LDS --output-format csv -d /tmp/prof_$tag -o x -- python /tmp/frames.py bear 1024 > /tmp/prof_$tag.log 2>&1)
  f=$(find /tmp/prof_$tag -name "*counter_collection.csv" | head -1)
  echo "== sched=$sched $(grep -v rocprof /tmp/prof_$tag.log | tail -1)"
  python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# per dispatch: counters; order by Dispatch_Id; tile kernels alternate stage 1, stage 2 per frame
disp = collections.OrderedDict()
for r in rows:
    d = int(r["Dispatch_Id"])
    disp.setdefault(d, {"name": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
tile = [v for k, v in sorted(disp.items()) if "k_eval_tiles" in v["name"]]
tile = tile[-16:]
for which in (0, 1):
    sel = tile[which::2]
    keys = [k for k in sel[0] if k != "name"]
    print("  tile stage launch #%d of a frame:" % (which + 1), {k: round(sum(s[k] for s in sel) / len(sel)) for k in keys})
for nm in ("k_skip0_parents", "k_eval_voxels_gen", "k_eval_normals_gen"):
    sel = [v for k, v in sorted(disp.items()) if nm in v["name"]][-8:]
    if sel:
        keys = [k for k in sel[0] if k != "name"]
        print("  %s:" % nm, {k: round(sum(s[k] for s in sel) / len(sel)) for k in keys})
PY
done 2>&1 | tee gpurun_out/r05b/pmc_tiles.txt
