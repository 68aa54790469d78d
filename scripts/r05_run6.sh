#!/bin/bash
# round 5: where do the lean last stage's wavefronts spend their cycles (SQ counters per dispatch, bear 1024^3)
cd "$(dirname "$0")/.." || exit 1
ROOT=$(pwd)
mkdir -p gpurun_out/r05c
export TMPDIR=/tmp
cat > /tmp/frames.py <<'PY'
import os, sys
sys.path.insert(0, os.environ['MPR_ROOT'])
import numpy as np, mpr_amd as m
T = np.eye(4, dtype=np.float32); T[3, 2] = 0.3
S = int(sys.argv[2]); tape = m.Tape(m.model(sys.argv[1])); ctx = m.Context(S)
for _ in range(8): ctx.render3D(tape, T)
print(ctx.tile_stage_forms())
PY
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64" "SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_FLAT"; do
  i=$((i+1))
  (cd /tmp && MPR_ROOT=$ROOT timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pm$i -o p -- python /tmp/frames.py bear 1024 > /tmp/pm$i.log 2>&1)
done
python - <<'PY' | tee gpurun_out/r05c/wave_cycles_lean.txt
import csv, glob, collections
disp = {}
for f in glob.glob("/tmp/pm*/**/*counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    per = collections.OrderedDict()
    for r in rows:
        per.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
    # tile-stage dispatches of the last 4 frames, by their order within a frame
    seq = [v for k, v in sorted(per.items())]
    frames, cur = [], []
    for v in seq:
        if "k_preload_tiles" in v["name"]:
            if cur: frames.append(cur)
            cur = []
        cur.append(v)
    frames.append(cur)
    for fr in frames[-5:-1]:
        n = collections.Counter()
        for v in fr:
            if "k_eval_tiles" not in v["name"]: continue
            key = ("lean" if "true, true>" in v["name"] else "full") + "#%d" % (n[v["name"]] + 1)
            n[v["name"]] += 1
            for c, x in v.items():
                if c != "name": disp.setdefault(key, collections.defaultdict(list))[c].append(x)
for key in sorted(disp):
    print(key)
    for c in sorted(disp[key]):
        v = disp[key][c]
        print("   %-28s %14.0f" % (c, sum(v) / len(v)))
PY
