#!/bin/bash
# round 5: the GPU suite + smoke + the bench line (what the driver runs at round end)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r05_check
export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_check/bench.json 2> gpurun_out/r05_check/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_check/bench.json"))
print({k: d[k] for k in ("ms_per_step", "value")}, "full", d["full_frames"]["ms_per_frame_mean"], "first", d["first_frames"]["ms_per_frame_mean"], "reader", d["reader"]["ms_read_tiles_after_a_frame_mean"])
print([(a["workload"], a["ms_per_frame_mean"]) for a in d["also"]])
print({k: d["roofline"][k] for k in ("kernel", "frac", "hbm_frac", "kernel_ms") if k in d["roofline"]})
print(d["cpu_baseline"])
PY
