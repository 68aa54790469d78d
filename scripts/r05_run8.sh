#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r05d
export TMPDIR=/tmp
timeout 300 python scripts/r05_probe.py bear:256 bear:512 bear:1024 bear:2048 2>/dev/null | grep '"sched": true' | tee gpurun_out/r05d/probe2.jsonl | cut -c1-330
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r05d/bench.json 2> gpurun_out/r05d/bench.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/r05d/bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05d/bench.json"))
for k in ("ms_per_step", "full_frames", "first_frames", "reader", "also"):
    print(k, json.dumps(d.get(k))[:600])
print("roofline", json.dumps(d["roofline"])[:300])
PY
