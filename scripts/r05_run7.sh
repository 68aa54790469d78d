#!/bin/bash
# round 5: after the atomics: probe (3-D sizes, 2-D configurations, architecture), bench, GPU suite
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r05d
export TMPDIR=/tmp
timeout 300 python scripts/r05_probe.py bear:256 bear:512 bear:1024 bear:2048 architecture:1024 architecture:2048 involute_gear_3d:1024 2>/dev/null | tee gpurun_out/r05d/probe.jsonl | cut -c1-400
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r05d/bench.json 2> gpurun_out/r05d/bench.err; echo "bench rc=$?"; cut -c1-900 gpurun_out/r05d/bench.json
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
