#!/bin/bash
# round 5, first GPU run: smoke, the GPU parity suite, the bench with the scheduled interval walks on / off, the probe
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r05a
export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke > gpurun_out/r05a/smoke.log 2>&1; echo "smoke rc=$?"
tail -2 gpurun_out/r05a/smoke.log
timeout 200 python scripts/r05_probe.py bear:256 bear:1024 > gpurun_out/r05a/probe.jsonl 2> gpurun_out/r05a/probe.err; echo "probe rc=$?"
cat gpurun_out/r05a/probe.jsonl; tail -3 gpurun_out/r05a/probe.err
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r05a/gputests.log 2>&1; echo "gpu tests rc=$?"
tail -15 gpurun_out/r05a/gputests.log
timeout 200 python bench.py --steps 20 --warmup 5 > gpurun_out/r05a/bench.json 2> gpurun_out/r05a/bench.err; echo "bench rc=$?"
cat gpurun_out/r05a/bench.json | cut -c1-1500
MPR_TILE_GEN_SCHED=0 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/r05a/bench_old.json 2> gpurun_out/r05a/bench_old.err; echo "bench(old walks) rc=$?"
cat gpurun_out/r05a/bench_old.json | cut -c1-600
