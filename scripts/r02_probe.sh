#!/bin/bash
# Round-2 first GPU call: issue-rate ceilings, SQ counters of the bear 1024^3 frame, device-side code generation probe.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02a
mkdir -p $OUT
cd $ROOT
timeout 300 scripts/ubench/issue_rates2 > $OUT/issue_rates.txt 2>&1
echo "issue_rates rc $?"
bash scripts/pmc_sq.sh r02a/sq_bear bear 3 1024 > $OUT/sq_bear.txt 2>&1
echo "sq rc $?"
timeout 200 scripts/ubench/jit_probe > $OUT/jit_probe.txt 2>&1
echo "jit_probe rc $?"
tail -40 $OUT/jit_probe.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc $?"
cat $OUT/bench.json
