#!/bin/bash
# Run on the GPU box (gpurun): bench line, rocprofv3 kernel stats of the same command, and the two
# HBM counter passes (FETCH_SIZE and WRITE_SIZE in separate runs, --kernel-trace only) and one pass of the
# SQ instruction counters of the same command.
# usage: scripts/profile_round.sh <tag>       outputs under gpurun_out/<tag>/
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
BENCH="python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu --no-also"
cd /tmp
# (the default line's 20 + 100 frames: a context's first dozen frames run at lower clocks, and 25 frames per leg would be mostly those)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o st -- python $ROOT/bench.py --no-cpu --no-also > $OUT/stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- $BENCH > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_WAVES --output-format csv -d $OUT/pmc_sq -o q -- $BENCH > $OUT/pmc_sq.log 2>&1
cd $ROOT
python scripts/profile_summary.py $TAG > /dev/null      # pmc_summary.json -> bench's roofline.traffic
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
find $OUT -name '*.csv' | head -20
python scripts/profile_summary.py $TAG
