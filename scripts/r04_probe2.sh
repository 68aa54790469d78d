#!/bin/bash
# Development probe (GPU box): the lean last stage with / without its guarded forward walk, frames rendered the reference's way,
# and every launch of the other BASELINE configurations.  usage: scripts/r04_probe2.sh <tag>
TAG=${1:-r04e}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
{
echo "== default"; MPR_QB_STAGES=1 python scripts/quick_bench.py bear:3:1024
echo "== MPR_TILE_GEN_GUARDS=0"; MPR_TILE_GEN_GUARDS=0 MPR_QB_STAGES=1 python scripts/quick_bench.py bear:3:1024
echo "== MPR_LAST_STAGE_PUSH=1"; MPR_LAST_STAGE_PUSH=1 MPR_QB_STAGES=1 python scripts/quick_bench.py bear:3:1024
echo "== others"; MPR_QB_STAGES=1 python scripts/quick_bench.py architecture:3:2048 involute_gear_2d:2:4096 prospero:2:1024 prospero:2:512 bear:3:256 bear:3:512
} > $OUT/probe.txt 2>&1
grep -v amdgpu.ids $OUT/probe.txt
