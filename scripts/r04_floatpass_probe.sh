#!/bin/bash
# Development probe (GPU box): the float pass on the root tape's host-generated code under its development switches, and the SQ
# counters of the frame.  usage: scripts/r04_floatpass_probe.sh <tag>
TAG=${1:-r04c}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
{
echo "== default"; python scripts/quick_bench.py bear:3:1024
for r in 0 3 8 12; do echo "== MPR_VOXEL_GEN_RUN=$r"; MPR_VOXEL_GEN_RUN=$r python scripts/quick_bench.py bear:3:1024; done
for w in 4 5 6; do echo "== MPR_VOXEL_GEN_WGS=$w"; MPR_VOXEL_GEN_WGS=$w python scripts/quick_bench.py bear:3:1024; done
echo "== MPR_VOXEL_GEN=0"; MPR_VOXEL_GEN=0 python scripts/quick_bench.py bear:3:1024
echo "== others"; python scripts/quick_bench.py bear:3:256 bear:3:512 bear:3:2048
} > $OUT/probe.txt 2>&1
bash scripts/pmc_sq.sh $TAG/sq bear 3 1024 > $OUT/sq.txt 2>&1
tail -50 $OUT/probe.txt
grep -A 30 "k_eval_voxels_gen" $OUT/sq.txt | head -40
