#!/bin/bash
# Development probe (GPU box): tiles per atomic of the float pass on the root tape's code.  usage: scripts/r04_probe3.sh <tag>
TAG=${1:-r04g}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
{
for r in 4 1 2 8 16; do echo "== MPR_VOXEL_GEN_TILES=$r"; MPR_VOXEL_GEN_TILES=$r python scripts/quick_bench.py bear:3:1024 bear:3:256; done
for w in 20 24; do echo "== MPR_VOXEL_GEN_WGS=$w"; MPR_VOXEL_GEN_WGS=$w python scripts/quick_bench.py bear:3:1024; done
} > $OUT/probe.txt 2>&1
grep -v amdgpu.ids $OUT/probe.txt
