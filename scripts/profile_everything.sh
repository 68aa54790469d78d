#!/bin/bash
# Everything a round's profiles/ holds, in one gpurun call: scripts/profile_round.sh (bench line, rocprofv3 kernel stats, PMC passes),
# the reference's C++ tables, one record per model x size (bench.py --all, CPU legs included), the simulated 1..8-rank scaling,
# 2000 frames per configuration (outliers), the forward walk's cycles, per-launch durations.   usage: profile_everything.sh TAG
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
bash scripts/profile_round.sh $TAG > $OUT/profile_round.log 2>&1; tail -3 $OUT/profile_round.log
bash scripts/cpp_tables.sh > $OUT/cpp_tables.txt 2>&1; tail -2 $OUT/cpp_tables.txt
timeout ${RECORDS_TIMEOUT:-900} python bench.py --all --out $OUT/records.jsonl > /dev/null 2> $OUT/records.err; tail -2 $OUT/records.err
timeout 600 python scripts/sim_scaling.py bear:1024 bear:2048 architecture:2048 2>/dev/null | grep -v amdgpu > $OUT/sim_scaling.txt; grep "world 8" $OUT/sim_scaling.txt
timeout 300 python scripts/outlier_probe.py ${OUTLIER_FRAMES:-2000} 2>/dev/null > $OUT/outliers.jsonl; cut -c1-260 $OUT/outliers.jsonl
timeout 120 python scripts/walk_cycles.py bear 2>/dev/null | grep -v amdgpu > $OUT/walk_cycles.txt; cat $OUT/walk_cycles.txt
{ bash scripts/launch_times.sh bear 1024; bash scripts/launch_times.sh bear 1024 MPR_TILE_TIGHT=0; bash scripts/launch_times.sh bear 1024 MPR_VOXEL_FP=0; bash scripts/launch_times.sh bear 1024 MPR_LAST_STAGE_PUSH=1;
  bash scripts/launch_times.sh bear 2048; bash scripts/launch_times.sh architecture 2048; bash scripts/launch_times.sh architecture 1024; bash scripts/launch_times.sh prospero 1024; bash scripts/launch_times.sh involute_gear_2d 4096; } > $OUT/launch_times.txt 2>&1
grep "frame_span\|^==" $OUT/launch_times.txt | cut -c1-200
