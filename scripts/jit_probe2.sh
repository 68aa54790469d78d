#!/bin/bash
# float-pass timing breakdown of the generated-code path (development)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
echo "== default"; python scripts/quick_bench.py bear:3:1024 architecture:3:1024 prospero:2:1024 involute_gear_2d:2:2048
echo "== MPR_VOXEL_JIT=0"; MPR_VOXEL_JIT=0 python scripts/quick_bench.py bear:3:1024 architecture:3:1024 prospero:2:1024 involute_gear_2d:2:2048
echo "== translate only"; MPR_JIT_DEBUG=1 python scripts/quick_bench.py bear:3:1024
echo "== translate once per wave"; MPR_JIT_DEBUG=2 python scripts/quick_bench.py bear:3:1024
echo "== neither"; MPR_JIT_DEBUG=3 python scripts/quick_bench.py bear:3:1024
