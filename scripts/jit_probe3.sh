#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
python -m pytest tests/test_gpu_render.py -x -q -m gpu -k "generated_code" 2>&1 | tail -3
python -m pytest tests/test_gpu_primitives.py -x -q -m gpu -k "assembly_interpreter and (6 or 7 or 8)" 2>&1 | tail -2
echo "== default"; python scripts/quick_bench.py bear:3:1024 architecture:3:1024 prospero:2:1024 involute_gear_2d:2:2048 2>&1 | grep -v amdgpu.ids
echo "== translate only"; MPR_JIT_DEBUG=1 python scripts/quick_bench.py bear:3:1024 2>&1 | grep -v amdgpu.ids
echo "== translate once per wave"; MPR_JIT_DEBUG=2 python scripts/quick_bench.py bear:3:1024 2>&1 | grep -v amdgpu.ids
