#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 300 python -m pytest tests/test_gpu_primitives.py -x -q -m gpu -k "assembly_interpreter or division_by" 2>&1 | tail -5
timeout 300 python -m pytest tests/test_gpu_render.py -x -q -m gpu 2>&1 | tail -5
echo "== default (groups)"; MPR_JIT_DEBUG=16 timeout 60 python scripts/quick_bench.py bear:3:1024 architecture:3:1024 prospero:2:1024 involute_gear_2d:2:2048 2>&1 | grep -v amdgpu.ids
