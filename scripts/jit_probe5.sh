#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
MPR_QB_STAGES=1 timeout 120 python scripts/quick_bench.py prospero:2:256 prospero:2:512 prospero:2:1024 architecture:3:256 architecture:3:512 architecture:3:1024 involute_gear_2d:2:512 involute_gear_2d:2:1024 involute_gear_3d:3:256 hello_world:2:256 bear:3:256 2>&1 | grep -v "amdgpu.ids\|launches"
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
