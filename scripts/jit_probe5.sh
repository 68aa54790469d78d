#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for rep in 1 2; do for G in 0 1 2; do echo "== MPR_VOXEL_GROUPS=$G"; MPR_VOXEL_GROUPS=$G MPR_QB_STAGES=1 timeout 120 python scripts/quick_bench.py architecture:3:2048 architecture:3:1024 2>&1 | grep -v "amdgpu.ids\|launches"; done; done
MPR_DEBUG_CHOICES=1 timeout 60 python scripts/quick_bench.py architecture:3:2048 2>&1 | grep "last stage" | head -3
