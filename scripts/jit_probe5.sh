#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for s in 16 1; do echo "== slots $s"; MPR_JIT_SLOTS=$s timeout 60 python scripts/quick_bench.py bear:3:1024 bear:3:512 bear:3:256 2>&1 | grep -v amdgpu.ids; done
