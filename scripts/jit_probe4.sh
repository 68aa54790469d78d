#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
echo "== shared hot code"; MPR_JIT_GAP=256 MPR_JIT_DEBUG=80 timeout 60 python scripts/quick_bench.py bear:3:1024 2>&1 | grep -v amdgpu.ids | tail -3
