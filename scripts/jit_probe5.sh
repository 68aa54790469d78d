#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for w in 4 2; do
echo "== group waves $w"; MPR_AMD_LIB=$ROOT/build_ab/libmpr_w$w.so MPR_JIT_DEBUG=16 timeout 60 python scripts/quick_bench.py bear:3:1024 architecture:3:1024 2>&1 | grep -v amdgpu.ids
done
