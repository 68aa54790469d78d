#!/bin/bash
# The reference's table benchmarks (benchmark/render_table.cpp against include/mpr.hpp) for every model: run on the GPU box.
# usage: scripts/cpp_tables.sh <tag>     -> gpurun_out/<tag>_cpp_tables.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02}
OUT=$ROOT/gpurun_out/${TAG}_cpp_tables.txt
cd $ROOT
g++ -O2 -std=c++17 -Iinclude benchmark/render_table.cpp -o /tmp/render_table -Lmpr_amd -lmpr_amd -Wl,-rpath,$ROOT/mpr_amd || exit 1
cd /tmp
{
echo "# benchmark/render_table (C++ against include/mpr.hpp), one MI355X,"
echo "# protocol of the reference's benchmark/stats.cpp:19-47 (20 warm-up + 100 timed blocking calls):"
echo "# size  mean_ms  stdev_ms"
for m in prospero involute_gear_2d hello_world; do echo "== render2D $m.frep"; timeout 300 /tmp/render_table 2 $ROOT/fixtures/models/$m.frep 2>&1 | grep -v amdgpu.ids; done
for m in bear architecture involute_gear_3d hello_world; do echo "== render3D $m.frep (heightmap + normals)"; timeout 300 /tmp/render_table 3 $ROOT/fixtures/models/$m.frep 2>&1 | grep -v amdgpu.ids; done
} > $OUT
cat $OUT
