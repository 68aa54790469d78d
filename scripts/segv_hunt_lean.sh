#!/bin/bash
# twelve scripts/segv_hunt_lean.py side by side; a core file, if one appears, through rocgdb.   usage: segv_hunt_lean.sh ROUNDS [ENV=VALUE ...]
cd "$(dirname "$0")/.." || exit 1
ROUNDS=${1:-2}; shift
OUT=gpurun_out/r06_segv_lean; mkdir -p $OUT; rm -f $OUT/*.log core core.*
ulimit -c unlimited
for i in $(seq 0 11); do
  env "$@" python scripts/segv_hunt_lean.py $((1000 + i * 170)) 170 $ROUNDS > $OUT/part_$i.log 2>&1 &
done
wait
echo "== $(grep -l 'no crash' $OUT/part_*.log | wc -l) of 12 processes finished without a crash ($*)"
grep -l "Fatal Python error\|Aborted\|corrupt\|malloc" $OUT/part_*.log | while read f; do echo "--- $f"; grep -n -B2 -A14 "Fatal Python error\|corrupt\|malloc()" $f | cut -c1-220 | head -40; done
for c in core core.*; do
  [ -f "$c" ] || continue
  echo "--- $c"; timeout 120 /opt/rocm/bin/rocgdb -batch -ex "bt 40" -ex "info threads" $(which python) $c 2>&1 | grep -v "^\[New LWP\|^warning" | head -80
  break
done
