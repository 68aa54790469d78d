#!/bin/bash
# one of the twelve processes of round 5's fuzz sweep (seeds 2700..2869) died of a segmentation fault.  The same
# sweep again, twelve processes side by side, python's faulthandler on (the Python-level stack of a fatal signal names the call that
# died: the oracle through ctypes, or the library), core dumps allowed.  (Two more legs were run in round 6 and taken out again, profiles/
# r06_segv_hunt.txt section 3: the oracle built with the address sanitizer — clean —, and the library's host side built with it, which the
# GPU boxes' runtime does not load.)
# leg "guard": scripts/heapguard.c preloaded — freed blocks are parked and checked for writes (who freed the block that was written to).
# usage: segv_hunt.sh ROUNDS [plain|guard]      -> gpurun_out/r06_segv/
cd "$(dirname "$0")/.." || exit 1
ROUNDS=${1:-3}; LEG=${2:-plain}
OUT=gpurun_out/r06_segv; mkdir -p $OUT
ulimit -c unlimited
export PYTHONFAULTHANDLER=1
if [ "$LEG" = guard ]; then
  gcc -O2 -fPIC -shared -o /tmp/heapguard.so scripts/heapguard.c -ldl || exit 1
  export LD_PRELOAD=/tmp/heapguard.so
  export HEAPGUARD_LOG=$PWD/$OUT/heapguard
fi
for r in $(seq 1 $ROUNDS); do
  for i in $(seq 0 11); do
    python scripts/fuzz_sweep.py $((1000 + i * 170)) 170 16 > $OUT/${LEG}_r${r}_part_$i.log 2>&1 &
  done
  wait
  echo "== $LEG round $r: $(grep -l '^seeds' $OUT/${LEG}_r${r}_part_*.log | wc -l) of 12 processes finished; $(grep -h '^seeds' $OUT/${LEG}_r${r}_part_*.log | awk '{s+=$6} END {print s}') frames differ"
  grep -l "Fatal Python error\|Segmentation\|AddressSanitizer\|Traceback" $OUT/${LEG}_r${r}_part_*.log | while read f; do echo "--- $f"; grep -n -A45 "Fatal Python error\|AddressSanitizer\|Traceback" $f | cut -c1-200 | head -120; done
  if [ "$LEG" = guard ]; then
    echo "-- heapguard: $(cat $OUT/heapguard.* 2>/dev/null | grep -c 'write after free') reports"
    cat $OUT/heapguard.* 2>/dev/null | grep -A1 'write after free' | head -60
  fi
  # a core file (core_pattern "core": the repository's root): the C-level stack of every thread
  for c in core core.*; do
    [ -f "$c" ] || continue
    echo "--- $c"
    timeout 300 /opt/rocm/bin/rocgdb -batch -ex "bt 60" -ex "thread apply all bt 12" $(which python) $c 2>&1 | grep -v "^\[New LWP\|^warning\|^Reading\|^Download" | head -220 | tee $OUT/${LEG}_r${r}_core_bt.txt
    rm -f core core.*
    break
  done
done
