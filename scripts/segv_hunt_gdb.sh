#!/bin/bash
# the fuzz sweep, twelve processes side by side, each under rocgdb: the C-level stack of all threads where one dies.   usage: segv_hunt_gdb.sh
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/r06_segv_gdb; mkdir -p $OUT; rm -f $OUT/*
echo "core_pattern: $(cat /proc/sys/kernel/core_pattern); stack limit: $(ulimit -s); host threads: $(nproc); free:"; free -g | head -2
for i in $(seq 0 11); do
  /opt/rocm/bin/rocgdb -q -batch -ex "handle SIGUSR1 nostop noprint" -ex run -ex "bt 12" -ex "info registers" -ex "x/6i \$rip-12" -ex "x/16wx \$rdi-32" -ex "x/4gx \$rsp+0xa8" -ex "info symbol \$rdi" -ex "thread apply all bt 4" \
      --args python scripts/fuzz_sweep.py $((1000 + i * 170)) 170 16 > $OUT/part_$i.log 2>&1 &
done
wait
echo "== $(grep -l '^seeds' $OUT/part_*.log | wc -l) of 12 finished"
for f in $(grep -l "SIGSEGV\|SIGABRT\|SIGBUS" $OUT/part_*.log); do echo "--- $f"; grep -n "SIGSEGV\|SIGABRT\|SIGBUS" -A70 $f | grep -v "^\S*-\[New Thread\|^\S*-\[Thread.*exited" | cut -c1-230 | head -110; done
