#!/bin/bash
# scripts/fuzz_sweep.py over many seeds, a dozen processes side by side (the oracle on the host's cores is most of the time)
# usage: fuzz_sweep_parallel.sh FIRST PER_PROCESS PROCESSES TAG
cd "$(dirname "$0")/.." || exit 1
FIRST=${1:-1000}; PER=${2:-170}; N=${3:-12}; TAG=${4:-r05_fuzz}
mkdir -p gpurun_out/$TAG
for i in $(seq 0 $((N - 1))); do
  python scripts/fuzz_sweep.py $((FIRST + i * PER)) $PER 16 > gpurun_out/$TAG/part_$i.log 2>&1 &
done
wait
cat gpurun_out/$TAG/part_*.log | grep -v "amdgpu.ids" > gpurun_out/$TAG/sweep.log
grep -c "^seeds" gpurun_out/$TAG/sweep.log
grep "DIFFERS\|^seeds\|Error\|Traceback" gpurun_out/$TAG/sweep.log | head -60
