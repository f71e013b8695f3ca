#!/bin/bash
# coalesce x streams sweep of the headline mode (GPU box): bash tools/gs_sweep.sh "8,2 8,3 4,4" > gpurun_out/gs.jsonl
for gs in $1; do
  G=${gs%,*}; S=${gs#*,}
  TS_BENCH_WATCHDOG=120 timeout 150 python bench.py --steps 48 --warmup 8 --coalesce $G --streams $S --no-cpu-baseline --no-face --no-modes --no-roofline 2>/dev/null | tail -1
done
