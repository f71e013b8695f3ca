#!/bin/bash
# quick GPU loop: the chain-related parity tests, then bench lines (no CPU baseline / face) under a few kernel knobs
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$1" 2>&1 | tail -15 > gpurun_out/q_tests.log
shift
i=0
for env in "$@"; do
  i=$((i+1))
  echo "=== $env" >> gpurun_out/q_bench.log
  env $env TS_BENCH_WATCHDOG=150 timeout 200 python bench.py --steps 24 --warmup 8 --no-cpu-baseline --no-face --no-modes 2>> gpurun_out/q_bench.err | tail -1 >> gpurun_out/q_bench.log
done
cat gpurun_out/q_tests.log
