mkdir -p gpurun_out
export TS_BENCH_WATCHDOG=100
for flags in "--no-cpu-baseline --no-face --no-modes --no-roofline" "--no-cpu-baseline --no-face --no-roofline" "--no-cpu-baseline --no-face --no-modes" "--no-cpu-baseline --no-modes --no-roofline" "--no-face --no-modes --no-roofline"; do
  echo "=== $flags" >> gpurun_out/bisect.log
  timeout 150 python bench.py --steps 20 --warmup 5 $flags >> gpurun_out/bisect.log 2>> gpurun_out/bisect.err
  echo "rc=$?" >> gpurun_out/bisect.log
done
tail -c 3000 gpurun_out/bisect.err
