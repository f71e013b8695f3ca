#!/bin/bash
# bench lines (no CPU baseline / face / modes) under kernel knobs, one per argument ("A=1 B=2"); prints the key numbers
mkdir -p gpurun_out; : > gpurun_out/knobs.log
for env in "$@"; do
  line=$(env $env TS_BENCH_WATCHDOG=150 timeout 200 python bench.py --steps 24 --warmup 8 --no-cpu-baseline --no-face --no-modes 2>/dev/null | tail -1)
  python - "$env" "$line" >> gpurun_out/knobs.log <<'PY'
import json, sys
try:
    d = json.loads(sys.argv[2])
    print(f'{sys.argv[1]:40s} value {d["value"]/1e6:.3f} M  chain256 {d["roofline"]["chain_ms_per_pass"]:.2f} ms  chain32 {d["roofline_one_batch"]["chain_ms_per_pass"]:.2f} ms  conv {d["roofline_conv_gemm"]["achieved"]:.1f} TF')
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
cat gpurun_out/knobs.log
