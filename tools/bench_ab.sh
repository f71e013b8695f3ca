#!/bin/bash
# same-box A/B of environment settings on the driver's bench command (no CPU baseline / face / modes): bench_ab.sh "ENV=.. ENV=.." "ENV=.." ...
for round in 1 2; do
for e in "$@"; do
  line=$(env $e TS_BENCH_WATCHDOG=150 timeout 200 python bench.py --steps 24 --warmup 8 --no-cpu-baseline --no-face --no-modes 2>/dev/null | tail -1)
  python - "$e" "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
r, r1 = d["roofline"], d.get("roofline_one_batch", {})
print(f'{sys.argv[1]:40s} value {d["value"]/1e6:.3f} M  ms/step {d["ms_per_step"]:.3f}  chain256 {r["chain_ms_per_pass"]:.2f} ms frac {r["frac"]:.3f}  chain32 {r1.get("chain_ms_per_pass", 0):.2f} ms')
PY
done; done
