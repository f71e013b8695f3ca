#!/bin/bash
# bench A/B over explicit pass plans for the driver's 20 steps: plan_env_ab.sh "8,8,4" "8,6,6" ...
for round in 1 2; do
for pl in "$@"; do
  line=$(TS_BENCH_PLAN=$pl TS_BENCH_WATCHDOG=200 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-face --no-modes --no-roofline 2>/dev/null | tail -1)
  python - "$pl" "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print(f'plan {sys.argv[1]:8s} -> {d["config"].get("batches_per_pass")} value {d["value"]/1e6:.3f} M  ms/step {d["ms_per_step"]:.3f}')
PY
done; done
