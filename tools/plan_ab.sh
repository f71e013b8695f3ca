#!/bin/bash
# bench A/B over pass plans: plan_ab.sh "G S" "G S" ...   (driver's step count)
for round in 1 2; do
for gs in "$@"; do
  set -- $gs
  line=$(TS_BENCH_WATCHDOG=200 timeout 300 python bench.py --steps 20 --warmup 5 --coalesce $1 --streams $2 --no-cpu-baseline --no-face --no-modes 2>/dev/null | tail -1)
  python - "$gs" "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
r = d["roofline"]
print(f'G,S = {sys.argv[1]:6s} plan {d["config"].get("batches_per_pass")} value {d["value"]/1e6:.3f} M  ms/step {d["ms_per_step"]:.3f}  chain {r["clips_per_stage"]} clips {r["chain_ms_per_pass"]:.2f} ms frac {r["frac"]:.3f}')
PY
done; done
