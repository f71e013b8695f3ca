#!/bin/bash
# r06 session 24: one batch in flight (strict batch 32) with the VQ-encode half of the batch on a second stream (it feeds nothing downstream)
mkdir -p gpurun_out/r06_s24
for rep in 1 2; do for es in 0 1; do
  line=$(TS_BENCH_WATCHDOG=150 timeout 300 python bench.py --steps 8 --warmup 2 --coalesce 1 --streams 1 --enc-streams $es --no-cpu-baseline --no-face --no-modes --no-roofline 2>/dev/null | tail -1)
  python - "$es" "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print(f'enc-streams {sys.argv[1]}: {d["ms_per_step"]:.2f} ms per 32-clip batch, one at a time ({d["value"]/1e6:.3f} M frames/s) runs {[round(x,1) for x in d["runs_ms"]]} selfcheck {d["selfcheck"]}')
PY
done; done | tee gpurun_out/r06_s24/one_batch.txt
