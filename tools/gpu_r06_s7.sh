#!/bin/bash
# r06 session 7: stream-K band in the body pass (paired body + hand layers) and the whole-body step: TS_CONV_SK = 0 / 1 / 2, same box, two rounds
mkdir -p gpurun_out/r06_s7
O=gpurun_out/r06_s7
for round in 1 2; do
for sk in 0 1 2; do
  line=$(TS_CONV_SK=$sk TS_BENCH_WATCHDOG=150 timeout 300 python bench.py --steps 24 --warmup 8 --no-cpu-baseline --no-modes 2>/dev/null | tail -1)
  python - "$sk" "$line" <<'PY' | tee -a gpurun_out/r06_s7/ab.txt
import json, sys
d = json.loads(sys.argv[2])
r, c = d["roofline"], d["roofline_conv_gemm"]
print(f'TS_CONV_SK={sys.argv[1]} value {d["value"]/1e6:.3f} M runs {[round(x,1) for x in d["runs_ms"]]} chain frac {r["frac"]:.3f} conv-in-pass {c["achieved"]:.1f} TF {c["ms_per_pass"]:.2f} ms | face {d["face"]["ms_per_batch"]:.2f} ms conv {d["face"]["conv_gemm_f32"]["achieved_TFLOPs"]:.1f} TF | whole_body {d["whole_body"]["fp32"]["ms_per_step"]:.2f} ms')
PY
done; done
