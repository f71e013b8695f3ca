#!/bin/bash
# r06 session 8: stream-K band with the last-arriver reduction: layer A/B, face pass per layer SK 0 / 1, face + canary + real-audio tests, bench SK 0 / 1
mkdir -p gpurun_out/r06_s8
O=gpurun_out/r06_s8
timeout 600 python tools/sk_layers.py > $O/sk_layers.txt 2>$O/sk_layers.err
cut -c1-330 $O/sk_layers.txt; tail -3 $O/sk_layers.err
for sk in 0 1; do
  TS_CONV_SK=$sk timeout 300 python tools/face_layers.py 2>$O/face_layers_sk$sk.err | tail -1 > $O/face_sk$sk.txt
  cat $O/face_sk$sk.txt
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_real_audio.py tests/test_gpu_canary.py -m gpu -x -q -k "face or canary" 2>&1 | tail -6 | tee $O/tests.log
for round in 1 2; do
for sk in 0 1; do
  line=$(TS_CONV_SK=$sk TS_BENCH_WATCHDOG=150 timeout 300 python bench.py --steps 24 --warmup 8 --no-cpu-baseline --no-modes 2>>$O/bench.err | tail -1)
  python - "$sk" "$line" <<'PY' | tee -a gpurun_out/r06_s8/ab.txt
import json, sys
d = json.loads(sys.argv[2])
r, c = d["roofline"], d["roofline_conv_gemm"]
print(f'TS_CONV_SK={sys.argv[1]} value {d["value"]/1e6:.3f} M chain frac {r["frac"]:.3f} conv-in-pass {c["achieved"]:.1f} TF | face {d["face"]["ms_per_batch"]:.2f} ms conv {d["face"]["conv_gemm_f32"]["achieved_TFLOPs"]:.1f} TF two-in-flight {d["face"]["two_batches_in_flight"]["ms_per_batch"]:.2f} | whole_body {d["whole_body"].get("fp32", d["whole_body"])}')
PY
done; done
tail -3 $O/bench.err
