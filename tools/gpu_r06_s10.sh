#!/bin/bash
# r06 session 10: the stream-K tests, the alternate-path children, then the whole GPU suite and one default bench line
mkdir -p gpurun_out/r06_s10
O=gpurun_out/r06_s10
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_real_audio.py -m gpu -q -s -k "stream_k or batch_64 or face_gemms" 2>&1 | grep -v "^$" | tail -120 > $O/sk_tests.log
tail -20 $O/sk_tests.log
TS_MEASURED_LOG=$O/measured.jsonl timeout 2000 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/suite.log
tail -8 $O/suite.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench.err | tail -1 > $O/bench_line.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_s10/bench_line.json"))
print(d["value"], d["runs_ms"], d["roofline"]["frac"], d["face"]["ms_per_batch"], d["face"]["conv_gemm_f32"], d["whole_body"]["fp32"]["ms_per_step"], d["cpu_baseline"])
PY
