#!/bin/bash
mkdir -p gpurun_out/r06_s11
O=gpurun_out/r06_s11
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_real_audio.py -m gpu -q -s -k "stream_k or batch_64 or face_gemms" 2>&1 | grep -v "^$" | tail -120 > $O/sk_tests.log
grep -E "measured|stream-K vs|passed|failed|Error" $O/sk_tests.log | tail -30
