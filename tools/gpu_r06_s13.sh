#!/bin/bash
# r06 session 13: double-buffered attention: face tests, canary face cases, per-family times of a face batch (two runs)
mkdir -p gpurun_out/r06_s13
O=gpurun_out/r06_s13
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_real_audio.py tests/test_gpu_canary.py -m gpu -x -q -k "face" 2>&1 | tail -5 | tee $O/tests.log
for i in 1 2; do timeout 300 python tools/face_layers.py 2>/dev/null | tail -1 | tee -a $O/face.txt; done
