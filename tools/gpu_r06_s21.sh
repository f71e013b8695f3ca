#!/bin/bash
# r06 session 21: branch-free GELU in the conv epilogues: cost (tools/gelu_cost.py), face tests, face pass per family (twice), bench face block
mkdir -p gpurun_out/r06_s21
O=gpurun_out/r06_s21
timeout 300 python tools/gelu_cost.py 2>&1 | tail -3 | tee $O/gelu_cost.txt
TS_MEASURED_LOG=$O/measured.jsonl timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_real_audio.py tests/test_gpu_canary.py -m gpu -x -q -k "face or strided_conv or taps48" 2>&1 | tail -4 | tee $O/tests.log
grep -E "face|gelu|taps" $O/measured.jsonl | cut -c1-160 | tail -30
for i in 1 2; do timeout 300 python tools/face_layers.py 2>/dev/null | tail -1 | tee -a $O/face.txt; done
