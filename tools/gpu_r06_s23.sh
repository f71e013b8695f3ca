#!/bin/bash
mkdir -p gpurun_out/r06_s23
timeout 600 python -m pytest tests/test_gpu_real_audio.py -m gpu -q -s -k "second_weight_set or stochastic" 2>&1 | grep -E "equal to|draws equal|measured|passed|failed|Error" | tee gpurun_out/r06_s23/w11.log
