#!/bin/bash
# r06 session 3: canary / poison harness over the C ABI at ragged shapes
mkdir -p gpurun_out/r06_s3
timeout 1200 python -m pytest tests/test_gpu_canary.py -m gpu -q 2>&1 | tail -120 > gpurun_out/r06_s3/canary.log
tail -100 gpurun_out/r06_s3/canary.log
