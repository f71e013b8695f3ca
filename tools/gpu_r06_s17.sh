#!/bin/bash
mkdir -p gpurun_out/r06_s17
timeout 600 python -m pytest tests/test_gpu_operating_points.py -m gpu -q -s -k "bounded_graph_cache" 2>&1 | grep -E "per-row|passed|failed|Error" | tee gpurun_out/r06_s17/cache.log
