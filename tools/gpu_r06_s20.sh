#!/bin/bash
mkdir -p gpurun_out/r06_s20
timeout 300 python tools/gelu_cost.py 2>&1 | tail -5 | tee gpurun_out/r06_s20/gelu_cost.txt
