#!/bin/bash
mkdir -p gpurun_out/r06_s14
timeout 900 python -m pytest tests/test_reference_callers.py -m gpu -q -x -k "diversity or continuity" 2>&1 | tail -40 > gpurun_out/r06_s14/callers.log
tail -40 gpurun_out/r06_s14/callers.log
