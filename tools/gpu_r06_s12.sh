#!/bin/bash
mkdir -p gpurun_out/r06_s12
timeout 900 python -m pytest tests/test_reference_callers.py -m gpu -q -x 2>&1 | tail -40 > gpurun_out/r06_s12/callers.log
tail -40 gpurun_out/r06_s12/callers.log
