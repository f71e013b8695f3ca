#!/bin/bash
# round 5, session 2: ring engine timeline (trace build), Philox-by-argument tests
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r05_s2}; mkdir -p $O
cd $R
timeout 250 python tools/ring_trace.py > $O/ring_trace.txt 2>&1
cat $O/ring_trace.txt | tail -80
timeout 600 python -m pytest tests/test_gpu_operating_points.py tests/test_gpu_parity.py -m gpu -q -k "queued_stochastic or full_size_sampling or stream or philox or continuity" > $O/tests.log 2>&1
tail -5 $O/tests.log
