#!/bin/bash
# round 5, session 9: bounded graph cache + chunk graphs, ring engine parity in its production routing, bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r05_s9}; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_operating_points.py tests/test_gpu_parity.py -m gpu -q -x -s -k "arbitrary_clip_lengths or queued_stochastic or stream or continuity or golden_counts or conv_tile or conv_banded or alternate_kernel or full_size_sampling or pixelcnn_golden" > $O/tests.log 2>&1
grep -E "passed|failed|FAILED|^E  |Error" $O/tests.log | tail -12
TS_BENCH_WATCHDOG=200 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-face --no-modes 2>> $O/bench.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value %.3f M ms/step %.3f chain %.2f frac %.3f conv frac %.3f selfcheck %s' % (d['value']/1e6, d['ms_per_step'], d['roofline']['chain_ms_per_pass'], d['roofline']['frac'], d['roofline_conv_gemm']['frac'], d.get('selfcheck')))" | tee $O/bench.txt
tail -3 $O/bench.err
