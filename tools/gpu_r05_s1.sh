#!/bin/bash
# round 5, session 1: the parity witnesses (equality asserts, measured logit errors, bf16x3 at baseline shapes), smoke(), ring probe
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r05_s1}; mkdir -p $O
cd $R
export TS_MEASURED_LOG=$O/measured_errors.jsonl
rm -f $TS_MEASURED_LOG
timeout 250 python tools/ring_probe.py > $O/ring_probe.txt 2>&1
tail -14 $O/ring_probe.txt
timeout 600 python -m pytest tests/test_gpu_operating_points.py tests/test_gpu_parity.py -m gpu -q -s -k "golden_counts or pixelcnn_golden or constructor_variants or single_layer or split_bf16" > $O/tests.log 2>&1
grep -E "equal to the reference|max \|pose|measured\]|passed|failed|FAILED|^E  " $O/tests.log | tail -40
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.txt
