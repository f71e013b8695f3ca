#!/bin/bash
# round 5, session 10: real-audio fixtures through the lifted demo.py, gate activation accuracy, RCCL plumbing at world 1, whole suite
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r05_s10}; mkdir -p $O
cd $R
export TS_MEASURED_LOG=$O/measured_errors.jsonl
rm -f $TS_MEASURED_LOG
timeout 1500 python -m pytest tests -m gpu -q -s > $O/tests_full.log 2>&1
grep -E "equal to the reference|device MFCC|gate_act|passed|failed|FAILED|^E  " $O/tests_full.log | tail -30
bash tools/rccl_smoke.sh $(basename $O) 2>&1 | tail -12
