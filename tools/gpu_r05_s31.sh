#!/bin/bash
# round 5, session 31: end-of-round run with the wide kernel's pair order: whole GPU suite, smoke, bench x2, profile set
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05_s31; mkdir -p $O
cd $R
export TS_MEASURED_LOG=$O/measured_errors.jsonl
rm -f $TS_MEASURED_LOG
bash tools/gpu_final.sh r05_s31
unset TS_MEASURED_LOG
bash tools/profile_r05.sh > $O/profile.log 2>&1
tail -3 $O/profile.log
