#!/bin/bash
# round 5, session 26: end-of-round run after the dealt ring tiles + conv_taps48: whole GPU suite (measured errors logged), smoke, the
# driver's bench command twice, the round's profile set
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05_s26; mkdir -p $O
cd $R
export TS_MEASURED_LOG=$O/measured_errors.jsonl
rm -f $TS_MEASURED_LOG
bash tools/gpu_final.sh r05_s26
unset TS_MEASURED_LOG
bash tools/profile_r05.sh > $O/profile.log 2>&1
tail -5 $O/profile.log
