#!/bin/bash
# round 5, session 29: end-of-round run with the ring engine's band plan (paired layers + FFN1): whole GPU suite, smoke, bench x2, profile set, face PMC
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05_s29; mkdir -p $O
cd $R
export TS_MEASURED_LOG=$O/measured_errors.jsonl
rm -f $TS_MEASURED_LOG
bash tools/gpu_final.sh r05_s29
unset TS_MEASURED_LOG
bash tools/profile_r05.sh > $O/profile.log 2>&1
tail -3 $O/profile.log
bash tools/pmc_face_r05.sh 2>&1 | grep "all conv\|conv_ring" | tee $O/pmc_face.txt
