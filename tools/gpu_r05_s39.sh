#!/bin/bash
# round 5, session 39: the round's last run: whole GPU suite, smoke, bench x2, face kernel stats
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r05_s39; mkdir -p $O
cd $R
export TS_MEASURED_LOG=$O/measured_errors.jsonl
rm -f $TS_MEASURED_LOG
bash tools/gpu_final.sh r05_s39
unset TS_MEASURED_LOG
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_face -- python $R/tools/face_pass.py > $O/stats_face.log 2>&1; tail -1 $O/stats_face.log | cut -c1-160
cp "$(find $O/stats_face -name '*kernel_stats.csv' | head -1)" $O/stats_face.csv; rm -rf $O/stats_face
