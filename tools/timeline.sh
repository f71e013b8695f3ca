#!/bin/bash
# kernel trace of the bench command (3 passes in flight) -> tools/timeline.py summary in gpurun_out/timeline.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/timeline
cd /tmp && export TMPDIR=/tmp
rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python $R/bench.py --steps ${STEPS:-24} --warmup 8 "$@" --no-face --no-cpu-baseline --no-modes --no-roofline > $O/bench.log 2>&1
tail -1 $O/bench.log | cut -c1-200
f=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python $R/tools/timeline.py "$f" | tee $R/gpurun_out/timeline${TAG}.txt
rm -rf $O
