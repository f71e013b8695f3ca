#!/bin/bash
# rocprofv3 --kernel-trace --stats of the chain at one operating point under extra env: stats_chain.sh <batch> <tag> [ENV=VAL ...]
B=${1:-256}; TAG=${2:-M$B}; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/stats_$TAG
cd /tmp && export TMPDIR=/tmp
rm -rf $O; mkdir -p $O
env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python $R/tools/chain_pass.py --batch $B --passes 2 > $O/run.log 2>&1
tail -1 $O/run.log
f=$(find $O -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/kernel_stats_$TAG.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
head -12 $R/gpurun_out/kernel_stats_$TAG.csv | cut -c1-200
