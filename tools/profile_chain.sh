#!/bin/bash
# rocprofv3 passes over the PixelCNN chain at one operating point (GPU box).  usage: profile_chain.sh <batch> <tag>
# One counter set per pass, kernel-trace only (no --stats / sys-trace together with --pmc).
B=${1:-256}; TAG=${2:-M$B}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof_chain_$TAG
cd /tmp && export TMPDIR=/tmp
rm -rf $O; mkdir -p $O
run() { name=$1; shift; timeout 240 rocprofv3 --kernel-trace "$@" --output-format csv -d $O/$name -- python $R/tools/chain_pass.py --batch $B --passes 2 > $O/$name.log 2>&1; tail -1 $O/$name.log; }
run stats --stats
run sq   --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run tcc  --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
du -sh $O
