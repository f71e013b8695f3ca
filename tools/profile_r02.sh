#!/bin/bash
# Round-2 profile set (GPU box): rocprofv3 kernel stats of the bench command in its three execution modes, then the PMC
# passes (one counter set per run, --kernel-trace only) over the PixelCNN chain at M = 256 and M = 32 clips per stage.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r02_profiles
cd /tmp && export TMPDIR=/tmp
rm -rf $O; mkdir -p $O
stats() { tag=$1; shift; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -- python $R/bench.py "$@" --no-face --no-cpu-baseline --no-modes --no-roofline > $O/$tag.log 2>&1; tail -1 $O/$tag.log | cut -c1-200; find $O/$tag -name "*kernel_trace.csv" -delete; }
stats stats_default --steps 24 --warmup 8
stats stats_single_stream --steps 16 --warmup 8 --coalesce 8 --streams 1
stats stats_one_batch --steps 8 --warmup 2 --coalesce 1 --streams 1
for B in 256 32; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_M${B}_$c -- python $R/tools/chain_pass.py --batch $B --passes 2 --convs > $O/pmc_M${B}_$c.log 2>&1
    find $O/pmc_M${B}_$c -name "*kernel_trace.csv" -delete
  done
  timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_M${B}_SQ -- python $R/tools/chain_pass.py --batch $B --passes 2 --convs > $O/pmc_M${B}_SQ.log 2>&1
  find $O/pmc_M${B}_SQ -name "*kernel_trace.csv" -delete
  timeout 240 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/pmc_M${B}_TCC -- python $R/tools/chain_pass.py --batch $B --passes 2 --convs > $O/pmc_M${B}_TCC.log 2>&1
  find $O/pmc_M${B}_TCC -name "*kernel_trace.csv" -delete
done
find $O -name "*agent_info.csv" -delete
du -sh $O
