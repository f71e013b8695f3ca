#!/bin/bash
# round 4, session 3: what bounds several chains in flight?  The wide kernel's ablations (trace build: 2 = no loads, 4 = no MFMAs) with 1, 2, 3 chains
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04_s3; mkdir -p $O
cd $R
for abl in 0 2 4 6; do
  TS_SKINNY_TRACE=1 TS_SKINNY_WIDE_ABLATE=$abl TS_N=3 timeout 200 python tools/chain_corun.py 2>&1 | grep chain | sed "s/^/trace ablate=$abl /" | tee -a $O/ablate.txt
done
