#!/bin/bash
# round 5, session 4: what a lone workgroup reaches (256 / 128 tiles), smaller tiles, 8-wave workgroups
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r05_s4}; mkdir -p $O
cd $R
TS_TILES=31,44,39,45 TS_SHAPES=0,1,2,3,4,5,6,7,9,10,11 timeout 400 python tools/ring_probe.py > $O/ring_probe.txt 2>&1
cat $O/ring_probe.txt | tail -14
