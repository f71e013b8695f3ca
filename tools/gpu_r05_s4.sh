#!/bin/bash
# round 5, session 4: what a lone workgroup reaches (256 / 128 tiles), smaller tiles, 8-wave workgroups
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r05_s4}; mkdir -p $O
cd $R
TS_TILES=1,31,36,37,38,39 TS_SHAPES=0,1,2,3,4,6,10,11,12,13,14 timeout 400 python tools/ring_probe.py > $O/ring_probe.txt 2>&1
cat $O/ring_probe.txt | tail -14
