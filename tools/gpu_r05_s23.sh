#!/bin/bash
# round 5, session 23: ring engine with tiles dealt to the XCDs (operand-sharing blocks) vs its plain grid: layers in isolation, then the face batch
set -u
O=gpurun_out/r05_s23; mkdir -p $O
cd /root/repo
timeout 300 python tools/ring_xcd_probe.py > $O/ring_xcd_probe.txt 2>&1
for r in 9 7 9 7; do
  echo "== TS_CONV_RING=$r" >> $O/face_layers_xcd.txt
  TS_CONV_RING=$r timeout 200 python tools/face_layers.py 2>&1 | grep -v "^\[ts_prof\] conv M=19200 N=\(2304\|768\|3072\)" >> $O/face_layers_xcd.txt
done
tail -30 $O/ring_xcd_probe.txt; tail -40 $O/face_layers_xcd.txt
