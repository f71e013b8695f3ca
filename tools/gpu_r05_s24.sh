#!/bin/bash
# round 5, session 24: TS_CONV_DEAL bits — ring tiles dealt to the XCDs (1), the grouped positional conv likewise (2), paired layers on
# the ring engine with dealt tiles instead of the banded launch (4): parity of the conv / face tests, then same-box A/B
set -u
O=gpurun_out/r05_s24; mkdir -p $O
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "conv or face or tile" > $O/tests_conv_face.log 2>&1
tail -3 $O/tests_conv_face.log
for r in 0 1 3 0 1 3; do
  echo "== TS_CONV_DEAL=$r" >> $O/face_layers_deal.txt
  TS_CONV_DEAL=$r timeout 200 python tools/face_layers.py 2>&1 | grep -v "^\[ts_prof\] conv M=19200 N=\(2304\|768\|3072\) " >> $O/face_layers_deal.txt
done
grep "==\|conv total\|N=48 " $O/face_layers_deal.txt
bash tools/conv_mix_ab.sh "TS_CONV_DEAL=3" "TS_CONV_DEAL=7" 2>&1 | tee $O/conv_stacks_deal.txt
