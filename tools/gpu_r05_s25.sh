#!/bin/bash
# round 5, session 25: conv_taps48.hip (the positional conv unpadded on the 16 x 16 MFMA): unit test, the kernel alone, face A/B
set -u
O=gpurun_out/r05_s25; mkdir -p $O
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "taps48 or strided_conv or face_golden or face_10s or conv_tile" > $O/tests.log 2>&1
tail -15 $O/tests.log
timeout 200 python tools/taps48_probe.py 2>&1 | grep -v amdgpu | tee $O/taps48_probe.txt
for r in 0 1 0 1; do
  echo "== TS_CONV_TAPS48=$r" >> $O/face_layers_taps48.txt
  TS_CONV_TAPS48=$r timeout 200 python tools/face_layers.py 2>&1 | grep "groups=16\|conv total" >> $O/face_layers_taps48.txt
done
cat $O/face_layers_taps48.txt
