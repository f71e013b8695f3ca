#!/bin/bash
# round 5, session 35: conv0's GroupNorm statistics from the input's second moments (TS_W2V_MOMENTS): face parity tests, face batch A/B
O=gpurun_out/r05_s35; mkdir -p $O
cd /root/repo
export TS_MEASURED_LOG=$O/measured_errors.jsonl
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_reference_callers.py -m gpu -q -x -k "face or demo_py" 2>&1 | tail -3 | tee $O/tests.log
unset TS_MEASURED_LOG
for r in 0 1 0 1; do
  echo "== TS_W2V_MOMENTS=$r" >> $O/face_moments_ab.txt
  TS_W2V_MOMENTS=$r timeout 200 python tools/face_layers.py 2>&1 | grep "conv total" >> $O/face_moments_ab.txt
  TS_W2V_MOMENTS=$r timeout 200 python tools/face_pass.py --passes 5 2>&1 | tail -1 >> $O/face_moments_ab.txt
done
cat $O/face_moments_ab.txt; grep -i "face" $O/measured_errors.jsonl | head -12
