#!/bin/bash
# r06 session 2: the real-audio parity tests (reference-made goldens on the reference's recordings) + smoke()
mkdir -p gpurun_out/r06_s2
O=gpurun_out/r06_s2
TS_MEASURED_LOG=$O/measured.jsonl timeout 900 python -m pytest tests/test_gpu_real_audio.py -m gpu -q -s 2>&1 | grep -v "^$" | tail -80 > $O/tests.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1
tail -60 $O/tests.log; tail -5 $O/smoke.log
