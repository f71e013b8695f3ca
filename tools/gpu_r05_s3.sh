#!/bin/bash
# round 5, session 3: coalesced (LDS-staged) epilogue vs register epilogue, both engines; parity of everything conv
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r05_s3}; mkdir -p $O
cd $R
timeout 300 python tools/ring_probe.py > $O/ring_probe.txt 2>&1
cat $O/ring_probe.txt | tail -14
TS_SHAPES=0,1 timeout 100 python tools/ring_trace.py > $O/ring_trace_staged.txt 2>&1
TS_SHAPES=0,1 TS_TRACE_TILE=141 timeout 100 python tools/ring_trace.py > $O/ring_trace_regs.txt 2>&1
grep -E "==|epilogue|ptr setup|lifetime" $O/ring_trace_staged.txt $O/ring_trace_regs.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "op_conv1d or conv_tile or conv_banded or vqvae or audioenc or face_golden or face_10s or wrapper_body_vq or golden_clips or ragged or 6d" > $O/tests.log 2>&1
tail -5 $O/tests.log
