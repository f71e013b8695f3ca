#!/bin/bash
# round 5, session 16: whole-body step with two face streams; face block with two batches in flight; whole-body / gloo parity tests
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r05_s16}; mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -q -x -k "whole_body or face_golden or face_10s" 2>&1 | tail -3 | tee $O/tests.log
timeout 500 python - <<'PY' | tee $O/blocks.txt
import json, sys
sys.path.insert(0, '.')
import bench, torch
from talkshow_amd import _lib
torch.cuda.set_device(0)
f = bench.face_block(0)
print("face", json.dumps({k: f[k] for k in ('frames_per_s', 'ms_per_batch', 'two_batches_in_flight')}))
w, _ = bench.build_models(0)
wb = bench.whole_body_block(w, _lib, 0, 256)
print("whole_body", json.dumps(wb))
PY
