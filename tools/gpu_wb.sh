#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/${1:-wb}
timeout 600 python - <<'PY' | tee gpurun_out/${1:-wb}/whole_body.txt
import json, sys
sys.path.insert(0, '.')
import bench, torch
from talkshow_amd import _lib
torch.cuda.set_device(0)
w, _ = bench.build_models(0)
print(json.dumps(bench.whole_body_block(w, _lib, 0, 256)))
PY
timeout 300 python -m pytest tests -m gpu -q -x -k "whole_body" 2>&1 | tail -3
