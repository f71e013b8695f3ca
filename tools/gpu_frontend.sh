#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/${1:-fe}
timeout 900 python -m pytest tests -m gpu -q -x -k "mfcc or wav_in or kaiser or reference_callers or wrapper_body_pixel_e2e" 2>&1 | tail -5
timeout 400 python - <<'PY' | tee gpurun_out/${1:-fe}/frontend.txt
import json, sys
sys.path.insert(0, '.')
import bench, torch
from talkshow_amd import _lib
torch.cuda.set_device(0)
w, _ = bench.build_models(0)
f = bench.frontend_block(w, _lib, 256)
print(json.dumps(f))
PY
