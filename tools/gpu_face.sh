#!/bin/bash
# face generator: parity tests of everything that goes through it, then its bench blocks
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-face}; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "face or whole_body or reference_callers" 2>&1 | tail -8 | tee $O/tests.log
timeout 400 python - <<'PY' | tee $O/face_bench.txt
import json, sys
sys.path.insert(0, '.')
import bench, torch
torch.cuda.set_device(0)
f = bench.face_block(0)
print(json.dumps({k: f[k] for k in ('frames_per_s', 'ms_per_batch', 'conv_gemm_f32', 'attention_fused', 'other_kernels_ms')}))
print(json.dumps(f['split_bf16']))
PY
