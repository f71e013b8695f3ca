#!/bin/bash
# round 5, session 20: the ring engine's 96 x 128 tile and the tile-count chooser: parity, per-shape probe, face A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r05_s20}; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "conv_tile or conv_banded or op_conv1d or face_golden or face_10s" 2>&1 | tail -3 | tee $O/tests.log
TS_TILES=0,39,33,7 TS_SHAPES=0,1,2,3,4,6 timeout 300 python tools/ring_probe.py 2>&1 | grep -v amdgpu | tee $O/ring_tall_tiles.txt
for v in 9 8 3 0 9 8 3 0; do
TS_CONV_RING=$v timeout 300 python - <<'PY' 2>&1 | tail -1 | tee -a $O/face_ab.txt
import json, os, sys
sys.path.insert(0, '.')
import bench, torch
torch.cuda.set_device(0)
f = bench.face_block(0)
print("TS_CONV_RING=" + os.environ["TS_CONV_RING"], json.dumps({k: round(f[k], 2) for k in ('frames_per_s', 'ms_per_batch')}), "conv ms", round(f['conv_gemm_f32']['ms'], 2), "TF", round(f['conv_gemm_f32']['achieved_TFLOPs'], 1))
PY
done
