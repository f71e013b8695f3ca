#!/bin/bash
# persistent chain kernel vs the launch graph: pass time at 256 / 128 / 512 clips, then the golden-clip parity tests
export TS_CHAIN_PERSIST_DEBUG=1
for B in ${BATCHES:-256 128 512}; do
  for P in ${MODES:-0 1 2}; do
    echo -n "TS_CHAIN_PERSIST=$P  "; TS_CHAIN_PERSIST=$P timeout 120 python tools/chain_pass.py --batch $B --passes 6 2>&1 | tail -2 | tr '\n' ' '; echo
  done
done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden_clips or pixelcnn_golden or pixelcnn_sampling" 2>&1 | tail -5
