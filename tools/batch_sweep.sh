#!/bin/bash
# Sweep of clips-per-chain (M of the PixelCNN stage GEMMs) x batches in flight: how chain / conv time scale with M.
# Usage (GPU box): bash tools/batch_sweep.sh "32 128 256" "1 2" > gpurun_out/batch_sweep.jsonl   (env passes through)
BS=${1:-"32 64 128 256"}; SS=${2:-"1 2 4"}
for B in $BS; do
  for S in $SS; do
    steps=$(( 24 * 32 / B )); [ $steps -lt 4 ] && steps=4
    timeout 300 python bench.py --batch $B --streams $S --steps $steps --warmup $S --no-cpu-baseline --no-face 2>/dev/null | tail -1
  done
done
