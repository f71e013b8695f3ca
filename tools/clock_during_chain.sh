#!/bin/bash
# What sclk does the GPU run while the PixelCNN chain (latency-bound, matrix pipe 26 % busy) is all it has to do?
# Samples rocm-smi during a 60-pass chain loop; then the same under `--setperflevel high`, restoring `auto` afterwards.
run() {
  python tools/chain_pass.py --batch $1 --passes $2 > /tmp/cp.log 2>&1 &
  pid=$!
  sleep 6
  for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|fclk|mclk|Average Graphics|Current Socket" | tr -s ' ' | tr '\n' ';'; echo; sleep 0.25; done
  wait $pid; tail -1 /tmp/cp.log
}
echo "== auto, 32 clips";  run 32 400
echo "== auto, 256 clips"; run 256 150
rocm-smi --showperflevel 2>/dev/null | grep -i perf
rocm-smi --setperflevel high 2>&1 | tail -2
echo "== high, 32 clips";  run 32 400
echo "== high, 256 clips"; run 256 150
rocm-smi --setperflevel auto 2>&1 | tail -2
