#!/bin/bash
# round 4, GPU session 2: full GPU suite (complete log), the shared-CU wide kernel: chain || chain and the 3-stream bench, A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04_s2; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q -s > $O/tests_full.log 2>&1
grep -E "equal to the reference|chi-square|first difference|passed|failed|FAILED|Error" $O/tests_full.log | tail -30
for lean in 0 1; do
  TS_SKINNY_WIDE_LEAN=$lean timeout 200 python tools/chain_corun.py 2>&1 | grep chain | sed "s/^/lean=$lean /" | tee -a $O/chain_corun.txt
done
for lean in 0 1 0 1; do
  TS_SKINNY_WIDE_LEAN=$lean timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-face 2> $O/bench_lean$lean.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('lean=$lean value %.3f M  ms/step %.3f  coalesced %.3f  one-batch %.2f  chain256 %.2f ms frac %.3f  selfcheck %s' % (d['value']/1e6, d['ms_per_step'], d['modes']['coalesced']['ms_per_step'], d['modes']['one_batch_in_flight']['ms_per_step'], d['roofline']['chain_ms_per_pass'], d['roofline']['frac'], d.get('selfcheck')))" | tee -a $O/bench_ab.txt
done
