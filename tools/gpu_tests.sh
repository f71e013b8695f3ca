#!/bin/bash
# the whole GPU suite with a complete log (counts printed by the operating-point tests included), then one default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-gpu_tests}; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -s > $O/tests_full.log 2>&1
grep -E "equal to the reference|chi-square|first difference|passed|failed|FAILED|^E  " $O/tests_full.log | tail -40
if [ "$2" != "nobench" ]; then
  timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
  python - $O/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('value %.3f M  ms/step %.3f  selfcheck %s  coalesced %.3f  pcie %.3f  one-batch %.2f  chain256 %.2f ms frac %.3f  conv frac %.3f' % (
    d['value'] / 1e6, d['ms_per_step'], d.get('selfcheck'), d['modes']['coalesced']['ms_per_step'], d['modes']['pcie_inclusive'].get('ms_per_step', -1),
    d['modes']['one_batch_in_flight']['ms_per_step'], d['roofline']['chain_ms_per_pass'], d['roofline']['frac'], d['roofline_conv_gemm']['frac']))
print('face', {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d['face'].items() if k in ('frames_per_s', 'ms_per_batch', 'other_kernels_ms')})
print('whole_body', json.dumps(d.get('whole_body'))[:400])
PY
fi
