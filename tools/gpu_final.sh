#!/bin/bash
# end-of-round run: the whole GPU suite, then the driver's bench command twice (the second line is kept as profiles/rNN_bench_full.json)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-final}; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -s > $O/tests_full.log 2>&1
grep -E "equal to the reference|chi-square|first difference|passed|failed|FAILED|^E  " $O/tests_full.log | tail -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for i in 1 2; do timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench$i.json 2> $O/bench$i.err; python -c "
import json,sys; d=json.loads(open('$O/bench$i.json').read().strip().splitlines()[-1]); print('run $i: value %.3f M ms/step %.3f runs_ms %s selfcheck %s captures %s frac %.3f conv %.3f face %.1f ms (conv %.3f) whole_body %.1f ms cpu %.0f (%s) enqueue cpu %.4f s one-batch %.2f ms wav_in %.3f M' % (d['value']/1e6, d['ms_per_step'], [round(x,1) for x in d['runs_ms']], d['selfcheck'], d['graph_captures_in_timed_regions'], d['roofline']['frac'], d['roofline_conv_gemm']['frac'], d['face']['ms_per_batch'], d['face']['conv_gemm_f32']['frac_of_fp32_mfma_peak'], d['whole_body']['fp32']['ms_per_step'], d['cpu_baseline']['value'], d['cpu_baseline']['kind'], d['host_enqueue_cpu_s'], d['modes']['one_batch_in_flight']['ms_per_step'], d['modes']['wav_in']['frames_per_s']/1e6))"; done
