#!/bin/bash
# round 5, session 30: the wide chain kernel's tile enumeration as (clip-block pair, column tile, block of the pair) (TS_SKINNY_WIDE_PAIR=1: a
# weight tile on two XCDs, an activation block on four — 2 W + 4 A instead of 4 W + 2 A bytes over the fabric): parity, bench A/B, chains in flight
O=gpurun_out/r05_s30; mkdir -p $O
cd /root/repo
TS_SKINNY_WIDE_PAIR=1 timeout 600 python -m pytest tests/test_gpu_operating_points.py -m gpu -q -x -k "golden_counts or full_size" 2>&1 | tail -3 | tee $O/tests.log
for v in 0 1 0 1 0 1 0 1; do
echo "TS_SKINNY_WIDE_PAIR=$v" >> $O/bench_ab.txt
TS_SKINNY_WIDE_PAIR=$v TS_BENCH_WATCHDOG=200 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-face --no-modes 2>> $O/bench.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value %.3f M ms/step %.3f chain %.2f frac %.4f conv frac %.3f selfcheck %s' % (d['value']/1e6, d['ms_per_step'], d['roofline']['chain_ms_per_pass'], d['roofline']['frac'], d['roofline_conv_gemm']['frac'], d['selfcheck']))" >> $O/bench_ab.txt
done
cat $O/bench_ab.txt
for v in 0 1; do echo "== TS_SKINNY_WIDE_PAIR=$v" >> $O/chain_corun.txt; TS_SKINNY_WIDE_PAIR=$v timeout 200 python tools/chain_corun.py 2>&1 | grep -v amdgpu | tail -6 >> $O/chain_corun.txt; done
cat $O/chain_corun.txt
