#!/bin/bash
# round 5, session 28: the ring engine banded + dealt (tile 37; TS_CONV_RING_PAIRED=1 gives it the paired body + hand layers): parity,
# single-problem shapes in isolation, the conv stacks of a 256-clip pass alone, the driver's bench command
O=gpurun_out/r05_s28; mkdir -p $O
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "conv_tile or conv_banded or strided_conv" 2>&1 | tail -3 | tee $O/tests.log
TS_TILES=39,35,36,37 TS_SHAPES=4,5,6,7,8,9,10,11 TS_ROUNDS=3 timeout 300 python tools/ring_xcd_probe.py 2>&1 | grep -v amdgpu | tee $O/ring_banded_probe.txt
bash tools/conv_mix_ab.sh "TS_CONV_RING_PAIRED=0" "TS_CONV_RING_PAIRED=1" 2>&1 | tee $O/conv_stacks_ab.txt
for v in 0 1 0 1 0 1 0 1; do
echo "TS_CONV_RING_PAIRED=$v" >> $O/bench_ab.txt
TS_CONV_RING_PAIRED=$v TS_BENCH_WATCHDOG=200 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-face --no-modes 2>> $O/bench.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value %.3f M ms/step %.3f chain %.2f conv frac %.3f selfcheck %s' % (d['value']/1e6, d['ms_per_step'], d['roofline']['chain_ms_per_pass'], d['roofline_conv_gemm']['frac'], d['selfcheck']))" >> $O/bench_ab.txt
done
cat $O/bench_ab.txt
