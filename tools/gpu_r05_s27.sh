#!/bin/bash
# round 5, session 27: does the conv stacks' fabric traffic matter to the chains they co-run with?  The paired body + hand layers on the ring
# engine with dealt tiles (TS_CONV_RING_PAIRED=1) against the banded launch, on the driver's bench command (three passes in flight)
O=gpurun_out/r05_s27; mkdir -p $O
cd /root/repo
for v in 0 1 0 1 0 1; do
echo "TS_CONV_RING_PAIRED=$v" >> $O/bench_ab.txt
TS_CONV_RING_PAIRED=$v TS_BENCH_WATCHDOG=200 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-face --no-modes 2>> $O/bench.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value %.3f M ms/step %.3f chain %.2f conv frac %.3f selfcheck %s' % (d['value']/1e6, d['ms_per_step'], d['roofline']['chain_ms_per_pass'], d['roofline_conv_gemm']['frac'], d['selfcheck']))" | tee -a $O/bench_ab.txt
done
cat $O/bench_ab.txt
