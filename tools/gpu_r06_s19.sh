#!/bin/bash
# r06 session 19: the whole GPU suite twice more on another box (flakiness check of the timing- and thread-based tests), smoke, one bench line
mkdir -p gpurun_out/r06_s19
for i in 1 2; do timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3 | tee -a gpurun_out/r06_s19/suite.log; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r06_s19/bench.json
python -c "
import json; d=json.load(open('gpurun_out/r06_s19/bench.json')); print(d['value'], d['runs_ms'], d['selfcheck'], d['roofline']['frac'], d['face']['ms_per_batch'], d['cpu_baseline']['value'], d['cpu_baseline']['kind'])"
