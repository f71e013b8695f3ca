#!/bin/bash
mkdir -p gpurun_out/r06_s25
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r06_s25/bench.err | tail -1 > gpurun_out/r06_s25/bench.json
python -c "
import json; d=json.load(open('gpurun_out/r06_s25/bench.json')); m=d['modes']; print(d['value'], d['selfcheck'], {k: round(v['ms_per_step'],2) for k,v in m.items() if 'ms_per_step' in v})"
tail -2 gpurun_out/r06_s25/bench.err
