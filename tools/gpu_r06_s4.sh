#!/bin/bash
# r06 session 4: canary harness, thread contract, graph-cache policy, then the whole GPU suite and three default bench lines
mkdir -p gpurun_out/r06_s4
O=gpurun_out/r06_s4
timeout 900 python -m pytest tests/test_gpu_canary.py tests/test_gpu_threads.py -m gpu -q 2>&1 | tail -40 > $O/canary_threads.log
timeout 600 python -m pytest tests/test_gpu_operating_points.py -m gpu -q -s -k "bounded_graph_cache or queued" 2>&1 | tail -40 > $O/graph_cache.log
TS_MEASURED_LOG=$O/measured.jsonl timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/suite.log
for i in 1 2 3; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>>$O/bench.err | tail -1 >> $O/bench_lines.jsonl
done
tail -15 $O/canary_threads.log; tail -12 $O/graph_cache.log; tail -8 $O/suite.log
python - <<'PY'
import json
for l in open("gpurun_out/r06_s4/bench_lines.jsonl"):
    try:
        d = json.loads(l)
        print(d["value"], d["ms_per_step"], d.get("runs_ms"), d.get("host_cpu_s"), d["roofline"]["frac"], d.get("selfcheck"), d.get("graph_captures_in_timed_regions"),
              d["modes"]["one_batch_in_flight"]["ms_per_step"], d["modes"]["coalesced"]["frames_per_s"], d["modes"].get("wav_in"))
    except Exception as e:
        print("bad line", e, l[:300])
PY
tail -5 $O/bench.err
