#!/bin/bash
# r06 session 15: the N > 1 code path over RCCL at world 1 with three timed regions per line; the no-flag default bench; a forced NUMA pin
mkdir -p gpurun_out/r06_s15
bash tools/rccl_smoke.sh r06_s15 2>&1 | tail -6 | cut -c1-700
TS_BENCH_PIN=1 timeout 600 python bench.py --no-face --no-modes --no-cpu-baseline 2>gpurun_out/r06_s15/default.err | tail -1 > gpurun_out/r06_s15/default_line.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_s15/default_line.json"))
print("no-flag default:", d["steps"], d["warmup"], "value %.3f M" % (d["value"] / 1e6), d["runs_ms"], d["config"]["batches_per_pass"], d["host_affinity"], d["selfcheck"], d["graph_captures_in_timed_regions"])
PY
