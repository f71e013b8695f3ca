#!/bin/bash
# Which HIP runtime knobs move the multi-stream throughput?  (informational)
run() { out=$(env "$@" timeout 300 python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-face --no-roofline --streams ${S:-4} 2>&1 | tail -1); python - "$out" "$*" <<'PY'
import json, sys
try:
    d = json.loads(sys.argv[1]); print(f"{sys.argv[2]:55s} streams={d['streams']} {d['value']:9.0f} frames/s  {d['ms_per_step']:.2f} ms/step  latency {d['batch_latency_ms']:.2f} ms")
except Exception as e:
    print(sys.argv[2], "FAILED", sys.argv[1][-200:])
PY
}
run X=0
run DEBUG_CLR_SKIP_RELEASE_SCOPE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run DEBUG_HIP_DYNAMIC_QUEUES=0
run DEBUG_HIP_DYNAMIC_QUEUES=1
run GPU_STREAMOPS_CP_WAIT=1
run ROC_SYSTEM_SCOPE_SIGNAL=0
S=8 run DEBUG_HIP_DYNAMIC_QUEUES=0
S=8 run DEBUG_HIP_FORCE_GRAPH_QUEUES=8
S=8 run DEBUG_HIP_DYNAMIC_QUEUES=1 GPU_MAX_HW_QUEUES=16
