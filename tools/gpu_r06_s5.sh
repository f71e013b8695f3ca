#!/bin/bash
# r06 session 5: graph-cache policy test, then the whole GPU suite
mkdir -p gpurun_out/r06_s5
O=gpurun_out/r06_s5
timeout 600 python -m pytest tests/test_gpu_operating_points.py -m gpu -q -s -k "bounded_graph_cache or queued" 2>&1 | tail -40 > $O/graph_cache.log
TS_MEASURED_LOG=$O/measured.jsonl timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/suite.log
tail -12 $O/graph_cache.log; tail -12 $O/suite.log
