#!/bin/bash
# One metered GPU cannot run the 8-rank job, but it can prove the plumbing of `bench.py --gpus N`: the driver's launcher
# (torch.distributed.run, one rank), the `nccl` (= RCCL) process group, barriers, the all-gather of every step's rows and the MAX
# all-reduce of the N > 1 code path (TS_BENCH_FORCE_COLLECTIVES=1), for configs[1] and for configs[4].  Prints the `rccl` blocks.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-rccl_smoke}; mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 TS_BENCH_FORCE_COLLECTIVES=1
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 8 --warmup 2 \
  --no-cpu-baseline --no-face --no-modes --no-roofline > $O/body.json 2> $O/body.err
python - $O/body.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("configs[1] world 1 over RCCL:", "value %.3f M" % (d["value"] / 1e6), "selfcheck", d.get("selfcheck"), "rccl", json.dumps(d["rccl"]))
PY
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --config whole_body --steps 2 --warmup 1 \
  > $O/whole_body.json 2> $O/whole_body.err
tail -c 600 $O/whole_body.json; tail -3 $O/whole_body.err
