#!/bin/bash
# r06 session 16: what the wide engine makes of the chain's head launches (128 tiles of 64 x 64 on 256 CUs) — the measurable bound on the unbuilt
# 32 x 64 head tile: TS_SKINNY_WIDE_MIN=128 sends them to the wide kernel (default 160: they run as 256 split-K workgroups of 64 x 32).
# rocprofv3 kernel stats of one 256-clip chain pass per setting + the bench's chain figure, twice.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_s16; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for wm in 160 128; do
  TS_SKINNY_WIDE_MIN=$wm timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/wm$wm -- python $R/tools/chain_pass.py --batch 256 --passes 3 > $O/wm$wm.log 2>&1
  cp "$(find $O/wm$wm -name '*kernel_stats.csv' | head -1)" $O/stats_wm$wm.csv; rm -rf $O/wm$wm
  echo "== TS_SKINNY_WIDE_MIN=$wm"; head -8 $O/stats_wm$wm.csv | cut -c1-160
done
cd $R
for round in 1 2; do for wm in 160 128; do
  line=$(TS_SKINNY_WIDE_MIN=$wm TS_BENCH_WATCHDOG=150 timeout 300 python bench.py --steps 24 --warmup 8 --no-cpu-baseline --no-face --no-modes 2>/dev/null | tail -1)
  python - "$wm" "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[2]); r = d["roofline"]; r1 = d["roofline_one_batch"]
print(f'WIDE_MIN={sys.argv[1]} value {d["value"]/1e6:.3f} M chain256 {r["chain_ms_per_pass"]:.2f} ms frac {r["frac"]:.3f} avg launch {r["avg_launch_us"]:.2f} us | chain32 {r1["chain_ms_per_pass"]:.2f} ms')
PY
done; done
