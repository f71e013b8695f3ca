#!/bin/bash
# same-box A/B of library builds: bash tools/ab_libs.sh head sdiv cur  (tools/lib_<name>.so), two rounds each
cp talkshow_amd/lib/libtalkshow_hip.so /tmp/lib_keep.so
for round in 1 2; do
for v in "$@"; do
  cp tools/lib_$v.so talkshow_amd/lib/libtalkshow_hip.so
  line=$(TS_BENCH_WATCHDOG=150 timeout 200 python bench.py --steps 24 --warmup 8 --no-cpu-baseline --no-face --no-modes 2>/dev/null | tail -1)
  python - "$v" "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print(f'{sys.argv[1]:8s} value {d["value"]/1e6:.3f} M  chain256 {d["roofline"]["chain_ms_per_pass"]:.2f} ms  chain32 {d["roofline_one_batch"]["chain_ms_per_pass"]:.2f} ms')
PY
done; done
cp /tmp/lib_keep.so talkshow_amd/lib/libtalkshow_hip.so
