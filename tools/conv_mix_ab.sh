#!/bin/bash
# same-box A/B of the banded conv_gemm launch: conv_mix_ab.sh "ENV=.. ENV=.." "ENV=.." ...  (TS_CONV_MIX, TS_CONV_TAIL, TS_CONV_XCD)
for round in ${ROUNDS:-1 2}; do
for e in "$@"; do
  env $e TS_B=${TS_B:-256} python tools/conv_layers.py > /tmp/cl.txt 2>&1
  python - "$e" <<'PY'
import re, sys
rows = []
for l in open('/tmp/cl.txt'):
    m = re.match(r"\[ts_prof\] conv M=(\d+) N=(\d+) K=(\d+) groups=(\d+).*?([\d.]+) us\s+([\d.]+) TF", l)
    if m: rows.append([float(x) for x in m.groups()])
rows = rows[len(rows) // 2:]
tot = sum(r[4] for r in rows); fl = sum(r[4] * r[5] for r in rows)
big = [r for r in rows if r[3] >= 2 and r[2] >= 768]
def tf(sel): return sum(r[4] * r[5] for r in sel) / max(sum(r[4] for r in sel), 1e-9)
print(f"{sys.argv[1]:44s}: {tot / 1e3:.2f} ms  {fl / tot:.1f} TF | K=3072 {tf([r for r in big if r[2] == 3072]):.1f}  K=2048 {tf([r for r in big if r[2] == 2048]):.1f}  K=1536 {tf([r for r in big if r[2] == 1536]):.1f}  K=1024 {tf([r for r in big if r[2] == 1024]):.1f}  K=768 {tf([r for r in big if r[2] == 768]):.1f} TF")
PY
done; done
