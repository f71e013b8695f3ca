#!/usr/bin/env python3
"""Overlap structure of the 3-stream bench from a rocprofv3 kernel trace (tools/timeline.sh): per stream the chain phases
(hipGraph replays of the PixelCNN) and conv phases, how long each phase lasts with the other streams active, and how the
device time splits into chain-only / conv-only / both / idle.

    python tools/timeline.py <kernel_trace.csv>
"""
import csv, sys, json
from collections import defaultdict
import numpy as np

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    cls = "chain" if ("skinny" in n or "sample_kernel" in n) else ("conv" if "conv_gemm" in n else "other")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), cls, n))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
# the timed region = the last 60 % of the trace (warm-up and model build come first)
lo = t0 + 0.4 * (t1 - t0)
rows = [r for r in rows if r[0] >= lo]
t0, t1 = rows[0][0], rows[-1][1]
print(f"{len(rows)} dispatches over {(t1 - t0) / 1e6:.1f} ms, queues: {sorted(set(r[2] for r in rows))}")

# busy-time unions per class via sweep
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for a, b in iv[1:]:
        if a > ce: tot += ce - cs; cs, ce = a, b
        else: ce = max(ce, b)
    return tot + ce - cs
ev = []
for a, b, q, c, n in rows:
    ev.append((a, 1, c)); ev.append((b, -1, c))
ev.sort()
cnt = defaultdict(int); last = ev[0][0]; state = defaultdict(float)
for t, d, c in ev:
    key = ("chain" if cnt["chain"] else "") + ("+conv" if cnt["conv"] else "") + ("+other" if cnt["other"] else "")
    nch = cnt["chain"]
    state[key or "idle"] += t - last
    state[f"chain kernels in flight = {min(nch, 3)}"] += t - last
    last = t; cnt[c] += d
tot = t1 - t0
print("device time by what is running:")
for k, v in sorted(state.items(), key=lambda kv: -kv[1]):
    print(f"   {k:32s} {v / 1e6:8.2f} ms  {100 * v / tot:5.1f} %")

# phases per queue: maximal runs of same-class kernels (other = ignored if short)
byq = defaultdict(list)
for r in rows: byq[r[2]].append(r)
print("phases per queue (class, n kernels, span ms, sum of kernel durations ms):")
allph = []
for q, rs in byq.items():
    ph = []
    for a, b, _, c, n in rs:
        if c == "other": continue
        if ph and ph[-1][0] == c and a - ph[-1][2] < 2e6: ph[-1][2] = max(ph[-1][2], b); ph[-1][3] += 1; ph[-1][4] += b - a
        else: ph.append([c, a, b, 1, b - a])
    for c, a, b, k, s_ in ph: allph.append((q, c, a, b, k, s_))
for c in ("chain", "conv"):
    sel = [p for p in allph if p[1] == c and p[4] > (1000 if c == "chain" else 10)]
    if sel:
        span = np.array([(p[3] - p[2]) / 1e6 for p in sel]); busy = np.array([p[5] / 1e6 for p in sel]); k = np.array([p[4] for p in sel])
        print(f"   {c:6s}: {len(sel)} phases, kernels {np.median(k):.0f}, span median {np.median(span):.2f} ms (min {span.min():.2f} max {span.max():.2f}), kernel time median {np.median(busy):.2f} ms")

# kernel durations: alone vs with another class in flight (by midpoint)
starts = np.array([r[0] for r in rows]); 
def overl(cls_other):
    iv = sorted((a, b) for a, b, q, c, n in rows if c == cls_other)
    A = np.array([x[0] for x in iv]); B = np.maximum.accumulate(np.array([x[1] for x in iv]))
    def f(t):
        i = np.searchsorted(A, t, side="right") - 1
        return i >= 0 and B[i] > t
    return f
in_conv, in_chain = overl("conv"), overl("chain")
agg = defaultdict(lambda: [[], []])
for a, b, q, c, n in rows:
    if c == "other": continue
    short = n.split("(")[0].replace("void ", "").replace("ts::", "")[:60]
    mid = (a + b) // 2
    other = in_conv(mid) if c == "chain" else in_chain(mid)
    agg[short][1 if other else 0].append(b - a)
print("kernel duration (us): alone | with the other class in flight")
for n, (al, ov) in sorted(agg.items(), key=lambda kv: -(sum(kv[1][0]) + sum(kv[1][1]))):
    f = lambda v: f"{len(v):6d} x {np.mean(v) / 1e3:8.2f}" if v else "     0 x     -   "
    print(f"   {n:60s} {f(al)} | {f(ov)}")
